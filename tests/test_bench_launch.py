"""`python bench.py --gpus N` must start its own N ranks (the driver's N=1 form, used for N>1 too): CPU check of the
launcher -- the parent spawns torch.distributed.run, the ranks rendezvous on 127.0.0.1 over gloo, rank 0 prints ONE
JSON line, and a failing rank's exit code comes back.  The GPU workload under the same launch is in test_gpu_bench.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks():
    p = _bench(["--gpus", "2", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "n_gpus": 2, "rank_sum": 3}


def test_self_launch_propagates_failure():
    # without a GPU the real workload cannot start: the ranks fail and the parent must return non-zero, not hang
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-others", "--no-ladder"])
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
