"""bench.py on the GPU box: the JSON contract of the headline line, and the RCCL code path (process-group init,
max-over-ranks all_reduce, barrier, result gather) exercised with a one-rank group -- the 8-GPU run is the driver's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    # the JSON line is the LAST line of stdout, whatever the libraries underneath print (RCCL's version banner is written to the C
    # stdout buffer when the communicator is made and would otherwise surface at exit, after the line)
    assert p.stdout.strip().splitlines()[-1] == lines[0], p.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run(["--steps", "20", "--warmup", "5", "--no-cpu", "--no-others"], {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    # value and the roofline describe the same launches: modmul/s * 120 B within 10 % of the achieved rate
    assert abs(d["value"] * 120 / 1e9 - r["achieved"]) / r["achieved"] < 0.10
    assert d["x25519"]["value"] > 1e7
    assert d["verified_against_oracle"]["all_ranks_equal_oracle"] is True and len(d["ranks"]) == 1
    assert d["roofline"]["frac_median_placement"] <= d["roofline"]["frac"] * 1.15 and d["value_median_placement"] > 0
    assert len(d["config"]["placement_probe_GBps"]) == 4
    # roofline.traffic is measured in the run itself: rocprofv3 --pmc child passes after the parent's timed regions
    assert "measured in this run" in r["traffic_source"], r["traffic_source"]
    assert 0.99 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05


def test_bench_rccl_path_one_rank():
    d = _run(["--steps", "10", "--warmup", "3", "--no-cpu", "--no-others", "--no-traffic"], {"MA_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert d["n_gpus"] == 1 and d["x25519"]["gather_ms"] is not None and d["x25519"]["gather_ms"] > 0


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts both ranks (they share the one GPU of
    this box; gloo carries the collectives) and relays rank 0's single line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # --scaling strong: BASELINE configs[4] literally -- 2^MA_BENCH_LOG2_LADDER_TOTAL records divided over the ranks in contiguous shards
    # (here 2^19 over two) and gathered to rank 0
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MA_BENCH_BACKEND="gloo", MA_BENCH_LOG2_ELEMS="22", MA_BENCH_LOG2_LADDER_TOTAL="19")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-cpu", "--no-others", "--scaling", "strong"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["elements_per_gpu"] == 1 << 22
    assert d["x25519"]["scalars_per_gpu"] == 1 << 18 and d["x25519"]["gather_ms"] > 0 and d["x25519"]["value"] > 0
    assert d["x25519"]["scaling"] == "strong" and d["x25519"]["scalars_total"] == 1 << 19 and [r["x25519_records"] for r in d["ranks"]] == [1 << 18, 1 << 18]
    assert d["x25519"]["gather_GBps"] > 0
    # the N > 1 line verifies itself: every rank checked its own outputs against the oracle, AND-reduced
    v = d["verified_against_oracle"]
    assert v is not None and v["all_ranks_equal_oracle"] is True and v["ranks_checked"] == 2
    assert v["modmul_elements_per_rank"] >= 8192 and v["x25519_records_per_rank"] >= 8192
    # per-rank inventory and what the collective layer saw
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    for r in d["ranks"]:
        assert r["device_name"] and r["modmul_per_s"] > 0 and r["x25519_per_s"] > 0 and r["verified_against_oracle"] is True
    assert d["dist"] == {"backend": "gloo", "world_size": 2, "distinct_devices": 1}
    sp = d["rank_spread"]["modmul_per_s"]
    assert sp["min"] <= sp["mean"] <= sp["max"]
    assert d["value"] <= 2 * sp["max"] * 1.001
    # the N = 1 side figures are not run at N > 1
    assert d["other_configs"] == {} and list(d["data_sets"]) == ["uniform_mod_p"]


def test_bench_eight_ranks_on_one_gpu_equal_one_rank():
    """Eight ranks before there are eight GPUs (round 5): `python bench.py --gpus 8 --scaling strong` on this one-GPU box -- the ranks
    share the device, gloo carries the collectives.  2^16 X25519 records, a function of their global index, are cut into eight
    contiguous shards (simd/README.md:4-16: independent units, no exchange step), every rank verifies its own shard against the oracle,
    rank 0 gathers all results; their digest must equal the digest of a ONE-rank run over the same 2^16 records."""
    def run(n):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MA_BENCH_BACKEND="gloo", MA_BENCH_LOG2_ELEMS="18", MA_BENCH_LOG2_LADDER_TOTAL="16", MA_BENCH_PLACEMENTS="1")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-cpu", "--no-others", "--no-traffic",
                            "--scaling", "strong"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and p.stdout.strip().splitlines()[-1] == lines[0], p.stdout[-2000:]      # the JSON line is the last line
        return json.loads(lines[0])
    d8, d1 = run(8), run(1)
    assert d8["n_gpus"] == 8 and len(d8["ranks"]) == 8 and [r["rank"] for r in d8["ranks"]] == list(range(8))
    shards = [r["x25519_shard"] for r in d8["ranks"]]
    assert shards[0][0] == 0 and shards[-1][1] == 1 << 16 and all(shards[i][1] == shards[i + 1][0] for i in range(7))       # a partition of 2^16
    assert all(b - a == 1 << 13 for a, b in shards)
    v = d8["verified_against_oracle"]
    assert v["all_ranks_equal_oracle"] is True and v["ranks_checked"] == 8 and all(r["verified_against_oracle"] is True for r in d8["ranks"])
    assert d8["dist"] == {"backend": "gloo", "world_size": 8, "distinct_devices": 1}
    assert d8["x25519"]["scalars_total"] == 1 << 16 and d1["x25519"]["scalars_total"] == 1 << 16 and d1["n_gpus"] == 1
    assert d8["x25519"]["records_sha256"] and d8["x25519"]["records_sha256"] == d1["x25519"]["records_sha256"]
