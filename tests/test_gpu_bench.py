"""bench.py on the GPU box: the JSON contract of the headline line, and the RCCL code path (process-group init,
max-over-ranks all_reduce, barrier, result gather) exercised with a one-rank group -- the 8-GPU run is the driver's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench(args, env_extra, tmp_path, timeout=900, clean_env=False):
    """run bench.py; returns (the parsed contract line, the detail dict it wrote).  The line is the ONLY stdout line that starts with
    "{", it is the LAST line of stdout -- whatever the libraries underneath print (RCCL's version banner is written to the C stdout
    buffer when the communicator is made and would otherwise surface at exit, after the line) -- and it is at most 4096 bytes: the
    driver could not parse round 5's 35 KB line (BENCH_r05.parsed = null)."""
    env = {k: v for k, v in os.environ.items() if not (clean_env and k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"))}
    detail = str(tmp_path / "detail.json")
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MA_BENCH_DETAIL=detail, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert p.stdout.strip().splitlines()[-1] == lines[0], p.stdout[-1500:]
    assert len(lines[0].encode()) <= 4096, len(lines[0])
    d = json.loads(lines[0])
    for key in CONTRACT:
        assert key in d, key
    full = json.load(open(detail))
    assert d["detail"] == "detail.json" and full["value"] == d["value"]
    assert any(l.startswith("# detail roofline: ") for l in p.stdout.splitlines())
    return d, full


def _run(args, env_extra, tmp_path):
    return _bench(args, env_extra, tmp_path)[1]


def test_bench_driver_command_exactly(tmp_path):
    """the command the driver runs at round end, no flag removed: `python bench.py --gpus 1 --steps 20 --warmup 5` -- every side leg, the CPU
    baseline and the rocprofv3 child passes included.  What the driver needs is in the <= 4 KB line; everything is in the detail file."""
    d, full = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], {}, tmp_path, timeout=1500)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    r, c, x = d["roofline"], d["cpu_baseline"], d["x25519"]
    assert r["bound"] == "hbm" and 0.3 < r["frac"] < 1.0 and 0.99 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert r["frac_first_placement"] > 0.3 and r["frac_median_placement"] > 0.3 and len(d["config"]["placement_probe_GBps"]) == 4
    assert c["kind"] == "port" and c["value"] > 1e6 and c["check_words_ok"] is True and c["check_words"] == 9 and c["cores"] >= 1 and c["ns_per_modmul_one_core"] > 1
    assert x["value"] > 1e7 and x["scalars_per_gpu"] == 1 << 23 and x["sclk_GHz"] > 1.0 and x["value_wall_clock_3_passes"] > 1e7
    # the VALU legs lead with the multiply-add-only ceiling (the figure that falls when instructions are wasted); the mix ceiling is secondary
    assert 0.3 < x["frac_of_mad_only_ceiling"] < x["frac_of_mix_ceiling"] <= 1.1
    assert x["mad_floor_per_scalar"] and 1.0 <= x["mad_per_scalar"] / x["mad_floor_per_scalar"] < 1.2
    assert d["verified_against_oracle"]["all_ranks_equal_oracle"] is True
    legs = [k for k in full["other_configs"] if "_ecn_" in k]
    assert len(legs) >= 24
    for k in legs + ["x25519", "x448"]:
        rl = full["other_configs"][k]["roofline"] if k in full["other_configs"] else full[k]["roofline"]
        assert rl["bound"] == "valu-mad" and 0.2 < rl["frac"] < 1.0 and rl["frac_of_mix_ceiling"] and rl["mad_floor_per_scalar"], k
        assert rl["mad_per_scalar"] >= 0.98 * rl["mad_floor_per_scalar"], k


def test_bench_line_contract(tmp_path):
    d = _run(["--steps", "20", "--warmup", "5", "--no-cpu", "--no-others"], {}, tmp_path)
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


def test_bench_rccl_path_one_rank(tmp_path):
    d = _run(["--steps", "10", "--warmup", "3", "--no-cpu", "--no-others", "--no-traffic"], {"MA_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29533"}, tmp_path)
    assert d["n_gpus"] == 1 and d["x25519"]["gather_ms"] is not None and d["x25519"]["gather_ms"] > 0


def test_bench_self_launch_two_ranks_gloo(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts both ranks (they share the one GPU of
    this box; gloo carries the collectives) and relays rank 0's single line."""
    # --scaling strong: BASELINE configs[4] literally -- 2^MA_BENCH_LOG2_LADDER_TOTAL records divided over the ranks in contiguous shards
    # (here 2^19 over two) and gathered to rank 0
    line, d = _bench(["--gpus", "2", "--steps", "10", "--warmup", "3", "--no-cpu", "--no-others", "--scaling", "strong"],
                     dict(MA_BENCH_BACKEND="gloo", MA_BENCH_LOG2_ELEMS="22", MA_BENCH_LOG2_LADDER_TOTAL="19"), tmp_path, clean_env=True)
    assert line["n_gpus"] == 2 and line["x25519"]["scalars_total"] == 1 << 19 and line["x25519"]["gather_ms"] > 0 and line["verified_against_oracle"]["ranks_checked"] == 2
    assert line["dist"] == {"backend": "gloo", "world_size": 2, "distinct_devices": 1} and len(line["rank_spread_modmul_per_s"]) == 3
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


def test_bench_eight_ranks_on_one_gpu_equal_one_rank(tmp_path):
    """Eight ranks before there are eight GPUs (round 5): `python bench.py --gpus 8 --scaling strong` on this one-GPU box -- the ranks
    share the device, gloo carries the collectives.  2^16 X25519 records, a function of their global index, are cut into eight
    contiguous shards (simd/README.md:4-16: independent units, no exchange step), every rank verifies its own shard against the oracle,
    rank 0 gathers all results; their digest must equal the digest of a ONE-rank run over the same 2^16 records."""
    def run(n):
        line, full = _bench(["--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-cpu", "--no-others", "--no-traffic", "--scaling", "strong"],
                            dict(MA_BENCH_BACKEND="gloo", MA_BENCH_LOG2_ELEMS="18", MA_BENCH_LOG2_LADDER_TOTAL="16", MA_BENCH_PLACEMENTS="1"), tmp_path, timeout=1500, clean_env=True)
        # the compact line alone carries what a SCALE record needs at every N
        assert line["n_gpus"] == n and line["verified_against_oracle"]["ranks_checked"] == n and line["x25519"]["records_sha256"] == full["x25519"]["records_sha256"]
        return full
    d8, d4, d2, d1 = run(8), run(4), run(2), run(1)
    assert d4["n_gpus"] == 4 and d2["n_gpus"] == 2 and d4["x25519"]["records_sha256"] == d2["x25519"]["records_sha256"] == d1["x25519"]["records_sha256"]
    assert d8["n_gpus"] == 8 and len(d8["ranks"]) == 8 and [r["rank"] for r in d8["ranks"]] == list(range(8))
    shards = [r["x25519_shard"] for r in d8["ranks"]]
    assert shards[0][0] == 0 and shards[-1][1] == 1 << 16 and all(shards[i][1] == shards[i + 1][0] for i in range(7))       # a partition of 2^16
    assert all(b - a == 1 << 13 for a, b in shards)
    v = d8["verified_against_oracle"]
    assert v["all_ranks_equal_oracle"] is True and v["ranks_checked"] == 8 and all(r["verified_against_oracle"] is True for r in d8["ranks"])
    assert d8["dist"] == {"backend": "gloo", "world_size": 8, "distinct_devices": 1}
    assert d8["x25519"]["scalars_total"] == 1 << 16 and d1["x25519"]["scalars_total"] == 1 << 16 and d1["n_gpus"] == 1
    assert d8["x25519"]["records_sha256"] and d8["x25519"]["records_sha256"] == d1["x25519"]["records_sha256"]
