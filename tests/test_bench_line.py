"""The line the driver parses (bench.py contract_line): built here, without a GPU, from RECORDED detail dicts -- the full output of
round 5's run (profiles/r05_bench.json: the 35 KB line the driver could not parse, BENCH_r05.parsed = null), this round's
(profiles/r06_bench_detail.json, when present) and an eight-rank variant -- and held to the size and the keys the driver needs."""
import copy
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")
RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_bench*.json")))


def _check(line, d):
    assert len(line.encode()) <= bench.LINE_LIMIT == 4096 and "\n" not in line and line.startswith("{")
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out, k
    assert out["value"] == d["value"] and out["ms_per_step"] == d["ms_per_step"] and out["n_gpus"] == d["n_gpus"]      # full precision
    assert set(out["config"]) == {"workload", "elements_per_gpu", "placement_probe_GBps", "placement_policy"} and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    for k in ("traffic", "frac_first_placement", "frac_median_placement", "kernel_ms", "algorithmic_bytes_per_launch"):
        assert k in r
    assert out["detail"]
    return out


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_line_from_recorded_detail(path):
    d = json.load(open(path))
    if "other_configs" not in d:
        pytest.skip("not a detail record")
    out = _check(bench.contract_line(d), d)
    c = out["cpu_baseline"]
    if d.get("cpu_baseline") is None:                                   # a --no-cpu record (the profiler passes)
        assert c is None
        return
    assert c["kind"] == "port" and c["value"] > 1e6 and c["cores"] >= 1 and c["check_words_ok"] is True and c["check_words"] == 9
    assert c["ns_per_modmul_one_core"] > 1 and c["x25519_scalar_mults_per_s"] > 1e3
    assert out["x25519"]["value"] > 1e7 and out["x25519"]["scalars_per_gpu"] == 1 << 23
    assert out["verified_against_oracle"]["all_ranks_equal_oracle"] is True
    # and the detail goes to stdout on lines that cannot be taken for the contract line
    dl = bench.detail_lines(d)
    assert dl and all(l.startswith("# detail ") and "\n" not in l for l in dl)
    assert sum(1 for l in dl if l.startswith("# detail other_configs.")) == len(d["other_configs"])


def test_line_at_eight_ranks_with_every_optional_block():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    d = copy.deepcopy(d)
    d.update(n_gpus=8, cpu_baseline=None)
    d["ranks"] = [dict(d["ranks"][0], rank=i, local_rank=i) for i in range(8)]
    d["dist"] = {"backend": "nccl", "world_size": 8, "distinct_devices": 8}
    d["rank_spread"] = {"modmul_per_s": {"min": 5.1e10, "mean": 5.3e10, "max": 5.4e10}, "kernel_ms": {"min": 0.30, "mean": 0.31, "max": 0.32}}
    d["x25519"].update(scaling="strong", scalars_total=1 << 26, gather_ms=12.5, gather_GBps=171.8, records_sha256="ab" * 32, value_wall_clock_3_passes=9.1e8)
    d["verified_against_oracle"]["ranks_checked"] = 8
    d["config"]["placement_policy"] = "x" * 1000                       # free text is bounded
    d["roofline"]["traffic_source"] = "y" * 5000
    out = _check(bench.contract_line(d), d)
    assert out["cpu_baseline"] is None and out["dist"]["world_size"] == 8 and out["rank_spread_modmul_per_s"] == [5.1e10, 5.3e10, 5.4e10]
    assert out["x25519"]["records_sha256"] == "ab" * 32 and out["x25519"]["scalars_total"] == 1 << 26 and out["verified_against_oracle"]["ranks_checked"] == 8


def test_line_survives_oversized_blocks():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    d = copy.deepcopy(d)
    d["x25519"]["unit"] = "z" * 6000                                     # something unbounded sneaks in: the contract keys still arrive
    line = bench.contract_line(d)
    assert len(line) <= bench.LINE_LIMIT
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out


def test_algorithmic_floors_lie_below_the_kernels_counts():
    """profiles/mad_floor.json (tools/mad_floor.py: field operations of each leg's algorithm x schoolbook column products) is a LOWER bound:
    every leg's floor must not exceed the multiply-adds its kernels execute (profiles/r06_valu_pmc.json) -- a floor above the count means a
    formula in the tool is wrong -- and every VALU leg of the counter summary has a floor.  The file is what the tool writes now."""
    import subprocess
    floors = json.load(open(os.path.join(ROOT, "profiles", "mad_floor.json")))["legs"]
    legs = json.load(open(os.path.join(ROOT, "profiles", "r06_valu_pmc.json")))["legs"]
    assert set(legs) <= set(floors) and len(legs) == 26
    for leg, L in legs.items():
        f = floors[leg]["mad_floor_per_scalar"]
        assert 0.98 * f <= L["mad_per_scalar"] < 2.0 * f, (leg, f, L["mad_per_scalar"])
    before = open(os.path.join(ROOT, "profiles", "mad_floor.json")).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mad_floor.py")], check=True, capture_output=True, timeout=120)
    assert open(os.path.join(ROOT, "profiles", "mad_floor.json")).read() == before, "profiles/mad_floor.json is stale: python tools/mad_floor.py"
