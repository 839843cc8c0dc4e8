#!/usr/bin/env python3
"""Summarise tools/gpu_r04_pmc.sh (gpurun_out/prof_r04/) into profiles/r04_ecn_pmc.json and profiles/r04_chain_pmc.json:
per kernel the counters of its longest dispatch (whole-GPU sums; GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_* cycle counters in
quad-cycles summed over waves), VALU instructions per unit, cycles per VALU instruction per SIMD, the clock seen during the pass."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r04")


def longest(files, sub):
    per, dur, grid = collections.defaultdict(dict), {}, {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                k = (f, r["Dispatch_Id"])
                per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                grid[k] = int(r["Grid_Size"])
    if not per:
        return None
    k = max(per, key=lambda k: dur[k])
    return dict(per[k], duration_us=dur[k] / 1e3, grid_threads=grid[k])


def derive(e, units):
    e["units"] = units
    if "SQ_INSTS_VALU" in e and "GRBM_GUI_ACTIVE" in e:
        e["valu_instr_per_unit"] = e["SQ_INSTS_VALU"] * 64 / units
        e["cycles_per_xcd"] = e["GRBM_GUI_ACTIVE"] / 8
        e["cycles_per_valu_instr_per_simd"] = e["cycles_per_xcd"] / (e["SQ_INSTS_VALU"] / 1024)
        e["gpu_clock_GHz"] = e["cycles_per_xcd"] / (e["duration_us"] * 1e3)
        e["units_per_s_under_profiler"] = units / (e["duration_us"] * 1e-6)
    return e


def merged(tag, sub, units):
    a = longest(glob.glob(os.path.join(SRC, "pmca_%s" % tag, "*", "*_counter_collection.csv")), sub)
    b = longest(glob.glob(os.path.join(SRC, "pmcb_%s" % tag, "*", "*_counter_collection.csv")), sub)
    if not a:
        return None
    e = derive(a, units)
    if b:
        for k, v in b.items():
            if k.startswith("SQ_"):
                e[k] = v
        e["duration_us_wait_pass"] = b["duration_us"]
        if e.get("SQ_WAVE_CYCLES"):
            e["wait_inst_fraction_of_wave_cycles"] = e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
            e["valu_active_fraction_of_wave_cycles"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) / e["SQ_WAVE_CYCLES"]
    return e


ecn = {"command": "tools/gpu_r04_pmc.sh: rocprofv3 --pmc <set A | set B> --kernel-trace --output-format csv -- tools/ecn_exp_{base,cur}_<CURVE>.bin <log2 n> 3",
       "note": "base = the round-3 curve layer (csrc at 11f6ab9) in the same harness; cur = this tree (ED25519: fh51.h resident form, four waves per SIMD). "
               "units = scalar multiplications of the profiled dispatch; rates under the profiler read 3-6 % low (gpurun_out/prof_r04/ecn_rates.log has the plain ones)"}
for c, lg in (("ED25519", 20), ("ED448", 19), ("NIST256", 20)):
    for v in ("base", "cur"):
        for kname, sub in (("mul", "k_ed_mul<"), ("mul2", "k_ed_mul2<")):
            e = merged("%s_%s" % (v, c), sub, 1 << lg)
            if e:
                ecn["%s_%s_%s" % (c, kname, v)] = e
json.dump(ecn, open(os.path.join(ROOT, "profiles", "r04_ecn_pmc.json"), "w"), indent=1)
chain = {"command": "tools/gpu_r04_pmc.sh: rocprofv3 --pmc <set A | set B> --kernel-trace -- python tools/run_chain.py 5 (2^24 elements of 2^255-19 on tiles of 4096)"}
for name, sub in (("k_chain_bench_prod", "k_chain<2>"), ("k_binary_modmul", "OpMulAuto<ma::P_X25519"), ("k_binary_modadd", "OpAdd<ma::P_X25519")):
    e = merged("chain", sub, 1 << 24)
    if e:
        chain[name] = e
json.dump(chain, open(os.path.join(ROOT, "profiles", "r04_chain_pmc.json"), "w"), indent=1)
for doc in (ecn, chain):
    for k, e in doc.items():
        if isinstance(e, dict):
            print("%-28s %9.1f us  %10.0f VALU/unit  %.2f cyc/instr  %.2f GHz  wait %.2f  valu-active %.2f" % (
                k, e["duration_us"], e.get("valu_instr_per_unit", 0), e.get("cycles_per_valu_instr_per_simd", 0), e.get("gpu_clock_GHz", 0),
                e.get("wait_inst_fraction_of_wave_cycles", 0), e.get("valu_active_fraction_of_wave_cycles", 0)))
