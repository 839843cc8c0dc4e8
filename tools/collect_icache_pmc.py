#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/pmc_{valu,icache,ifetch} (tools/gpu_icache_pmc.sh) into profiles/<tag>_icache_pmc.json:
per scalar-multiplication kernel the VALU issue rate, the instruction-cache hit rate and the wait fractions."""
import collections, csv, glob, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02i"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(d):
    f = glob.glob(os.path.join(root, "gpurun_out", "prof_" + tag, d, "runc", "*_counter_collection.csv"))[0]
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_ed_mul2?<ma::(\w+)<ma::C_(\w+)>", r["Kernel_Name"])
        if not m:
            continue
        k = (("mul2 " if "k_ed_mul2" in r["Kernel_Name"] else "mul ") + m.group(2), r["Dispatch_Id"])
        disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[k]["grid_threads"] = float(r["Grid_Size"])
    best = {}
    for (s, _), c in disp.items():                      # the timed launch is the one with the largest grid (the first is a 4096-point warm-up)
        if s not in best or c["grid_threads"] > best[s]["grid_threads"]:
            best[s] = c
    return best
out = {}
for d in ("pmc_valu", "pmc_icache", "pmc_ifetch"):
    for s, c in load(d).items():
        out.setdefault(s, {}).update(c)
for s, c in out.items():
    c["cycles_per_valu_instr_per_simd"] = (c["GRBM_GUI_ACTIVE"] / 8) / (c["SQ_INSTS_VALU"] / 1024)
    c["icache_miss_rate"] = c["SQC_ICACHE_MISSES"] / c["SQC_ICACHE_REQ"]
    c["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    c["wait_inst_any_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
res = {"command": "bash tools/gpu_icache_pmc.sh " + tag + " (three rocprofv3 --pmc passes over tools/time_ecn.py, --kernel-trace only)",
       "note": "whole-GPU sums; GRBM_GUI_ACTIVE summed over 8 XCDs, 1024 SIMDs; kernel code sizes (llvm-readelf on the gfx950 code objects): "
               "ED25519 mul 64 KB, SECP256K1 97 KB, NIST256 138 KB, ED448 141 KB, NIST384 285 KB against a 64 KB instruction cache per CU pair",
       "kernels": {k: out[k] for k in sorted(out)}}
p = os.path.join(root, "profiles", tag.rstrip("i") + "_icache_pmc.json" if tag.endswith("i") else tag + "_icache_pmc.json")
json.dump(res, open(p, "w"), indent=1)
for k in sorted(out):
    c = out[k]
    print("%-16s %.2f cyc/VALU  icache miss %.4f %%  wait_any %.1f %%  wait_inst %.1f %%" % (k, c["cycles_per_valu_instr_per_simd"], 100 * c["icache_miss_rate"], 100 * c["wait_any_frac"], 100 * c["wait_inst_any_frac"]))
print("->", p)
