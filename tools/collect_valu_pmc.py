#!/usr/bin/env python3
"""Summarise a VALU-issue PMC pass into profiles/<tag>_valu_pmc.json.

  on the GPU box (counters in their own run, --kernel-trace only, as gpurun requires):
     cd /tmp && export TMPDIR=/tmp
     rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
               -d $R/gpurun_out/prof_<tag>/pmc_ladder -- python3 $R/tools/run_ladder.py
  here:  python tools/collect_valu_pmc.py <tag> [more "name=substring:units" entries]

Per kernel (largest-grid dispatch): SQ_INSTS_VALU (wave-instructions, whole GPU), GRBM_GUI_ACTIVE (summed over the 8 XCDs),
duration from the dispatch timestamps; derived: VALU instructions per SIMD (1024 SIMDs), cycles per XCD, cycles per VALU
instruction per SIMD, clock seen during the pass.  Nothing is estimated here: issue costs of the mix belong to DESIGN.md."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1]
src = "gpurun_out/prof_%s" % tag
# (substring of the demangled kernel name, units of the profiled pass); round 3: the split ladders = ladder kernel + batched finish
want = {"k_x25519_fe26_xz": ("k_x25519_fe26_xz", 1 << 22), "k_fe_finish_fe26": ("k_fe_finish<ma::Fe26", 1 << 22),
        "k_x448_fe28_xz": ("k_x448_fe28_xz", 1 << 20), "k_fe_finish_fe28": ("k_fe_finish<ma::Fe28", 1 << 20),
        "k_x25519_fe26": ("k_x25519_fe26(", 1 << 22), "k_x448_fe28": ("k_x448_fe28(", 1 << 20)}
for a in sys.argv[2:]:
    name, rest = a.split("=")
    sub, units = rest.rsplit(":", 1)
    want[name] = (sub, int(units))
doc = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE [...] --kernel-trace --output-format csv -- python3 tools/run_ladder.py (and tools/time_ecn.py for the curve kernels)",
       "note": "whole-GPU sums; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs; SQ_INSTS_VALU counts wave-level instructions"}
for f in glob.glob(src + "/pmc_*/*/*_counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(f)):
        for name, (sub, units) in want.items():
            if sub in r["Kernel_Name"]:
                key = (name, int(r["Grid_Size"]), r["Dispatch_Id"])
                per[key][r["Counter_Name"]] += float(r["Counter_Value"])
                dur[key] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for name in want:
        keys = [k for k in per if k[0] == name]
        if not keys:
            continue
        k = max(keys, key=lambda k: (k[1], dur[k]))
        c, units = per[k], want[name][1]
        e = doc.setdefault(name, {"units": units, "scalars": units, "grid_threads": k[1], "duration_us": dur[k] / 1e3, "source": os.path.relpath(f, "gpurun_out")})
        e.update({n: v for n, v in c.items()})
        if "SQ_INSTS_VALU" in e and "GRBM_GUI_ACTIVE" in e:
            e["valu_instr_per_simd"] = e["SQ_INSTS_VALU"] / 1024
            e["cycles_per_xcd"] = e["GRBM_GUI_ACTIVE"] / 8
            e["cycles_per_valu_instr_per_simd"] = e["cycles_per_xcd"] / e["valu_instr_per_simd"]
            e["gpu_clock_GHz"] = e["cycles_per_xcd"] / (e["duration_us"] * 1e3)
            e["valu_instr_per_unit"] = e["SQ_INSTS_VALU"] * 64 / units
# warm-up dispatches of the self-contained kernels (a few thousand records) are not passes over `units` records: drop them
for name in [n for n, e in doc.items() if isinstance(e, dict) and e.get("SQ_WAVES", 1e9) < 1024]:
    del doc[name]
json.dump(doc, open("profiles/%s_valu_pmc.json" % tag, "w"), indent=1)
print(json.dumps(doc, indent=1))
