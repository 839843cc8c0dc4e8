#!/usr/bin/env python3
"""python tools/collect_valu_legs_pmc.py <tag>: summarise tools/gpu_valu_legs_pmc.sh (gpurun_out/prof_legs/) into profiles/<tag>_valu_pmc.json: per
VALU-bound leg of bench.py (tools/valu_legs.py) and per kernel of the leg

  measured  SQ_INSTS_VALU (summed over the leg's dispatches of that kernel, whole GPU) x 64 / records = VALU instructions per record;
            GRBM_GUI_ACTIVE / 8 XCDs / duration = the shader clock under the profiler; duration
  static    tools/isa_mix.py on the object that was profiled (the same build: objects travel to the box, and this script runs in the
            build container next to them): VALU instructions and v_mad_u64_u32 per record from the disassembly weighted by loop
            trip counts, and the per-class mix

and per leg the sums: instr_per_scalar (measured), mad_per_scalar (measured: SQ_INSTS_VALU_INT64 = v_mad_u64_u32 + v_lshl_add_u64, less
the latter's static share), static_over_measured (how well the static count reproduces SQ_INSTS_VALU: exact where every loop's trip
count is inferred -- the ladders, k_ed25519_lad, k_ed_mul<ED25519 / ED448> --, off where the compiler's loop form defeats the inference:
those kernels keep their measured totals and only borrow the static SHARE of the two 64-bit classes).
bench.py's valu_roofline() reads the leg sums.  The summary records the build it was taken on: `unit_hashes` (modarith_amd/unit_hashes.json as
it was on the GPU box) and per leg the `units` its kernels come from; bench.py flags a leg whose units have been rebuilt since (source_stale)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix  # noqa: E402
from valu_legs import legs, records  # noqa: E402

SRC = os.path.join(ROOT, "gpurun_out", "prof_legs")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"


def dispatches(tag):
    """{(kernel name, dispatch id): {counter: value, duration_ns, grid}} of one pass"""
    per = collections.defaultdict(dict)
    files = sorted(glob.glob(os.path.join(SRC, tag, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                 # the newest pass only: gpurun merges every call's files into gpurun_out/, older passes stay there
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], r["Dispatch_Id"])
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            per[k]["duration_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            per[k]["grid"] = int(r["Grid_Size"])
    return per


def main():
    pa, pb = dispatches("pmca"), dispatches("pmcb")
    rounds_of = lambda n: max(1, min(32, (n + 65535) // 65536))
    uh_box = json.load(open(os.path.join(SRC, "unit_hashes.json"))) if os.path.exists(os.path.join(SRC, "unit_hashes.json")) else {}
    from modarith_amd.build import unit_hashes
    uh_here = unit_hashes()
    doc = {"command": "tools/gpu_valu_legs_pmc.sh: rocprofv3 --pmc <set A | set B> --kernel-trace --output-format csv -- <realpath of python> tools/run_valu_legs.py",
           "unit_hashes": uh_box, "objects_here_are_the_profiled_build": all(uh_here.get(k) == v for k, v in uh_box.items()) if uh_box else None,
           "interpreter": open(os.path.join(SRC, "interpreter.txt")).read().strip() if os.path.exists(os.path.join(SRC, "interpreter.txt")) else None,
           "note": "per leg: instr_per_scalar = measured SQ_INSTS_VALU x 64 / records over all kernels of the leg; mad_per_scalar = measured instructions x the "
                   "static v_mad_u64_u32 share of each kernel (tools/isa_mix.py on the profiled objects); issue-cost model 5.0 cycles per multiply-add, 2.5 per other "
                   "VALU instruction per wave and SIMD (profiles/history/r01_valubench.log)", "legs": {}}
    rates = {}
    rp = os.path.join(SRC, "valu_leg_rates.json")
    if os.path.exists(rp):
        rates = json.load(open(rp))
    for leg, ks in legs().items():
        n = records(leg)
        L = {"records": n, "units": sorted({o[:-2] for _, o, _, _ in ks}), "kernels": {}, "instr_per_scalar": 0.0, "mad_per_scalar": 0.0, "static_valu_per_scalar": 0.0, "duration_us": 0.0}
        cyc = 0.0
        for sub, obj, per_pass, unknown in ks:
            sel = [v for (name, _), v in pa.items() if sub in name]
            # a leg's kernel may also serve another leg of the same curve (k_ed_mul builds the input points of nobody here: each family runs once
            # per curve); several dispatches = the chunks of one call
            if not sel:
                continue
            valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in sel)
            dur = sum(v["duration_ns"] for v in sel)
            gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in sel)
            selb = [v for (name, _), v in pb.items() if sub in name]
            e = {"dispatches": len(sel), "SQ_INSTS_VALU": valu, "valu_instr_per_record": valu * 64 / n, "duration_us": dur / 1e3,
                 "gpu_clock_GHz_under_profiler": gui / 8 / dur if dur else None,
                 "SQ_INSTS_VALU_INT32_per_record": sum(v.get("SQ_INSTS_VALU_INT32", 0.0) for v in selb) * 64 / n if selb else None,
                 "SQ_INSTS_VALU_INT64_per_record": sum(v.get("SQ_INSTS_VALU_INT64", 0.0) for v in selb) * 64 / n if selb else None,
                 "SQ_ACTIVE_INST_VALU": sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in selb) if selb else None,
                 "SQ_BUSY_CYCLES": sum(v.get("SQ_BUSY_CYCLES", 0.0) for v in selb) if selb else None}
            ut = rounds_of(n) if unknown == "rounds" else 1
            found = isa_mix.kernels(obj, sub)
            if found:
                (_, kname), code = sorted(found.items())[0]
                st = isa_mix.analyse(code, unknown_trips=ut)
                scale = (1.0 / ut if unknown == "rounds" else 1.0) / per_pass      # the shared inversion's pass covers `rounds` records per lane
                cls = st["dynamic"]
                e["static"] = {"kernel": kname.split("(")[0], "valu_per_record": st["valu"] * scale, "mad_per_record": st["multiplier"] * scale,
                               "mad_share": st["multiplier"] / st["valu"], "non_mad_per_mad": st["non_multiplier_per_multiplier"],
                               "classes_per_record": {c: v * scale for c, v in cls.items()}, "loops": st["loops"]}
                e["static_over_measured"] = e["static"]["valu_per_record"] / e["valu_instr_per_record"] if e["valu_instr_per_record"] else None
                # v_mad_u64_u32 per record: measured instructions x the static multiply-add share where the static count is validated by the
                # counter.  Elsewhere SQ_INSTS_VALU_INT64 helps: on the unsigned-limb kernels it counts exactly v_mad_u64_u32 + v_lshl_add_u64
                # (checked on the kernels whose static count reproduces SQ_INSTS_VALU to the instruction: both ladders, k_ed25519_lad,
                # k_ed26l_prep), so mad = INT64 x the multiply-adds' share of those two classes in the kernel's code.
                i64 = e.get("SQ_INSTS_VALU_INT64_per_record")
                pair = cls.get("multiplier", 0) + cls.get("add_u64", 0)
                if e["static_over_measured"] and abs(e["static_over_measured"] - 1.0) <= 0.06:
                    e["mad_per_record"], e["mad_source"] = e["valu_instr_per_record"] * e["static"]["mad_share"], "SQ_INSTS_VALU x static v_mad share (static count within 6 % of the counter)"
                elif i64 and pair:
                    e["mad_per_record"], e["mad_source"] = i64 * cls.get("multiplier", 0) / pair, "estimate: SQ_INSTS_VALU_INT64 x static v_mad share of {v_mad_u64_u32, v_lshl_add_u64} (trip counts of this kernel not all inferred)"
                else:
                    e["mad_per_record"], e["mad_source"] = e["valu_instr_per_record"] * e["static"]["mad_share"], "estimate: SQ_INSTS_VALU x static v_mad share (trip counts of this kernel not all inferred)"
                L["mad_per_scalar"] += e["mad_per_record"]
                L["static_valu_per_scalar"] += e["static"]["valu_per_record"]
            L["instr_per_scalar"] += e["valu_instr_per_record"]
            L["duration_us"] += dur / 1e3
            cyc += gui / 8
            L["kernels"][sub] = e
        if not L["kernels"]:
            continue
        L["static_over_measured"] = L["static_valu_per_scalar"] / L["instr_per_scalar"] if L["instr_per_scalar"] else None
        L["non_mad_per_mad"] = (L["instr_per_scalar"] - L["mad_per_scalar"]) / L["mad_per_scalar"] if L["mad_per_scalar"] else None
        L["gpu_clock_GHz_under_profiler"] = cyc / (L["duration_us"] * 1e3) if L["duration_us"] else None
        L["records_per_s_under_profiler"] = n / (L["duration_us"] * 1e-6) if L["duration_us"] else None
        if leg in rates:
            L["unprofiled"] = rates[leg]
        doc["legs"][leg] = L
    out = os.path.join(ROOT, "profiles", TAG + "_valu_pmc.json")
    json.dump(doc, open(out, "w"), indent=1)
    # the --stats summary of the same pass (newest file only: gpurun merges every pass it has seen into gpurun_out/) and the leg rates
    import glob
    import shutil
    # (tools/gpu_profile.sh r05 writes bench.py's --stats pass into the same directory: the leg pass is the one that holds a curve kernel)
    st = [f for f in sorted(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime) if "k_ed_mul<" in open(f).read()]
    if st:
        shutil.copy(st[-1], os.path.join(ROOT, "profiles", TAG + "_legs_kernel_stats.csv"))
    if os.path.exists(os.path.join(SRC, "leg_rates.log")):
        shutil.copy(os.path.join(SRC, "leg_rates.log"), os.path.join(ROOT, "profiles", TAG + "_leg_rates.log"))
    print("%-32s %11s %11s %6s %7s %7s %9s" % ("leg", "VALU/rec", "mad/rec", "o/mad", "st/meas", "GHz(p)", "rate"))
    for leg, L in doc["legs"].items():
        print("%-32s %11.0f %11.0f %6.2f %7.3f %7.3f %9.3e" % (leg, L["instr_per_scalar"], L["mad_per_scalar"], L["non_mad_per_mad"] or 0, L["static_over_measured"] or 0,
                                                            L["gpu_clock_GHz_under_profiler"] or 0, (L.get("unprofiled") or {}).get("per_s", 0)))


if __name__ == "__main__":
    main()
