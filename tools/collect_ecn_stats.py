#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_ecn_stats.sh run (gpurun_out/prof_r02ecn) into profiles/: the three rocprofv3 --stats
kernel tables (newest run of each leg) into r02_ecn_kernel_stats.csv, the timing lines of the same runs into r02_curve_timings.txt."""
import csv, glob, os, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r02ecn")
legs = (("ecn", "tools/time_ecn.py (ecn mul / mul2, all eleven curves)"),
        ("fused", "tools/time_fused.py (fused mul_get / mul2_get against the two-call forms)"),
        ("ladder", "tools/ladder_rate.py (rfc7748 X25519 / X448)"))
with open(os.path.join(root, "profiles", "r02_ecn_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for leg, what in legs:
        files = sorted(glob.glob(os.path.join(src, leg, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        w.writerow(["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 %s; kernel names trimmed to 140 chars" % what])
        for r in csv.reader(open(files[-1])):
            if r:
                r[0] = r[0][:140]
                w.writerow(r)
keep = re.compile(r"ecn mul|fused mul|scalar mults/s")
with open(os.path.join(root, "profiles", "r02_curve_timings.txt"), "w") as f:
    f.write("# final round-2 build, one GPU box, under rocprofv3 --kernel-trace --stats (tools/gpu_ecn_stats.sh)\n")
    for log in ("time_ecn.log", "time_fused.log", "ladder_rate.log"):
        for l in open(os.path.join(src, log), errors="replace"):
            if keep.search(l) and "rocprofv3" not in l:
                f.write(l)
print(open(os.path.join(root, "profiles", "r02_curve_timings.txt")).read())
