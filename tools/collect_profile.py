#!/usr/bin/env python3
"""Copy the judged summaries of a tools/gpu_profile.sh run from gpurun_out/ into profiles/ (tracked).
   python tools/collect_profile.py <tag>"""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_%s" % tag
os.makedirs("profiles", exist_ok=True)
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]      # gpurun_out/ keeps the files of earlier runs of the same tag
rows = list(csv.reader(open(newest(src + "/stats/*/*_kernel_stats.csv"))))
with open("profiles/%s_bench_kernel_stats.csv" % tag, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu --no-others (kernel names trimmed to 140 chars)"])
    for r in rows:
        r[0] = r[0][:140]
        w.writerow(r)
def pmc(kind):
    f = newest(src + "/pmc_%s/*/*_counter_collection.csv" % kind)
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_binary<ma::P_X25519, ma::OpMulAuto<ma::P_X25519" in r["Kernel_Name"]]
fetch, write = pmc("fetch"), pmc("write")
fk, wk = sum(fetch) / len(fetch), sum(write) / len(write)
stat = [r for r in rows if r and "k_binary<ma::P_X25519, ma::OpMulAuto<ma::P_X25519" in r[0]][0]
doc = {"tag": tag, "kernel": "ma::k_binary<ma::P_X25519, ma::OpMulAuto<ma::P_X25519>, 2>", "launches_sampled": len(fetch),
       "rocprof_stats_avg_ns": float(stat[3]), "rocprof_stats_calls": int(stat[1]),
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-ladder (two separate passes)",
       "FETCH_SIZE_KB_per_launch_raw": fk, "WRITE_SIZE_KB_per_launch_raw": wk,
       "correction": "gfx950: FETCH_SIZE reports exactly 1/2 of the bytes of a 16-B-per-lane coalesced streaming read (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact for 16-B-per-lane stores",
       "hbm_read_bytes_per_launch": 2 * fk * 1024, "hbm_write_bytes_per_launch": wk * 1024,
       "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024, "algorithmic_bytes_per_launch": 120 * (1 << 24)}
doc["traffic_over_algorithmic"] = doc["hbm_bytes_per_launch"] / doc["algorithmic_bytes_per_launch"]
json.dump(doc, open("profiles/traffic_modmul_X25519.json", "w"), indent=1)
# round 6: the last stdout line of bench.py is the compact contract line; the full record is the detail file
line = open(src + "/bench_plain.log").read().strip().splitlines()[-1]
open("profiles/%s_bench_line.json" % tag, "w").write(line + "\n")
json.dump(json.load(open(src + "/bench_plain_detail.json")), open("profiles/%s_bench.json" % tag, "w"), indent=1)
if os.path.exists(src + "/bench_under_rocprof_detail.json"):
    json.dump(json.load(open(src + "/bench_under_rocprof_detail.json")), open("profiles/%s_bench_under_rocprof.json" % tag, "w"), indent=1)
print(json.dumps(doc, indent=1))
print(len(line), line[:400])
