#!/bin/bash
# Round 4, second half: same-box A/B of the scalar-multiplication kernels through tools/ecn_exp.hip -- P-256 and secp256k1 with the
# digit x prime-limb terms as shifts (`old`: -DMA_MHALF_SHIFT_TERMS) and as multiply-adds (`fin`), ED448 on the limb form (`u1`,
# first half of the round) and on fh56.h (`fin`): rates, limb digests, and SQ_INSTS_VALU / GRBM_GUI_ACTIVE per kernel.
#   tools/build_ecn_exp.sh fin modarith_amd/csrc NIST256 -DMA_MUL_WPS=3;  ... old ... -DMA_MHALF_SHIFT_TERMS   (see docs/curve_layer.md)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r04_final
mkdir -p $OUT
cd $R
for rep in 1 2; do
  for pair in "NIST256 old 19" "NIST256 fin 19" "SECP256K1 old 19" "SECP256K1 fin 19" "ED448 u1 19" "ED448 fin 19"; do
    set -- $pair
    tools/ecn_exp_$2_$1.bin $3 7
  done
done > $OUT/ab_rates.log 2>&1
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
for pair in "NIST256 old" "NIST256 fin" "SECP256K1 old" "SECP256K1 fin" "ED448 u1" "ED448 fin"; do
  set -- $pair
  rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/pmca_$2_$1 -- $R/tools/ecn_exp_$2_$1.bin 19 1 > $OUT/pmca_$2_$1.log 2>&1
done
cd $R
cat $OUT/ab_rates.log
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmca_*")):
    if not d.endswith(".log"):
        best = None
        for f in glob.glob(d + "/*/*_counter_collection.csv"):
            per = collections.defaultdict(dict); dur = {}
            for r in csv.DictReader(open(f)):
                if "k_ed_mul" in r["Kernel_Name"]:
                    k = r["Dispatch_Id"]; per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if per:
                k = max(per, key=lambda k: dur[k]); best = (per[k], dur[k])
        if best:
            c, t = best
            n = 1 << 19
            print("%-28s VALU instr per scalar %9.0f   cycles per VALU instr per SIMD %.3f   clock %.3f GHz   %.3e/s under the profiler"
                  % (d.split("pmca_")[-1], c["SQ_INSTS_VALU"] * 64 / n, (c["GRBM_GUI_ACTIVE"] / 8) / (c["SQ_INSTS_VALU"] / 1024), (c["GRBM_GUI_ACTIVE"] / 8) / t, n / (t * 1e-9)))
PY
