#!/bin/bash
# Counter passes (GPU box) over tools/run_valu_legs.py: every VALU-bound leg of bench.py once (both ladders; ecn mul / mul2 and the
# fused mul_get / mul2_get / mulgen_get / mulgen2_get kernels of ED25519, ED448, NIST P-256, secp256k1).  Counters in their own runs with
# --kernel-trace only; the program itself (the resolved interpreter) after `--`.  Then the same legs timed without a profiler, with the
# shader-clock probe (modarith_amd/clock.py).
#   bash tools/gpu_valu_legs_pmc.sh   ->  gpurun_out/prof_legs/ ; summarise in the build container (next to the objects) with
#   python tools/collect_valu_legs_pmc.py <tag>  ->  profiles/<tag>_valu_pmc.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_legs
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
mkdir -p $OUT
rm -rf $OUT/pmca $OUT/pmcb $OUT/stats
cp $R/modarith_amd/unit_hashes.json $OUT/unit_hashes.json
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
B="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
timeout 900 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/pmca -- $PY $R/tools/run_valu_legs.py > $OUT/pmca.log 2>&1
timeout 900 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/pmcb -- $PY $R/tools/run_valu_legs.py > $OUT/pmcb.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PY $R/tools/run_valu_legs.py > $OUT/stats.log 2>&1
echo "$PY" > $OUT/interpreter.txt
cd $R
timeout 600 $PY tools/run_valu_legs.py --time > $OUT/leg_rates.log 2>&1
cp gpurun_out/valu_leg_rates.json $OUT/ 2>/dev/null
tail -n 30 $OUT/leg_rates.log
find $OUT -name "*.csv" | wc -l
# keep what comes back small: the counter files hold one row per dispatch, counter and XCD
find $OUT -name "*_agent_info.csv" -delete
