#!/bin/bash
# VALU-issue PMC passes for the (split) ladder kernels only (GPU box); counters in their own runs with --kernel-trace.
# Usage: bash tools/gpu_ladder_pmc.sh <tag>  -> gpurun_out/prof_<tag>/pmc_*; summarise with tools/collect_valu_pmc.py <tag>
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the program after `--` is the real interpreter binary (a python3 found through PATH may be a shim: an exec hop after the profiler's preload)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_ladder -- $PY $R/tools/run_ladder.py > $OUT/pmc_ladder.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_ladderwait -- $PY $R/tools/run_ladder.py > $OUT/pmc_ladderwait.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ladder -- $PY $R/tools/run_ladder.py > $OUT/stats_ladder.log 2>&1
cd $R
tail -2 $OUT/pmc_*.log
