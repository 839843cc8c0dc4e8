#!/bin/bash
# VALU-issue PMC passes for the ladder and curve kernels (GPU box).  Counters in their own runs with --kernel-trace only.
# Usage: bash tools/gpu_valu_pmc.sh <tag>   -> gpurun_out/prof_<tag>/pmc_*; summarise with tools/collect_valu_pmc.py <tag>
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the program after `--` is the real interpreter binary (a python3 found through PATH may be a shim: an exec hop after the profiler's preload)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_ladder -- $PY $R/tools/run_ladder.py > $OUT/pmc_ladder.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_ecn -- $PY $R/tools/time_ecn.py ED25519 ED448 > $OUT/pmc_ecn.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_ecnwait -- $PY $R/tools/time_ecn.py ED25519 > $OUT/pmc_ecnwait.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_ladderwait -- $PY $R/tools/run_ladder.py > $OUT/pmc_ladderwait.log 2>&1
cd $R
tail -2 $OUT/pmc_*.log
