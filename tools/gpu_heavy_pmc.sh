#!/bin/bash
# VALU-issue and wait counters of the long-chain field kernels (modinv / modsqrt / modpro), GPU box.
# Usage: bash tools/gpu_heavy_pmc.sh <tag> [primes...]
set -u
TAG=${1:-r03h}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the program after `--` is the real interpreter binary (a python3 found through PATH may be a shim: an exec hop after the profiler's preload)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_heavy -- $PY $R/tools/time_field.py child "$@" > $OUT/pmc_heavy.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_heavywait -- $PY $R/tools/time_field.py child "$@" > $OUT/pmc_heavywait.log 2>&1
cd $R
tail -3 $OUT/pmc_heavy.log
