#!/bin/bash
# instruction-cache counters of the scalar-multiplication kernels (GPU box): kernels of 64-285 KB against a 64 KB I-cache per CU pair.
# Usage: bash tools/gpu_icache_pmc.sh <tag>
set -u
TAG=${1:-r02i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the program after `--` is the real interpreter binary (a python3 found through PATH may be a shim: an exec hop after the profiler's preload)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" $OUT/counters.txt | sort -u > $OUT/sqc_names.txt
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_valu -- $PY $R/tools/time_ecn.py ED25519 NIST256 SECP256K1 NIST384 NUMS256W ED448 > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-trace --output-format csv -d $OUT/pmc_icache -- $PY $R/tools/time_ecn.py ED25519 NIST256 SECP256K1 NIST384 NUMS256W ED448 > $OUT/pmc_icache.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_ifetch -- $PY $R/tools/time_ecn.py ED25519 NIST256 SECP256K1 NIST384 NUMS256W ED448 > $OUT/pmc_ifetch.log 2>&1
cd $R
cat $OUT/sqc_names.txt | tr '\n' ' '
tail -3 $OUT/pmc_*.log
