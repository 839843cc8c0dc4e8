# VALU-issue and wait counters of the fused mul_get kernels (GPU box).  Usage: bash tools/gpu_fused_pmc.sh [tag] [curves ...]
#   -> gpurun_out/prof_<tag>/pmc_fused, pmc_fusedwait; summarise with tools/collect_valu_pmc.py <tag> name=substring:units ...
set -u
TAG=${1:-r02e}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# the program after `--` is the real interpreter binary (a python3 found through PATH may be a shim: an exec hop after the profiler's preload)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_fused -- $PY $R/tools/run_fused.py "$@" > $OUT/pmc_fused.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_fusedwait -- $PY $R/tools/run_fused.py "$@" > $OUT/pmc_fusedwait.log 2>&1
tail -n 2 $OUT/pmc_fused.log
