set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02ecn
rm -rf $OUT; mkdir -p $OUT
# the program after `--` is the real interpreter binary (a python3 found through PATH may be a shim: an exec hop after the profiler's preload)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ecn -- $PY $R/tools/time_ecn.py > $OUT/time_ecn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fused -- $PY $R/tools/time_fused.py > $OUT/time_fused.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ladder -- $PY $R/tools/ladder_rate.py > $OUT/ladder_rate.log 2>&1
grep -v amdgpu $OUT/time_ecn.log | grep -v rocprof | tail -24; grep "fused" $OUT/time_fused.log; grep "scalar mults" $OUT/ladder_rate.log
