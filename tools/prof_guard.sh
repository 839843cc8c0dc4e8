#!/bin/bash
# per-kernel durations of tools/guard_rate.py (rocprofv3 --kernel-trace --stats; the program itself after `--`) -> gpurun_out/prof_guard/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
rm -rf $R/gpurun_out/prof_guard
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_guard -- $PY $R/tools/guard_rate.py > $R/gpurun_out/prof_guard.log 2>&1
cd $R
f=$(ls -t gpurun_out/prof_guard/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r06_guard_kernel_stats.csv; head -30 "$f" | cut -c1-220; else tail -5 gpurun_out/prof_guard.log; fi
