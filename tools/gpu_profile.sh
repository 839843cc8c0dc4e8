#!/bin/bash
# rocprofv3 passes for bench.py (GPU box): kernel-trace+stats, then FETCH_SIZE and WRITE_SIZE PMC passes
# (separate runs, as the gfx950 counter slots require).  Usage: bash tools/gpu_profile.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
mkdir -p $OUT
# the program after `--` is the real interpreter binary (a python3 found through PATH may be a shim: an exec hop after the profiler's preload)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
# one operand placement in the profiled processes: the per-kernel average of --stats then covers the timed launches only
# (with the bench's placement probe on, it would also average the probe launches of the placements that were not kept)
export MA_BENCH_PLACEMENTS=1
# (--no-others: the side figures launch the same kernel symbol on other data sets and on the other HBM layout; without them the
# per-kernel average of --stats covers the headline launches only)
MA_BENCH_DETAIL=$OUT/bench_under_rocprof_detail.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PY $R/bench.py --steps 100 --warmup 10 --no-cpu --no-others --no-traffic > $OUT/bench_under_rocprof.log 2>&1
MA_BENCH_DETAIL=$OUT/pmc_detail.json timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $PY $R/bench.py --steps 5 --warmup 2 --no-cpu --no-ladder --no-others --no-traffic > $OUT/pmc_fetch.log 2>&1
MA_BENCH_DETAIL=$OUT/pmc_detail.json timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $PY $R/bench.py --steps 5 --warmup 2 --no-cpu --no-ladder --no-others --no-traffic > $OUT/pmc_write.log 2>&1
cd $R
unset MA_BENCH_PLACEMENTS
# the driver's command, as the driver runs it; the detail file beside bench.py is copied next to the log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null > $OUT/bench_plain.log 2>&1
cp $R/bench_detail.json $OUT/bench_plain_detail.json
tail -1 $OUT/bench_plain.log | cut -c1-600
