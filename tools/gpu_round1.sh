#!/bin/bash
# profile + sweep batch for one gpurun call (GPU box).  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
# 1. kernel-trace + stats of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/stats -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu > $OUT/prof/bench_under_rocprof.log 2>&1
# 2. PMC passes (separate runs, no trace domains beyond kernel-trace)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-ladder > $OUT/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-ladder > $OUT/prof/pmc_write.log 2>&1
cd $R
# 3. sweeps
python3 tools/sweep.py 24 30 > $OUT/sweep_default.json 2>$OUT/sweep_default.err
for mb in 512 1024 2048 8192 16384 65536; do
  MA_MAX_BLOCKS=$mb python3 tools/sweep.py 24 20 > $OUT/sweep_mb$mb.json 2>/dev/null
done
MA_FORCE_EPT1=1 python3 tools/sweep.py 24 20 > $OUT/sweep_ept1.json 2>/dev/null
ls -R $OUT/prof | head -50
