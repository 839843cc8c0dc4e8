#!/bin/bash
# Round-4 PMC passes (GPU box): VALU-issue and wait counters of the scalar-multiplication kernels before (round-3 sources) and after
# (this tree) for ED25519 / ED448 / NIST P-256 through the A/B harness (tools/ecn_exp.hip), and of the fused chain beside the
# headline modmul (tools/run_chain.py).  Counters in their own runs with --kernel-trace only; the program itself after `--`.
#   bash tools/gpu_r04_pmc.sh   ->  gpurun_out/prof_r04/pmc*_*; summarise with tools/collect_r04_pmc.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r04
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
for c in ED25519 ED448 NIST256; do
  lg=20; [ $c = ED448 ] && lg=19
  for v in base cur; do
    rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/pmca_${v}_$c -- $R/tools/ecn_exp_${v}_$c.bin $lg 3 > $OUT/pmca_${v}_$c.log 2>&1
    rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/pmcb_${v}_$c -- $R/tools/ecn_exp_${v}_$c.bin $lg 3 > $OUT/pmcb_${v}_$c.log 2>&1
  done
done
rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/pmca_chain -- $PY $R/tools/run_chain.py 5 > $OUT/pmca_chain.log 2>&1
rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/pmcb_chain -- $PY $R/tools/run_chain.py 5 > $OUT/pmcb_chain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_chain -- $PY $R/tools/run_chain.py 10 > $OUT/stats_chain.log 2>&1
cd $R
# un-profiled rates of the same binaries, for the record
for c in ED25519 ED448 NIST256; do lg=20; [ $c = ED448 ] && lg=19; for v in base cur; do tools/ecn_exp_${v}_$c.bin $lg 7; done; done > $OUT/ecn_rates.log 2>&1
$PY tools/run_chain.py 10 > $OUT/chain_rates.log 2>&1
tail -n 40 $OUT/ecn_rates.log $OUT/chain_rates.log
find $OUT -name "*.csv" | wc -l
