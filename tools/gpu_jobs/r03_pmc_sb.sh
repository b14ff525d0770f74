#!/bin/bash
# Round 3: stall / LDS / MFMA counters of the split-bf16 kernels at the benched shapes (one pass per counter set: 8 SQ slots).
export TMPDIR=/tmp
export CSEG_BENCH_GUARD=0      # job scripts choose the routes themselves: no automatic re-run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03pmc
mkdir -p $O
cd /tmp
run() {   # name, counters, command...
  local name=$1 ctr=$2; shift 2
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/$name -o p --output-format csv -- "$@" > $O/$name.out 2> $O/$name.err
  f=$(find $O/$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/summarize_pmc.py $f > $O/$name.summary.txt 2>&1
  rm -rf $O/$name
}
SETA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
SETB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run fwd_a "$SETA" python $R/tools/conv3x3_sb_probe.py head_720 branch_96 branch_48
run fwd_b "$SETB" python $R/tools/conv3x3_sb_probe.py head_720 branch_96 branch_48
run wrw_a "$SETA" python $R/tools/conv3x3_sb_wrw_probe.py branch_48 head_720
run wrw_b "$SETB" python $R/tools/conv3x3_sb_wrw_probe.py branch_48 head_720
tail -n +1 $O/*.summary.txt | cut -c1-220 | head -120
