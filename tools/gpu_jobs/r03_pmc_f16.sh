#!/bin/bash
# Counters of the f16x3 split kernels at the benched shapes: one pass per counter set (8 SQ slots), never together with a trace
# domain other than the kernel trace.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03pmc_f16
mkdir -p $O
cd /tmp
run() {   # name, counters
  local name=$1 ctr=$2
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $O/$name -o p --output-format csv -- python $R/tools/head_kernels_only.py branch > $O/$name.out 2> $O/$name.err
  f=$(find $O/$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/summarize_pmc.py $f | grep -E "^kernel|conv3x3_sb" > $O/$name.summary.csv 2>&1
  rm -rf $O/$name
}
run set_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
run set_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
run set_c "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum"
run set_d "WRITE_SIZE SQ_INSTS_SALU SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VALU"
tail -n +1 $O/*.summary.csv | cut -c1-400
