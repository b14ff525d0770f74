#!/bin/bash
# Round 3, last GPU call: the multi-rank paths after fused_bn.check_equal_counts (one-rank RCCL check of test_zz, the 2-rank tests).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j30
mkdir -p $O
cd $R
timeout 240 python -m pytest tests/test_gpu_multirank.py -m gpu -q --timeout 200 > $O/multirank.log 2>&1; tail -3 $O/multirank.log | cut -c1-300
timeout 120 python tools/rccl_single_rank_check.py > $O/rccl1.log 2>&1; tail -3 $O/rccl1.log | cut -c1-300
