#!/bin/bash
# Round 3, GPU call 6 (short): persistent chunk-barrier kernel vs the one-tile kernels on the branch shapes; counters of the f16x3
# head kernels.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j6
mkdir -p $O
cd $R
timeout 300 python tools/branch_conv_probe.py > $O/branch_conv_probe.jsonl 2> $O/branch_conv_probe.err; cat $O/branch_conv_probe.jsonl; tail -3 $O/branch_conv_probe.err
timeout 200 python -m pytest tests/test_gpu_conv3x3_sb.py tests/test_zz_gpu_default_routes.py -q -x -k "not whole_step and not kernel_trace and not rccl" > $O/sb_tests.log 2>&1; tail -3 $O/sb_tests.log | cut -c1-400
bash $R/tools/r03_pmc_f16.sh > $O/pmc_f16.log 2>&1; tail -40 $O/pmc_f16.log | cut -c1-420
