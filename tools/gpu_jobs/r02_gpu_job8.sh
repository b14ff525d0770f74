#!/bin/bash
# Round 2, last seconds of GPU budget: first hardware run of the split-bf16 weight-gradient kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02v
CSEG_TEST_SB_WRW=1 timeout 25 python -m pytest tests/test_gpu_conv3x3_sb.py -q -k weight_gradient > gpurun_out/r02v/t.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02v/t.log
tail -6 gpurun_out/r02v/t.log | cut -c1-300
timeout 20 python tools/conv3x3_sb_wrw_probe.py branch_48 branch_96 head_720 > gpurun_out/r02v/probe.jsonl 2> gpurun_out/r02v/probe.err
echo "probe rc=$?"
cat gpurun_out/r02v/probe.jsonl
