#!/bin/bash
# Round 2, last GPU call: split-bf16 3x3 convolution -- parity against fp64 and timing against MIOpen fp32.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02s
CSEG_TEST_SPLIT_BF16=1 timeout 140 python -m pytest tests/test_gpu_conv3x3_sb.py -q > gpurun_out/r02s/t.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02s/t.log
timeout 150 python tools/conv3x3_sb_probe.py head_720 branch_48 branch_96 > gpurun_out/r02s/probe.jsonl 2> gpurun_out/r02s/probe.err
echo "probe rc=$?" >> gpurun_out/r02s/t.log
tail -25 gpurun_out/r02s/t.log | cut -c1-300
cat gpurun_out/r02s/probe.jsonl
tail -3 gpurun_out/r02s/probe.err
