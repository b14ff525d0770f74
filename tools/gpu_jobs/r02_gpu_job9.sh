#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02w
timeout 12 python tools/conv3x3_sb_probe.py branch_192 > gpurun_out/r02w/probe_192_nt6.jsonl 2>/dev/null
CSEG_CONV3X3_SB_NT=3 timeout 12 python tools/conv3x3_sb_probe.py branch_192 branch_96 > gpurun_out/r02w/probe_nt3.jsonl 2>/dev/null
grep -h "glds=1\|miopen" gpurun_out/r02w/*.jsonl | cut -c1-200
