#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j14
mkdir -p $O
cd $R
timeout 300 python tools/stats_grad_probe.py > $O/probe.log 2>&1; grep -v "amdgpu.ids\|Warning\|warn" $O/probe.log | tail -30 | cut -c1-1200
