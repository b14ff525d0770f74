#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j16
mkdir -p $O
cd $R
timeout 200 python tools/wrw_debug.py > $O/wrw_debug.txt 2>&1; cat $O/wrw_debug.txt | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py -q -k "weight_gradient" > $O/pytest.log 2>&1; tail -8 $O/pytest.log | cut -c1-200
