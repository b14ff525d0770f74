#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j15
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py tests/test_gpu_conv3x3_s2.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/wrw_split_probe.py > $O/wrw_split_probe.jsonl 2> $O/wrw_split_probe.err; cat $O/wrw_split_probe.jsonl; tail -3 $O/wrw_split_probe.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
