#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j21
mkdir -p $O
cd $R
timeout 300 python tools/sb8_probe.py > $O/sb8_probe.jsonl 2> $O/sb8_probe.err; cat $O/sb8_probe.jsonl; tail -3 $O/sb8_probe.err
timeout 400 python -m pytest tests/test_gpu_conv3x3_sb.py -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
