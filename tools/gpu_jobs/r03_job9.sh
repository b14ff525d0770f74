#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j9
mkdir -p $O
cd $R
timeout 300 python tools/sps_probe.py > $O/sps_probe.jsonl 2> $O/sps_probe.err; cat $O/sps_probe.jsonl; tail -3 $O/sps_probe.err
timeout 400 python -m pytest tests/test_gpu_conv3x3_sb.py tests/test_zz_gpu_default_routes.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-600
