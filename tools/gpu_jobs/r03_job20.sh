#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j20
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv3x3_s2.py tests/test_gpu_conv3x3_sb.py tests/test_models_golden.py tests/test_step_golden.py tests/test_zz_gpu_default_routes.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-1200
grep -E "^(FAILED|ERROR)" $O/pytest.log | head
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
