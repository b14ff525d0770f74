#!/bin/bash
# Round 4, GPU call 24: last sanity on the closing tree (stream tests, train-step tests, short bench).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j24
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_streams.py tests/test_gpu_train_step.py tests/test_gpu_step_graph.py -m gpu -q --timeout 300 > $O/tests.log 2>&1; grep -E "passed|failed|Error|Fatal|^FAILED" $O/tests.log | cut -c1-300 | tail -5
CSEG_BENCH_GUARD=0 timeout 200 python bench.py --no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3 > $O/bench_short.log 2> $O/bench_short.err; tail -1 $O/bench_short.log | cut -c1-300
