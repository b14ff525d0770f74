#!/bin/bash
# Round 4, GPU call 1: the ADVICE fixes on hardware (fused-SGD packs, SyncBN count row), the new parity tests (20 steps default vs
# strict fp32, step goldens through the fused-SGD factory, ResNet-101 step), the short bench line, and the hipGraph capture probe.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j1
mkdir -p $O
cd $R
timeout 200 python tools/graph_probe.py fwd two one relaxed > $O/graph_probe.log 2>&1; cat $O/graph_probe.log | cut -c1-600
timeout 400 python -m pytest tests/test_cabi.py tests/test_gpu_bn.py tests/test_gpu_train_step.py::test_twenty_steps_default_arithmetic_tracks_strict_fp32 tests/test_step_golden.py -m gpu -q -x -s --timeout 300 > $O/tests.log 2>&1; grep -E "20-step|passed|failed|Error|error" $O/tests.log | cut -c1-400 | tail -12
timeout 400 python bench.py > $O/bench_stdout.log 2> $O/bench_stderr.log; tail -c 2500 $O/bench_stdout.log; echo; tail -3 $O/bench_stderr.log | cut -c1-300
cp bench_detail.json $O/ 2>/dev/null
