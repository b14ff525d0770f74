#!/bin/bash
# Round 4, GPU call 11: step-graph tests (segfault in capture_end inside the full suite of call 10), stream tests, wgrad A/B.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j11
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_step_graph.py -m gpu -q -x --timeout 400 > $O/tests_graph.log 2>&1; grep -E "passed|failed|Error|Fatal|^E  " $O/tests_graph.log | cut -c1-300 | tail -5
timeout 500 python -m pytest tests/test_gpu_streams.py -m gpu -q -x -s --timeout 400 > $O/tests_streams.log 2>&1; grep -E "single stream|passed|failed|Error|Fatal|^E  " $O/tests_streams.log | cut -c1-600 | tail -6
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "wgrad:1" "nowgrad:0" "wgrad_again:1" "nowgrad_again:0"; do
  IFS=: read name st <<< "$cfg"
  CSEG_WGRAD_STREAM=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
