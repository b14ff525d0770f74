#!/bin/bash
# Round 4, GPU call 3: the step graph inside the trainer -- parity against eager steps for every model family, then bench with and
# without it at batch 8 and at one image per GPU.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j3
mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_gpu_step_graph.py -m gpu -q -x -s --timeout 300 > $O/tests.log 2>&1; grep -E "worst|passed|failed|Error|error|step graph" $O/tests.log | cut -c1-300 | tail -14
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "graph8:1:1:" "eager8:0:1:" "nostreams8:1:0:" "graph1:1:1:--global-batch 1" "eager1:0:1:--global-batch 1"; do
  IFS=: read name g st extra <<< "$cfg"
  CSEG_STEP_GRAPH=$g CSEG_STEP_GRAPH_STREAMS=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B $extra > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["step_graph"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  tail -2 $O/bench_$name.err | cut -c1-300
done
