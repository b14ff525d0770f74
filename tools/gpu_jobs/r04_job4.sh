#!/bin/bash
# Round 4, GPU call 4: step graph parity (forked branches on / off, after record_stream + the batch-1 stand-in fix), bench at one image
# per GPU with and without the replay.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_step_graph.py -m gpu -q -s --timeout 300 > $O/tests.log 2>&1; grep -E "worst|passed|failed|Error|error|step graph|^E  " $O/tests.log | cut -c1-260 | tail -24
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "graph1:1:1:--global-batch 1" "nostreams1:1:0:--global-batch 1" "eager1:0:1:--global-batch 1" "graph2:1:1:--global-batch 2" "eager2:0:1:--global-batch 2"; do
  IFS=: read name g st extra <<< "$cfg"
  CSEG_STEP_GRAPH=$g CSEG_STEP_GRAPH_STREAMS=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B $extra > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["step_graph"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
