#!/bin/bash
# Round 4, GPU call 19: measurement pass on the current tree -- the full default bench line (per-kernel tables, cpu_baseline),
# one image per GPU eager vs replay, and the other BASELINE workloads (cfg4 / cfg4mem / cfg5).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j19
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_default_full.log 2> $O/bench_default_full.err; tail -1 $O/bench_default_full.log | cut -c1-2100; cp bench_detail.json $O/bench_detail_default.json
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "b1_auto:auto:--global-batch 1" "b1_eager:0:--global-batch 1" "b2_eager:0:--global-batch 2" "cfg4:0:--workload cfg4" "cfg4mem:0:--workload cfg4mem" "cfg5:0:--workload cfg5"; do
  IFS=: read name g extra <<< "$cfg"
  CSEG_STEP_GRAPH=$g CSEG_BENCH_GUARD=0 timeout 300 python bench.py $B $extra > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["step_graph"], d["config"]["final_loss"], d["roofline"]["frac"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
