#!/bin/bash
# Round 4, GPU call 23: the compute stream at HIGH priority (the forked side streams stay at the default): A/B/A.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j23
mkdir -p $O
cd $R
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "hi:-1" "def:" "hi2:-1" "def2:"; do
  IFS=: read name pr <<< "$cfg"
  CSEG_MAIN_PRIORITY=$pr CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -1 | cut -c1-200
done
