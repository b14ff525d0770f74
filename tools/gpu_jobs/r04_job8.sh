#!/bin/bash
# Round 4, GPU call 8: forked HRNet branches in eager steps (four queues fed by one host thread) vs one stream.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j8
mkdir -p $O
cd $R
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "forks:1" "one:0" "forks_again:1" "one_again:0"; do
  IFS=: read name st <<< "$cfg"
  CSEG_BRANCH_STREAMS=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
