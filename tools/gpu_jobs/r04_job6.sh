#!/bin/bash
# Round 4, GPU call 6: the whole GPU suite on the tree with the convolution-epilogue BN statistics, then bench with the switch on / off.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/tests.log 2>&1; grep -E "passed|failed|Error|CSEG_ZZ|^E  " $O/tests.log | cut -c1-900 | tail -12
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "stats_on:1" "stats_off:0" "stats_on_again:1"; do
  IFS=: read name st <<< "$cfg"
  CSEG_CONV_STATS=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
