#!/bin/bash
# Round 4, GPU call 16: full GPU suite on the tree with the fused residual-block node + lean host path, then bench A/B.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j16
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/tests.log 2>&1; grep -E "passed|failed|Error|Fatal|CSEG_ZZ|^E  " $O/tests.log | cut -c1-900 | tail -8
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "fused:1" "unfused:0" "fused_again:1" "unfused_again:0"; do
  IFS=: read name st <<< "$cfg"
  CSEG_BLOCK_FUSED=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
CSEG_STEP_GRAPH=0 timeout 200 python tools/host_profile.py 8 2>&1 | grep "host enqueue"
