#!/bin/bash
# Round 4, GPU call 9: forked branches in eager steps after the stream-safe max|.| arena: losses must agree with one stream; then the
# model / step goldens with the forks on.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j9
mkdir -p $O
cd $R
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "forks:1" "one:0" "forks_again:1"; do
  IFS=: read name st <<< "$cfg"
  CSEG_BRANCH_STREAMS=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
CSEG_BRANCH_STREAMS=1 timeout 600 python -m pytest tests/test_models_golden.py tests/test_step_golden.py tests/test_gpu_train_step.py tests/test_zz_gpu_default_routes.py -m gpu -q -x --timeout 500 > $O/tests_forks.log 2>&1; grep -E "passed|failed|Error|^E  " $O/tests_forks.log | cut -c1-400 | tail -8
