#!/bin/bash
# Round 4, GPU call 17: heads forked (projection head under the classifier head's 3x3 convolution) A/B, stream / golden tests.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j17
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_streams.py tests/test_models_golden.py tests/test_gpu_conv3x3_sb.py tests/test_gpu_sparse_embed.py -m gpu -q -x --timeout 400 > $O/tests.log 2>&1; grep -E "passed|failed|Error|Fatal|^E  " $O/tests.log | cut -c1-400 | tail -5
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "headfork:1" "nofork:0" "headfork_again:1" "nofork_again:0"; do
  IFS=: read name st <<< "$cfg"
  CSEG_HEAD_FORK=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
