#!/bin/bash
# Round 4, GPU call 20: the persistent branch kernel with precomputed staging descriptors (isolated timings), bench at bs 8 / 2.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j20
mkdir -p $O
cd $R
timeout 200 python tools/branch_conv_probe.py > $O/branch_probe.jsonl 2> $O/branch_probe.err; grep default $O/branch_probe.jsonl | cut -c1-300
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "b8:" "b8_again:" "b2:--global-batch 2"; do
  IFS=: read name extra <<< "$cfg"
  CSEG_STEP_GRAPH=0 CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B $extra > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_$name.err | tail -2 | cut -c1-300
done
