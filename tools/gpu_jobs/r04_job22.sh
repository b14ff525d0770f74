#!/bin/bash
# Round 4, GPU call 22: HIP stream priorities for the forked side streams (the 48-channel branch on the compute stream is the longest chain).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j22
mkdir -p $O
cd $R
python -c "import torch; print('priority range (least, greatest):', torch.cuda.Stream.priority_range())" 2>&1 | tail -1
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for pr in 0 1 -1 0 1; do
  CSEG_FORK_PRIORITY=$pr CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_p$pr.log 2> $O/bench_p$pr.err
  echo "priority $pr: $(tail -1 $O/bench_p$pr.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
  grep -v "amdgpu.ids\|UserWarning\|run_backward" $O/bench_p$pr.err | tail -1 | cut -c1-200
done
