#!/bin/bash
# Round 3: identity-path gradient added in the epilogue of the backward-data kernel (BasicBlock): parity, A/B on one box, full suite.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j29
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py -q > $O/pytest_sb.log 2>&1; tail -3 $O/pytest_sb.log | cut -c1-300
for f in 0 1; do
  CSEG_BENCH_GUARD=0 CSEG_CONV3X3_FORK=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench_fork_$f.json 2> $O/bench_fork_$f.err
  python -c "
import json; d=json.loads(open('$O/bench_fork_$f.json').read().strip().splitlines()[-1]); print('fork', $f, d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x > $O/gputest.log 2>&1; tail -3 $O/gputest.log | cut -c1-600
