#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j26
mkdir -p $O
cd $R
for f in 0 1 0 1; do
  CSEG_BENCH_GUARD=0 CSEG_FUSED_SGD=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench_fused_$f.json 2> $O/bench_fused_$f.err
  python -c "
import json; d=json.loads(open('$O/bench_fused_$f.json').read().strip().splitlines()[-1]); print('fused_sgd', $f, d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
CSEG_FUSED_SGD=1 timeout 600 python -m pytest tests/test_step_golden.py -m gpu -q > $O/step_golden_fused.log 2>&1; tail -3 $O/step_golden_fused.log | cut -c1-200
