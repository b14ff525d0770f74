#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j23
mkdir -p $O
cd $R
for mc in 256 48 16; do
  CSEG_BENCH_GUARD=0 CSEG_CONV1X1_SB_WRW_MIN_CH=$mc timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench_minch_$mc.json 2> $O/bench_minch_$mc.err
  python -c "
import json; d=json.loads(open('$O/bench_minch_$mc.json').read().strip().splitlines()[-1]); print('min_ch', $mc, d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
