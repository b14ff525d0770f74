#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j25
mkdir -p $O
cd $R
for mt in 256 96 32; do
  CSEG_BENCH_GUARD=0 CSEG_SB_MIN_TILES=$mt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench_mintiles_$mt.json 2> $O/bench_mintiles_$mt.err
  python -c "
import json; d=json.loads(open('$O/bench_mintiles_$mt.json').read().strip().splitlines()[-1]); print('min_tiles', $mt, d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
