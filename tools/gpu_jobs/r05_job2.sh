#!/bin/bash
# Prepared at the end of round 4 for an early Python call of the next round: what the host-side work of the closing hours is worth
# on the MI355X -- (a) the step as it is (max|.| records and packed-weight records without per-call tensor indexing: default),
# (c) CSEG_FANOUT_SUM=1 (one gradient sum per branch of an exchange unit instead of autograd's add launches),
# (b) + the native residual-block executor (CSEG_NATIVE_BLOCK=1: one C++ call per block and direction instead of 6-10 Python-wrapped
# launches), A/B/A/B on one box; the GPU tests of the executor first (bit-identity with the Python node on hardware).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05j2
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py tests/test_zz_gpu_default_routes.py -m gpu -q -k "native or eight_row or basic_block" --timeout 200 > $O/tests.log 2>&1; timeout 150 python tools/native_block_check.py
grep -E "passed|failed|Error|^FAILED" $O/tests.log | tail -3
for r in 1 2; do
  for nb in 0 1; do
    CSEG_NATIVE_BLOCK=$nb CSEG_BENCH_GUARD=0 timeout 200 python bench.py --no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4 \
      > $O/bench_nb${nb}_$r.log 2> $O/bench_nb${nb}_$r.err
    echo "native_block=$nb run $r: $(tail -1 $O/bench_nb${nb}_$r.log | cut -c1-220)"
  done
done
for r in 1 2; do
  CSEG_FANOUT_SUM=1 CSEG_BENCH_GUARD=0 timeout 200 python bench.py --no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4 \
    > $O/bench_fan_$r.log 2> $O/bench_fan_$r.err
  echo "fanout_sum=1 run $r: $(tail -1 $O/bench_fan_$r.log | cut -c1-220)"
done
CSEG_NATIVE_BLOCK=1 timeout 200 python tools/host_profile.py 8 > $O/host_profile_native.txt 2>&1; head -2 $O/host_profile_native.txt
timeout 200 python tools/host_profile.py 8 > $O/host_profile_default.txt 2>&1; head -2 $O/host_profile_default.txt
