#!/bin/bash
# Round 4, GPU call 7: convolution-epilogue BN statistics with the block-per-channel finalize: A/B bench + kernel trace of steady steps.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j7
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py tests/test_gpu_bn.py tests/test_gpu_aug.py -m gpu -q -x --timeout 250 > $O/tests.log 2>&1; tail -2 $O/tests.log | cut -c1-300
B="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3"
for cfg in "stats_on:1" "stats_off:0" "stats_on_again:1" "stats_off_again:0"; do
  IFS=: read name st <<< "$cfg"
  CSEG_CONV_STATS=$st CSEG_BENCH_GUARD=0 timeout 200 python bench.py $B > $O/bench_$name.log 2> $O/bench_$name.err
  echo "$name: $(tail -1 $O/bench_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])' 2>&1 | tail -1)"
done
cd /tmp
CSEG_BENCH_GUARD=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_under_rocprof.json 2> $O/trace.err
cd $R
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
MS=$(tail -1 $O/bench_under_rocprof.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
echo "under rocprof: $MS ms/step; trace $T"
python tools/trace_window_stats.py $T $(python -c "print(5*$MS/1000.0)") > $O/step_steady_kernel_stats.csv 2> $O/window.txt; cat $O/window.txt
head -30 $O/step_steady_kernel_stats.csv | cut -c1-160
rm -rf $O/trace
