#!/bin/bash
# Round 3, GPU call 2: first hardware run of the f16x3 arithmetic (all split kernels) -- kernel parity in both arithmetics,
# per-shape timings bf16x6 vs f16x3 vs MIOpen, the full GPU suite on the new default, bench lines for both arithmetics, trace.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j2
mkdir -p $O
cd $R
export CSEG_TEST_SB_WRW_V2=1 CSEG_TEST_SB_1X1=1 CSEG_TEST_SB_NT=1 CSEG_TEST_SB_WRW=1
for ar in f16x3 bf16x6; do
  CSEG_SPLIT_ARITH=$ar timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py -q > $O/sb_kernels_$ar.log 2>&1; echo "$ar: $(tail -1 $O/sb_kernels_$ar.log | cut -c1-200)"
done
grep -E "^(FAILED|ERROR)|Error|assert" $O/sb_kernels_f16x3.log | head -20 | cut -c1-300
timeout 300 python tools/split_arith_probe.py > $O/split_arith_probe.jsonl 2> $O/split_arith_probe.err; cat $O/split_arith_probe.jsonl; tail -3 $O/split_arith_probe.err
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > $O/gputest.log 2>&1; tail -4 $O/gputest.log | cut -c1-2500
for ar in f16x3 bf16x6; do
  CSEG_SPLIT_ARITH=$ar timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_$ar.json 2> $O/bench_$ar.err
  python -c "import json;d=json.loads(open('$O/bench_$ar.json').read().strip().splitlines()[-1]);print('BENCH $ar', d['value'], d['ms_per_step'], d['config']['final_loss'], d['config']['route_fallback'])" || tail -5 $O/bench_$ar.err
done
cd /tmp
export CSEG_BENCH_GUARD=0
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_under_rocprof.json 2> $O/trace.err
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.loads(open('$O/bench_under_rocprof.json').read().strip().splitlines()[-1])['ms_per_step'])")
  python $R/tools/trace_window_stats.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_kernel_stats.csv 2> $O/step_steady_window.txt
  python $R/tools/trace_gaps.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_gaps.txt 2>&1
  head -16 $O/step_steady_kernel_stats.csv | cut -c1-150; cat $O/step_steady_window.txt; head -3 $O/step_steady_gaps.txt
fi
rm -rf $O/trace
