#!/bin/bash
# Round 3, GPU call 3: f16x3 with max|.| produced by the fused-BN apply kernels (no extra pass per convolution operand):
# per-shape probe of both arithmetics, full GPU suite, bench, trace, counters of the f16x3 kernels.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j3
mkdir -p $O
cd $R
timeout 400 python tools/split_arith_probe.py > $O/split_arith_probe.jsonl 2> $O/split_arith_probe.err; cat $O/split_arith_probe.jsonl | cut -c1-700; tail -3 $O/split_arith_probe.err
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/gputest.log 2>&1; tail -6 $O/gputest.log | cut -c1-3000
grep -E "^(FAILED|ERROR)" $O/gputest.log | head -20
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python -c "import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print('BENCH', d['value'], d['ms_per_step'], d['config']['final_loss'], d['config']['route_fallback'], d.get('fp32_conv_path'))" || tail -5 $O/bench_default.err
cd /tmp
export CSEG_BENCH_GUARD=0
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_under_rocprof.json 2> $O/trace.err
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.loads(open('$O/bench_under_rocprof.json').read().strip().splitlines()[-1])['ms_per_step'])")
  python $R/tools/trace_window_stats.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_kernel_stats.csv 2> $O/step_steady_window.txt
  python $R/tools/trace_gaps.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_gaps.txt 2>&1
  head -12 $O/step_steady_kernel_stats.csv | cut -c1-150; cat $O/step_steady_window.txt; head -3 $O/step_steady_gaps.txt
fi
rm -rf $O/trace
