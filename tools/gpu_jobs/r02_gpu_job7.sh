#!/bin/bash
# Round 2, closing call: steady-state kernel trace of the default step (split-bf16 convolutions on), then the multi-rank
# tests on the same default as far as the budget reaches.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02u
mkdir -p $O
cd $R
( cd /tmp && timeout 100 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- \
    python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_traced.json 2> $O/bench_traced.err )
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.load(open('$O/bench_traced.json'))['ms_per_step'])")
  win=$(python -c "print(5*$ms/1000.0)")
  python tools/trace_window_stats.py $f $win > $O/step_steady_kernel_stats.csv 2> $O/step_steady_window.txt
  python tools/trace_gaps.py $f $win 8 > $O/step_steady_gaps.txt 2>&1
fi
rm -rf $O/kt
cut -c1-260 $O/bench_traced.json
head -8 $O/step_steady_kernel_stats.csv | cut -c1-200
timeout 60 python -m pytest tests/test_gpu_multirank.py -q -x -k "not bench" > $O/t.log 2>&1
tail -3 $O/t.log | cut -c1-300
