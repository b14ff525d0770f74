#!/bin/bash
# Round-2 final GPU job: all GPU tests, default bench line, steady-state kernel trace, other workloads, per-GPU batch 1.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 > $O/gputest.log 2>&1
tail -4 $O/gputest.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-300 $O/bench_default.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- \
    python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernels > $O/bench_traced.json 2> $O/bench_traced.err )
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.load(open('$O/bench_traced.json'))['ms_per_step'])")
  win=$(python -c "print(5*$ms/1000.0)")
  python tools/trace_window_stats.py $f $win > $O/step_steady_kernel_stats.csv 2> $O/step_steady_window.txt
  python tools/trace_gaps.py $f $win 8 > $O/step_steady_gaps.txt 2>&1
fi
rm -rf $O/kt
# per-GPU batch 1 (what one rank of the 8-GPU strong-scaling run does, minus the collectives): GPU-busy fraction
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/kt1 -o kt1 --output-format csv -- \
    python $R/bench.py --global-batch 1 --steps 12 --warmup 3 --no-cpu-baseline --no-kernels > $O/bench_b1.json 2> $O/bench_b1.err )
f=$(find $O/kt1 -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.load(open('$O/bench_b1.json'))['ms_per_step'])")
  win=$(python -c "print(10*$ms/1000.0)")
  python tools/trace_gaps.py $f $win 5 > $O/b1_gaps.txt 2>&1
fi
rm -rf $O/kt1
for wl in cfg4 cfg4mem cfg5; do
  timeout 900 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-kernels > $O/bench_$wl.json 2> $O/bench_$wl.err
  cut -c1-200 $O/bench_$wl.json
done
ls $O
