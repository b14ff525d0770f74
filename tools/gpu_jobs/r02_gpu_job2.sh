#!/bin/bash
# Round-2 GPU job: GPU tests, steady-state kernel trace of the benched step, whole-step PMC passes, default bench line.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 -s > $O/gputest.log 2>&1
tail -4 $O/gputest.log
# steady-state kernel trace
( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- \
    python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernels > $O/bench_traced.json 2> $O/bench_traced.err )
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.load(open('$O/bench_traced.json'))['ms_per_step'])")
  win=$(python -c "print(5*$ms/1000.0)")
  python tools/trace_window_stats.py $f $win > $O/step_steady_kernel_stats.csv 2> $O/step_steady_window.txt
  python tools/trace_gaps.py $f $win 12 > $O/step_steady_gaps.txt 2>&1
fi
rm -rf $O/kt
# whole-step PMC passes (separate runs, kernel-trace only besides the counter)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 1200 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernels > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err )
  f=$(find $O/pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/step_pmc_summary.py $f $c 2 > $O/step_pmc_$c.json 2> $O/step_pmc_$c.err
  rm -rf $O/pmc_$c
done
# default bench line (N=1, cpu baseline + per-kernel rooflines)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-400 $O/bench_default.json
ls -la $O
