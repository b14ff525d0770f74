#!/bin/bash
# Round-2 closing GPU job: all GPU tests, whole-step PMC passes at the final build, default bench line.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 > $O/gputest.log 2>&1
tail -3 $O/gputest.log
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernels > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err )
  f=$(find $O/pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/step_pmc_summary.py $f $c 2 > $O/step_pmc_$c.json 2> $O/step_pmc_$c.err
  rm -rf $O/pmc_$c
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-300 $O/bench_default.json
