#!/bin/bash
# Round 4, GPU call 21 (closing): smoke, the whole GPU suite on the closing tree, HBM traffic of one steady step (FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 passes), a short bench line.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j21
mkdir -p $O
cd $R
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/tests.log 2>&1; grep -E "passed|failed|Error|Fatal|CSEG_ZZ|^FAILED" $O/tests.log | cut -c1-1200 | tail -10
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  CSEG_BENCH_GUARD=0 timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_$ctr -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/pmc_$ctr.out 2> $O/pmc_$ctr.err
  c=$(find $O/pmc_$ctr -name '*counter_collection.csv' | head -1)
  [ -n "$c" ] && python $R/tools/step_pmc_summary.py $c $ctr 2 > $O/step_pmc_$ctr.json 2> $O/step_pmc_$ctr.err
  rm -rf $O/pmc_$ctr
  cut -c1-300 $O/step_pmc_$ctr.json
done
cd $R
CSEG_BENCH_GUARD=0 timeout 200 python bench.py --no-kernels --no-cpu-baseline --no-fp32-pass --steps 10 --warmup 3 > $O/bench_short.log 2> $O/bench_short.err; tail -1 $O/bench_short.log | cut -c1-700
