#!/bin/bash
# Round 3, last GPU call: the full GPU suite and the bench line on the final tree.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j22
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/gputest.log 2>&1; tail -4 $O/gputest.log | cut -c1-1500
grep -E "^(FAILED|ERROR)" $O/gputest.log | head -20
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
