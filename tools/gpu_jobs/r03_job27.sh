#!/bin/bash
# Round 3, closing GPU call: the full GPU suite, the full bench line and the host profile on the last tree (fused SGD, cheap weight-pack
# refresh, 1x1 weight gradient from 16 channels, weight-gradient row image with an 8-pixel lead).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j27
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/gputest.log 2>&1; tail -4 $O/gputest.log | cut -c1-1500
grep -E "^(FAILED|ERROR)" $O/gputest.log | head -20
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json; tail -3 $O/bench_default.err
timeout 200 python tools/host_profile.py 8 > $O/host_profile_b8.txt 2> $O/host_profile_b8.err; head -1 $O/host_profile_b8.txt
timeout 200 python tools/host_profile.py 1 > $O/host_profile_b1.txt 2> $O/host_profile_b1.err; head -1 $O/host_profile_b1.txt
