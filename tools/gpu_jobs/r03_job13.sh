#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j13
mkdir -p $O
cd $R
timeout 300 python tools/wrw_split_probe.py > $O/wrw_split_probe.jsonl 2> $O/wrw_split_probe.err; cat $O/wrw_split_probe.jsonl; tail -3 $O/wrw_split_probe.err
timeout 200 python tools/host_profile.py 8 > $O/host_profile_b8.txt 2> $O/host_profile_b8.err; head -1 $O/host_profile_b8.txt
timeout 200 python tools/host_profile.py 1 > $O/host_profile_b1.txt 2> $O/host_profile_b1.err; head -1 $O/host_profile_b1.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
