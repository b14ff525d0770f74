#!/bin/bash
# Round 4, GPU call 18: kernel trace of steady steps on the current tree (forked streams): per-kernel sums + idle gaps of the whole device.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j18
mkdir -p $O
cd /tmp
CSEG_BENCH_GUARD=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_under_rocprof.json 2> $O/trace.err
cd $R
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
MS=$(tail -1 $O/bench_under_rocprof.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
echo "under rocprof: $MS ms/step"
python tools/trace_window_stats.py $T $(python -c "print(5*$MS/1000.0)") > $O/step_steady_kernel_stats.csv 2> $O/window.txt; cat $O/window.txt
python tools/trace_gaps.py $T $(python -c "print(3*$MS/1000.0)") 30 > $O/gaps.txt; head -34 $O/gaps.txt | cut -c1-200
rm -rf $O/trace
