#!/bin/bash
# Round 3, GPU call 4: max|.| records sharded over 32 cache lines (the single-word atomics doubled the BN apply kernels),
# 384- and 64-channel 3x3 convolutions on the split kernels. Suite, bench, trace (+ who launches the 0.9 ms copy kernels).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/gputest.log 2>&1; tail -4 $O/gputest.log | cut -c1-3000
grep -E "^(FAILED|ERROR)" $O/gputest.log | head -20
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("BENCH", d["value"], d["ms_per_step"], d["config"]["final_loss"], d["config"]["route_fallback"], d["roofline"].get("blended_roof"))
    print("DOMINANT", d["roofline"].get("dominant_kernel"))
    for r in (d.get("split_kernels") or [])[:14]:
        print("  ", r["ms_per_step"], r["calls_per_step"], r["us_per_launch"], r["frac"], r["kernel"])
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench_default.err").read()[-2500:])
PY
cd /tmp
export CSEG_BENCH_GUARD=0
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_under_rocprof.json 2> $O/trace.err
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.loads(open('$O/bench_under_rocprof.json').read().strip().splitlines()[-1])['ms_per_step'])")
  python $R/tools/trace_window_stats.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_kernel_stats.csv 2> $O/step_steady_window.txt
  python $R/tools/trace_gaps.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_gaps.txt 2>&1
  python $R/tools/trace_neighbors.py $f $(python -c "print(1.2*$ms/1000.0)") direct_copy 300 4 > $O/copy_neighbors.txt 2>&1
  python $R/tools/trace_neighbors.py $f $(python -c "print(1.2*$ms/1000.0)") CUDAFunctor_add 60 2 > $O/add_neighbors.txt 2>&1
  head -8 $O/step_steady_kernel_stats.csv | cut -c1-150; cat $O/step_steady_window.txt; head -2 $O/step_steady_gaps.txt; cat $O/copy_neighbors.txt | cut -c1-200
fi
rm -rf $O/trace
