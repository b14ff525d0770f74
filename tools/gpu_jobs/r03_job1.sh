#!/bin/bash
# Round 3, GPU call 1: the tree with the round-2 driver-pass winners as defaults (weight gradient v2, 3x3 variants per shape,
# 1x1 split kernels, row-sparse embedding gradient). Produces: f16 MFMA hardware facts, the full GPU suite, a bench line,
# the kernel trace of the benched build, the counter passes on the split kernels.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j1
mkdir -p $O
cd $R
timeout 60 tools/probes/mfma_f16_probe > $O/mfma_f16_probe.jsonl 2> $O/mfma_f16_probe.err; cat $O/mfma_f16_probe.jsonl
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > $O/gputest.log 2>&1; tail -4 $O/gputest.log | cut -c1-2500
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("BENCH", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["route_fallback"], d.get("fp32_conv_path"))
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench_default.err").read()[-1500:])
PY
# kernel trace of the benched build (steady-state window = the last 5 of 6 timed steps)
cd /tmp
export CSEG_BENCH_GUARD=0
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_under_rocprof.json 2> $O/trace.err
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then
  ms=$(python -c "import json;print(json.loads(open('$O/bench_under_rocprof.json').read().strip().splitlines()[-1])['ms_per_step'])")
  python $R/tools/trace_window_stats.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_kernel_stats.csv 2> $O/step_steady_window.txt
  python $R/tools/trace_gaps.py $f $(python -c "print(5*$ms/1000.0)") > $O/step_steady_gaps.txt 2>&1
  head -25 $O/step_steady_kernel_stats.csv | cut -c1-160; cat $O/step_steady_window.txt $O/step_steady_gaps.txt | head -12
fi
rm -rf $O/trace
# counters on the split kernels at the benched shapes
bash $R/tools/r03_pmc_sb.sh > $O/pmc_sb.log 2>&1; tail -60 $O/pmc_sb.log | cut -c1-200
