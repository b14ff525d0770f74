#!/bin/bash
# Round 3, GPU call 5: whole-network weight packs in one launch per step, 32-pixel weight-gradient segments (384 channels),
# sparse-tail scatter without the 755 MB layout copies. Suite, bench (full line incl. cpu baseline), trace, whole-step PMC.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j5
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 > $O/gputest.log 2>&1; tail -4 $O/gputest.log | cut -c1-3000
grep -E "^(FAILED|ERROR)" $O/gputest.log | head -20
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("BENCH", d["value"], d["ms_per_step"], d["config"]["final_loss"], d["config"]["route_fallback"], d["roofline"].get("blended_roof"), d.get("cpu_baseline"))
    print("DOMINANT", d["roofline"].get("dominant_kernel"))
    for r in (d.get("split_kernels") or [])[:40]:
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
  head -6 $O/step_steady_kernel_stats.csv | cut -c1-150; cat $O/step_steady_window.txt; head -2 $O/step_steady_gaps.txt
fi
rm -rf $O/trace
# whole-step HBM-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), 2 steady steps each
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_$ctr -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/pmc_$ctr.out 2> $O/pmc_$ctr.err
  c=$(find $O/pmc_$ctr -name '*counter_collection.csv' | head -1)
  [ -n "$c" ] && python $R/tools/step_pmc_summary.py $c $ctr 2 > $O/step_pmc_$ctr.json 2> $O/step_pmc_$ctr.err
  rm -rf $O/pmc_$ctr
  cat $O/step_pmc_$ctr.json | cut -c1-600
done
