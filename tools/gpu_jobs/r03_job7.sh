#!/bin/bash
# Round 3, GPU call 7 (short): conflict-free LDS strides (real ds_read_b128 lane groups), measured kernel routing.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j7
mkdir -p $O
cd $R
timeout 300 python tools/split_arith_probe.py > $O/split_arith_probe.jsonl 2> $O/split_arith_probe.err
python - <<PY
import json
for l in open("$O/split_arith_probe.jsonl"):
    d = json.loads(l)
    if d["op"].startswith("conv3x3"):
        print(d["op"], d["shape"], "f16x3", d["f16x3"]["us"], "bf16x6", d["bf16x6"]["us"], "miopen", d["miopen_fp32"]["us"], "err", d["f16x3"]["max_err_vs_fp64"])
    else:
        print(d["op"], d["shape"], d["f16x3"], d["torch_fp32"])
PY
tail -2 $O/split_arith_probe.err
timeout 200 python -m pytest tests/test_gpu_conv3x3_sb.py -q > $O/sb_tests.log 2>&1; tail -2 $O/sb_tests.log | cut -c1-300
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("BENCH", d["value"], d["ms_per_step"], d["config"]["final_loss"], d["roofline"].get("blended_roof", {}).get("frac"))
    for r in (d.get("split_kernels") or [])[:16]:
        print("  ", r["ms_per_step"], r["calls_per_step"], r["us_per_launch"], r["frac"], r["kernel"])
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench_default.err").read()[-2500:])
PY
