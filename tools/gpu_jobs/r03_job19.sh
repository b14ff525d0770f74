#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j19
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_conv3x3_sb.py tests/test_gpu_conv3x3_s2.py -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-200
timeout 200 python tools/c1_probe.py > $O/c1_probe.jsonl 2> $O/c1_probe.err; cat $O/c1_probe.jsonl; tail -2 $O/c1_probe.err
timeout 300 python tools/s2_probe.py > $O/s2_probe.jsonl 2> $O/s2_probe.err; python - <<PY
import json
for l in open("$O/s2_probe.jsonl"):
    d = json.loads(l); print(d["shape"], "wrw", d["wrw_us"], d["wrw_dev"])
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-pass --no-kernels > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
