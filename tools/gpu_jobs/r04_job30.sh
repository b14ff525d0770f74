#!/bin/bash
# Lean probe call 5: two branch convolutions side by side on two streams, each on half (a quarter) of the CUs, against the same two
# launches with 256 blocks each -- does partitioning the chip between independent branches overlap their latency-bound phases?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j30
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 40 tools/probes/conv_probe --iters 30 --pair --shape 8,48,128,256 --shape 8,192,32,64 --shape 8,384,16,32 \
  --variant 'cus256:' --variant 'cus128:CSEG_PERSIST_CUS=128' --variant 'cus64:CSEG_PERSIST_CUS=64' --variant 'cus192:CSEG_PERSIST_CUS=192' \
  > $O/pair.jsonl 2> $O/err.txt
python3 - <<'PY'
import json
for l in open("gpurun_out/r04j30/pair.jsonl"):
    d = json.loads(l)
    if "shape" in d:
        print(d["shape"][1], "%-8s single %.1f  pair %.1f  diff %.3g" % (d["variant"], d["fwd_us"], d["pair_us"], d["max_abs_diff_vs_first"]))
PY
tail -2 $O/err.txt
