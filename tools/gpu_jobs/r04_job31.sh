#!/bin/bash
# Lean probe call 6: XCD-contiguous tile order of the 8-row kernel (halos of neighbouring tiles in one L2) against the plain order.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j31
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 30 tools/probes/conv_probe --iters 40 --shape 8,48,128,256 --shape 4,48,128,256 \
  --variant 'plain:' --variant 'xcd:CSEG_SB16_XCD=1' --variant 'plain2:' --variant 'xcd2:CSEG_SB16_XCD=1' --variant 'rows4:CSEG_SB16_ROWS8=0' \
  > $O/xcd.jsonl 2> $O/err.txt
python3 - <<'PY'
import json
for l in open("gpurun_out/r04j31/xcd.jsonl"):
    d = json.loads(l)
    if "shape" in d:
        print(d["shape"][0], d["shape"][1], "%-8s st %.1f plain %.1f diff %.3g" % (d["variant"], d["fwd_st_us"], d["fwd_us"], d["max_abs_diff_vs_first"]))
PY
tail -2 $O/err.txt
