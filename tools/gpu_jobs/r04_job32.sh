#!/bin/bash
# Lean probe call 7 (the round's last GPU seconds): XCD-aware tile order in the 4-row persistent kernel (192 / 384 / 64 channels:
# the n_cot blocks of a tile on one XCD) against the plain order; output compared with the plain order's on the hardware.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j32
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 25 tools/probes/conv_probe --iters 30 --shape 8,192,32,64 --shape 8,384,16,32 --shape 8,64,128,256 --shape 8,48,128,256 --shape 2,192,32,64 \
  --variant 'plain:CSEG_SB16_XCD=0' --variant 'xcd:' --variant 'plain2:CSEG_SB16_XCD=0' --variant 'xcd2:' \
  > $O/xcd.jsonl 2> $O/err.txt
python3 - <<'PY'
import json
for l in open("gpurun_out/r04j32/xcd.jsonl"):
    d = json.loads(l)
    if "shape" in d:
        print(d["shape"][0], d["shape"][1], "%-8s st %.1f plain %.1f diff %.3g" % (d["variant"], d["fwd_st_us"], d["fwd_us"], d["max_abs_diff_vs_first"]))
PY
tail -2 $O/err.txt
