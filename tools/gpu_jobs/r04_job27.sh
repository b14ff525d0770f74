#!/bin/bash
# Lean probe call 3: the 8-row-tile kernel against the defaults (which now carry the 16-instruction split and the unmasked
# statistics pass) at the branch shapes, weight gradient alongside.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j27
mkdir -p $O
P=tools/probes/conv_probe
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 100 $P --iters 30 --wrw --shape 8,48,128,256 --shape 8,96,64,128 --shape 8,192,32,64 --shape 8,384,16,32 --shape 2,48,128,256 \
  --variant 'default:' --variant 'rows8:CSEG_SB16_ROWS8=1' --variant 'rows8_force:CSEG_SB16_ROWS8=2' \
  --variant 'sb16ch:CSEG_CONV3X3_SB16_CH=48,96,192,384' --variant 'sb16ch_rows8:CSEG_CONV3X3_SB16_CH=48,96,192,384;CSEG_SB16_ROWS8=1' \
  --variant 'sb16ch_rows8_force:CSEG_CONV3X3_SB16_CH=48,96,192,384;CSEG_SB16_ROWS8=2' \
  > $O/fwd.jsonl 2> $O/fwd.err
echo "rc $?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r04j27/fwd.jsonl"):
    d = json.loads(l)
    if "shape" in d:
        print(d["shape"][0], d["shape"][1], "%-20s st %.1f plain %.1f wrw %.1f diff %.3g" % (d["variant"], d["fwd_st_us"], d["fwd_us"], d["wrw_us"], d["max_abs_diff_vs_first"]))
    else:
        print(d)
PY
tail -3 $O/fwd.err
