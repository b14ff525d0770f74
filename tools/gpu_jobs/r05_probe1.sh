#!/bin/bash
# Prepared at the end of round 4 for the FIRST GPU call of the next round (a lean call: no Python, ~20 s of box time):
# the opt-in XCD-aware block order (CSEG_XCD_REMAP=1) on the kernels that do not have it yet -- the 96-channel branch layers
# (conv3x3_sb_kernel<6>), the 64-channel layer-1 kernel (conv3x3_sb16_kernel<4>), the 720-channel head (sb8) and the 720 -> 720 1x1 convolution -- A/B/A/B on one
# box, outputs compared on the hardware. If it pays like it did on the 48-channel layers (-7 %), make it the default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05p1
mkdir -p $O
P=tools/probes/conv_probe
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
[ -x $P ] || g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/conv_probe.cpp -o $P -L/opt/rocm/lib -lamdhip64 -ldl
timeout 40 $P --iters 30 --shape 8,96,64,128 --shape 8,64,128,256 --shape 8,192,32,64 \
  --variant 'plain:' --variant 'remap:CSEG_XCD_REMAP=1' --variant 'plain2:' --variant 'remap2:CSEG_XCD_REMAP=1' > $O/branches.jsonl 2> $O/err.txt
timeout 40 $P --iters 4 --nt 265 --c1 --shape 8,720,128,256 \
  --variant 'plain:' --variant 'remap:CSEG_XCD_REMAP=1' --variant 'plain2:' --variant 'remap2:CSEG_XCD_REMAP=1' > $O/head.jsonl 2>> $O/err.txt
python3 - <<'PY'
import json
for f in ("branches", "head"):
    for l in open("gpurun_out/r05p1/%s.jsonl" % f):
        d = json.loads(l)
        if "shape" in d:
            print(d["shape"][1], "%-8s st %.1f plain %.1f diff %.3g" % (d["variant"], d["fwd_st_us"], d["fwd_us"], d["max_abs_diff_vs_first"]),
                  ("c1 %.1f diff %.3g" % (d["c1_us"], d["c1_max_abs_diff_vs_first"])) if "c1_us" in d else "")
PY
tail -2 $O/err.txt
