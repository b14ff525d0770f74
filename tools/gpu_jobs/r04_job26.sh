#!/bin/bash
# Lean probe call 2: launch floor, the 16-instruction split and the unmasked statistics pass on the opt-in kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j26
mkdir -p $O
P=tools/probes/conv_probe
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 150 $P --iters 30 --shape 8,48,128,256 --shape 8,192,32,64 --shape 8,384,16,32 --shape 1,48,4,64 \
  --variant 'default:' --variant 'q:CSEG_SB16_PF=1' --variant 'q_split:CSEG_SB16_PF=1;CSEG_SB16_FEAT=1' \
  --variant 'q_stats:CSEG_SB16_PF=1;CSEG_SB16_FEAT=2' --variant 'q_both:CSEG_SB16_PF=1;CSEG_SB16_FEAT=3' \
  --variant 'q_both_pf2:CSEG_SB16_PF=2;CSEG_SB16_FEAT=3' \
  --variant 'q_both_nostatstore:CSEG_SB16_PF=1;CSEG_SB16_FEAT=3;CSEG_ABLATE=64' \
  --variant 'q_both_nostats:CSEG_SB16_PF=1;CSEG_SB16_FEAT=3;CSEG_ABLATE=8' \
  --variant 'q_both_nosplit:CSEG_SB16_PF=1;CSEG_SB16_FEAT=3;CSEG_ABLATE=2' \
  > $O/fwd.jsonl 2> $O/fwd.err
echo "fwd rc $?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r04j26/fwd.jsonl"):
    d = json.loads(l)
    if "shape" in d:
        print(d["shape"][1], d["shape"][2], "%-22s st %.1f plain %.1f diff %.3g" % (d["variant"], d["fwd_st_us"], d["fwd_us"], d["max_abs_diff_vs_first"]))
    else:
        print(d)
PY
