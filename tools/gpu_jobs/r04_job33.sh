#!/bin/bash
# The round's last seconds of GPU time: the opt-in XCD-aware block order on the 720-channel head kernel and the 720 -> 720 1x1
# convolution (the two largest one-tile launches), plain / remap / plain / remap, outputs compared on the hardware.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j33
mkdir -p $O
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 12 tools/probes/conv_probe --iters 3 --nt 265 --c1 --shape 4,720,128,256 \
  --variant 'plain:' --variant 'remap:CSEG_XCD_REMAP=1' --variant 'plain2:' --variant 'remap2:CSEG_XCD_REMAP=1' > $O/head.jsonl 2> $O/err.txt
cat $O/head.jsonl | cut -c1-330
