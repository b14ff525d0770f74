#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j12
mkdir -p $O
cd $R
CSEG_ABLATE_ONLY=0 timeout 200 python - > $O/wrw720.json 2> $O/wrw720.err <<'PY'
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from contrastiveseg_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
out = {}
for (B, C, H, W) in ((8, 720, 128, 256), (8, 48, 128, 256), (8, 96, 64, 128), (8, 192, 32, 64)):
    x = torch.randn(B, C, H, W, generator=g).relu_().to(dev)
    dy = (torch.randn(B, C, H, W, generator=g) * 1e-3).to(dev)
    ax, ad = K.tensor_amax(x), K.tensor_amax(dy)
    f = lambda: K.conv3x3_sb_wrw(x, dy, ax=ax, ady=ad)
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 200)
    out["wrw_%d" % C] = round(best, 1)
print(json.dumps(out))
PY
cat $O/wrw720.json; tail -2 $O/wrw720.err
timeout 300 python tools/host_profile.py 8 > $O/host_profile_b8.txt 2> $O/host_profile_b8.err; head -120 $O/host_profile_b8.txt; tail -3 $O/host_profile_b8.err
timeout 200 python tools/host_profile.py 1 > $O/host_profile_b1.txt 2> $O/host_profile_b1.err; head -3 $O/host_profile_b1.txt
