#!/bin/bash
# Round 4, GPU call 13: do forked streams / epilogue statistics change the first losses of the MIOpen-heavy hrnet18 trainer?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j13
mkdir -p $O
cd $R
for c in 0 1; do PROBE_CASE=$c timeout 300 python tools/forks_determinism_probe.py > $O/probe_case$c.log 2>&1; grep -E "^one|^forks" $O/probe_case$c.log; echo; done
