#!/bin/bash
# Round 4, GPU call 2: the capture probe again, with every eager iteration on the capture stream and no surviving autograd graph.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j2
mkdir -p $O
cd $R
timeout 280 python tools/graph_probe.py two two_st one_st relaxed_st > $O/graph_probe.log 2>&1; cat $O/graph_probe.log | cut -c1-900
