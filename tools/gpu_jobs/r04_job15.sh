#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j15
mkdir -p $O
cd $R
CSEG_STEP_GRAPH=0 timeout 300 python tools/host_profile.py 8 > $O/host_profile_b8.txt 2>&1; head -62 $O/host_profile_b8.txt | cut -c1-180
