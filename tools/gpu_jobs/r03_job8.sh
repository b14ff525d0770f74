#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j8
mkdir -p $O
cd $R
timeout 300 python tools/ablate_probe.py > $O/ablate_probe.jsonl 2> $O/ablate_probe.err; cat $O/ablate_probe.jsonl; tail -3 $O/ablate_probe.err
