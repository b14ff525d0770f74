#!/bin/bash
# Round 4, GPU call 12: step-graph parity again with diagnostics (which gradients differ).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j12
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_step_graph.py -m gpu -q -s --timeout 400 > $O/tests_graph.log 2>&1; grep -E "losses eager|passed|failed|Fatal|AssertionError" $O/tests_graph.log | cut -c1-1500 | tail -14
