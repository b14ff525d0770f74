#!/bin/bash
# Round 4, GPU call 5: step-graph parity with the eager path measured against itself (deterministic MIOpen solvers).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_step_graph.py -m gpu -q -s --timeout 400 > $O/tests.log 2>&1; grep -E "loss dev|passed|failed|Error|^E  " $O/tests.log | cut -c1-330 | tail -30
