#!/bin/bash
# Round 4, closing validation of the tree with the 16-instruction split, the unmasked statistics pass and the 8-row tiles as
# defaults: the whole GPU suite, kernel / step / golden tests first (a cut-off then only loses the least affected files).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j29
mkdir -p $O
cd $R
timeout 325 python -m pytest -m gpu -q -p no:cacheprovider --timeout 200 --durations=12 \
  tests/test_gpu_conv3x3_sb.py tests/test_gpu_conv3x3_s2.py tests/test_gpu_kernels.py tests/test_gpu_bn.py tests/test_gpu_train_step.py \
  tests/test_step_golden.py tests/test_models_golden.py tests/test_zz_gpu_default_routes.py tests/test_gpu_streams.py \
  tests/test_gpu_sparse_embed.py tests/test_gpu_conv3x3.py tests/test_running_score.py tests/test_gpu_step_graph.py \
  tests/test_gpu_multirank.py tests/test_gpu_aug.py tests/test_gpu_aug_host.py > $O/tests.log 2>&1
echo "pytest rc $?"
grep -E "passed|failed|Error|Fatal|^FAILED|CSEG_ZZ" $O/tests.log | cut -c1-600 | tail -8
