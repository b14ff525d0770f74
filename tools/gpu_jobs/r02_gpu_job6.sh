#!/bin/bash
# Round 2: the step with the split-bf16 convolutions switched on (head 720->720 + 48/96-channel branches): bench line,
# then the model / one-SGD-step goldens and the train-step tests under the same switch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02t
export CSEG_CONV3X3_SPLIT_BF16=1
timeout 110 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernels > gpurun_out/r02t/bench_sb.json 2> gpurun_out/r02t/bench_sb.err
echo "bench rc=$?"
cat gpurun_out/r02t/bench_sb.json | cut -c1-700
timeout 170 python -m pytest tests/test_models_golden.py tests/test_step_golden.py tests/test_gpu_train_step.py -q -x -m gpu > gpurun_out/r02t/t.log 2>&1
echo "pytest rc=$?"
tail -12 gpurun_out/r02t/t.log | cut -c1-400
