#!/bin/bash
# Round 3, first GPU call: everything round 2 wrote after its GPU budget ran out.
#  1. the full GPU suite on the defaults (sanity of the tree as committed)
#  2. kernel-level parity of the unverified kernels: weight gradient v2, 1x1; the row-sparse projection-head backward
#  3. probes: weight gradient v1 vs v2 vs MIOpen, 1x1 vs rocBLAS, 192-channel forward with explicit tiling
#  4. step-level goldens with the split weight gradient on (v1, then v2), with 192 channels on, with the 1x1 kernel on
#  5. bench lines for each switch
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 > $O/gputest_default.log 2>&1; tail -3 $O/gputest_default.log
CSEG_TEST_SB_WRW_V2=1 CSEG_TEST_SB_1X1=1 CSEG_TEST_SB_NT=1 timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py -q > $O/sb_kernels.log 2>&1; tail -5 $O/sb_kernels.log | cut -c1-300
CSEG_TEST_SPARSE_EMBED=1 timeout 200 python -m pytest tests/test_gpu_sparse_embed.py -q > $O/sparse_embed.log 2>&1; tail -3 $O/sparse_embed.log | cut -c1-300
timeout 200 python tools/conv3x3_sb_wrw_probe.py > $O/wrw_probe.jsonl 2> $O/wrw_probe.err; cat $O/wrw_probe.jsonl
timeout 200 python tools/conv1x1_sb_probe.py > $O/c1_probe.jsonl 2> $O/c1_probe.err; cat $O/c1_probe.jsonl
GOLD="tests/test_models_golden.py tests/test_step_golden.py tests/test_gpu_train_step.py"
for cfg in "CSEG_CONV3X3_SB_WRW=1" "CSEG_CONV3X3_SB_WRW=1 CSEG_CONV3X3_SB_WRW_V=2" "CSEG_CONV3X3_SB_CHANNELS=48,96,192" "CSEG_CONV1X1_SPLIT_BF16=1" "CSEG_CONV1X1_SPLIT_BF16=1 CSEG_CONV1X1_SB_WRW=1" "CSEG_SPARSE_EMBED_GRAD=1"; do
  tag=$(echo "$cfg" | tr ' =,' '___')
  env $cfg timeout 400 python -m pytest $GOLD -q -x -m gpu > $O/gold_$tag.log 2>&1; echo "$cfg: $(tail -1 $O/gold_$tag.log)"
  env $cfg timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$cfg', d['value'], d['ms_per_step'])"
done
