#!/bin/bash
# Round 3, first GPU calls: everything round 2 wrote after its GPU budget ran out. READ FIRST: the CSEG_ZZ line in the tail of
# GPUTEST_r02.json (tests/test_zz_gpu_default_routes.py ran these pieces once in the driver's round-end pass). Two stages so that the second can be
# trimmed to the switches whose kernels passed the first:
#   tools/r03_gpu_job1.sh a     (~15 GPU-min)
#     1. the full GPU suite on the defaults (sanity of the tree as committed)
#     2. kernel-level parity of the unverified pieces: weight gradient v2, 1x1 forward / weight gradient, explicit
#        channel tiling, the row-sparse projection-head backward
#     3. probes: weight gradient v1 vs v2 vs MIOpen, 1x1 vs rocBLAS, 192 / 384-channel forward
#   tools/r03_gpu_job1.sh b [switch ...]     (~5 GPU-min per switch; default: all of SWITCHES below)
#     4. step-level goldens with one switch on at a time
#     5. a bench line for each switch
export TMPDIR=/tmp
export CSEG_BENCH_GUARD=0      # job scripts choose the routes themselves: no automatic re-run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
stage=${1:-a}
shift
if [ "$stage" = "a" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 > $O/gputest_default.log 2>&1; tail -3 $O/gputest_default.log
  CSEG_TEST_SB_WRW_V2=1 CSEG_TEST_SB_1X1=1 CSEG_TEST_SB_NT=1 timeout 300 python -m pytest tests/test_gpu_conv3x3_sb.py -q > $O/sb_kernels.log 2>&1; tail -5 $O/sb_kernels.log | cut -c1-300
  CSEG_TEST_SPARSE_EMBED=1 timeout 200 python -m pytest tests/test_gpu_sparse_embed.py -q > $O/sparse_embed.log 2>&1; tail -3 $O/sparse_embed.log | cut -c1-300
  timeout 200 python tools/conv3x3_sb_wrw_probe.py > $O/wrw_probe.jsonl 2> $O/wrw_probe.err; cat $O/wrw_probe.jsonl
  timeout 200 python tools/conv1x1_sb_probe.py > $O/c1_probe.jsonl 2> $O/c1_probe.err; cat $O/c1_probe.jsonl
  timeout 200 python tools/conv3x3_sb_probe.py > $O/c3_probe.jsonl 2> $O/c3_probe.err; cat $O/c3_probe.jsonl
  exit 0
fi
SWITCHES=("CSEG_CONV3X3_SB_WRW=0" "CSEG_CONV3X3_SB_WRW_V=2" "CSEG_CONV3X3_SB_CHANNELS=48,96"
          "CSEG_CONV3X3_SB_WRW_CHANNELS=48,96,192,720"
          "CSEG_CONV3X3_SB_CHANNELS=48,96,192,384" "CSEG_CONV1X1_SPLIT_BF16=1" "CSEG_CONV1X1_SPLIT_BF16=1 CSEG_CONV1X1_SB_WRW=1"
          "CSEG_SPARSE_EMBED_GRAD=1")
if [ $# -gt 0 ]; then SWITCHES=("$@"); fi
GOLD="tests/test_models_golden.py tests/test_step_golden.py tests/test_gpu_train_step.py"
for cfg in "${SWITCHES[@]}"; do
  tag=$(echo "$cfg" | tr ' =,' '___')
  # CSEG_SB_MIN_TILES=1: at the goldens' own (small) shapes the grid-fill thresholds would keep every split-bf16 kernel off
  env $cfg CSEG_SB_MIN_TILES=1 timeout 400 python -m pytest $GOLD -q -x -m gpu > $O/gold_$tag.log 2>&1; echo "$cfg: $(tail -1 $O/gold_$tag.log)"
  env $cfg timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$cfg', d['value'], d['ms_per_step'])"
done
