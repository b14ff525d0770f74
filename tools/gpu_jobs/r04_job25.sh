#!/bin/bash
# Lean probe call (no Python): branch-convolution variants of the opt-in deeper-prefetch kernel + ablations, and the weight
# gradient's ablations at the branch shapes. Seconds of GPU time.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j25
mkdir -p $O
P=tools/probes/conv_probe
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 150 $P --iters 30 --shape 8,48,128,256 --shape 8,192,32,64 --shape 8,384,16,32 \
  --variant 'default:' --variant 'pf1:CSEG_SB16_PF=1' --variant 'pf2:CSEG_SB16_PF=2' --variant 'pf3:CSEG_SB16_PF=3' \
  --variant 'pf2_noload:CSEG_SB16_PF=2;CSEG_ABLATE=1' --variant 'pf2_nosplit:CSEG_SB16_PF=2;CSEG_ABLATE=2' \
  --variant 'pf2_nomfma:CSEG_SB16_PF=2;CSEG_ABLATE=4' --variant 'pf2_nostore:CSEG_SB16_PF=2;CSEG_ABLATE=16' \
  --variant 'pf2_nomem:CSEG_SB16_PF=2;CSEG_ABLATE=17' --variant 'pf2_memonly:CSEG_SB16_PF=2;CSEG_ABLATE=6' \
  --variant 'pf2_loadsonly:CSEG_SB16_PF=2;CSEG_ABLATE=22' --variant 'pf2_storesonly:CSEG_SB16_PF=2;CSEG_ABLATE=7' \
  --variant 'pf2_mfmaonly:CSEG_SB16_PF=2;CSEG_ABLATE=19' --variant 'pf2_nothing:CSEG_SB16_PF=2;CSEG_ABLATE=23' \
  --variant 'pf1_nomem:CSEG_SB16_PF=1;CSEG_ABLATE=17' --variant 'pf3_memonly:CSEG_SB16_PF=3;CSEG_ABLATE=6' \
  > $O/fwd.jsonl 2> $O/fwd.err
echo "fwd rc $?"
timeout 100 $P --iters 30 --wrw --shape 8,48,128,256 --shape 8,96,64,128 --shape 8,192,32,64 --shape 8,384,16,32 \
  --variant 'default:' --variant 'w_nomfma:CSEG_ABLATE=1' --variant 'w_nosplit:CSEG_ABLATE=2' --variant 'w_noload:CSEG_ABLATE=4' \
  --variant 'w_nosplit_noload:CSEG_ABLATE=6' --variant 'w_consumers_only:CSEG_ABLATE=14' --variant 'w_mfma_off_all:CSEG_ABLATE=7' \
  > $O/wrw.jsonl 2> $O/wrw.err
echo "wrw rc $?"
cat $O/fwd.jsonl | cut -c1-200
