#!/bin/bash
# Lean probe call 4: A/B/A/B of the library against a copy built with -DCSEG_SPLIT_CLASSIC (the compiler's own split sequence) --
# is the weight gradient slower with the 8-instruction split, or was that box-to-box noise? Head shapes alongside.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j28
mkdir -p $O
P=tools/probes/conv_probe
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
for round in 1 2; do
  for lib in new classic; do
    if [ $lib = classic ]; then export CSEG_LIB=tools/probes/libcseg_classic.so; else unset CSEG_LIB; fi
    timeout 60 $P --iters 30 --wrw --shape 8,48,128,256 --shape 8,96,64,128 --shape 8,192,32,64 --shape 8,384,16,32 --variant 'default:' \
      2>> $O/err.txt | sed "s/^{/{\"lib\": \"$lib\", \"round\": $round, /" >> $O/ab.jsonl
  done
done
for lib in new classic; do
  if [ $lib = classic ]; then export CSEG_LIB=tools/probes/libcseg_classic.so; else unset CSEG_LIB; fi
  timeout 60 $P --iters 4 --wrw --nt 265 --shape 8,720,128,256 --variant 'default:' 2>> $O/err.txt | sed "s/^{/{\"lib\": \"$lib\", \"round\": 0, /" >> $O/ab.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04j28/ab.jsonl"):
    d = json.loads(l)
    if "shape" in d:
        print(d["lib"], d["round"], d["shape"][1], "st %.1f plain %.1f wrw %.1f" % (d["fwd_st_us"], d["fwd_us"], d["wrw_us"]))
PY
tail -3 $O/err.txt
