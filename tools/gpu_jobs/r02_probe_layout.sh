#!/bin/bash
# Round-2 probe: which convolution kernels MIOpen picks for NCHW vs channels_last tensors (whole-step kernel stats).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_layout
mkdir -p $O
cd $R
for cl in 0 1; do
  export PYTORCH_MIOPEN_SUGGEST_NHWC=1 PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM=1
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/cl$cl -o cl$cl --output-format csv -- \
     python bench.py --steps 3 --warmup 2 --channels-last $cl --no-kernels --no-cpu-baseline > $O/bench_cl$cl.json 2> $O/bench_cl$cl.err
  f=$(find $O/cl$cl -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/cl${cl}_kernel_stats.csv
  find $O/cl$cl -name '*kernel_trace.csv' -delete
done
ls -la $O
