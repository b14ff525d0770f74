# GPU box: A/B/A/B of the short bench between the library in the tree and a previous build kept as tools/probes/libcseg_hip_prev.so
S="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4"
cp contrastiveseg_amd/libcseg_hip.so /tmp/lib_new.so
for r in 1 2 3; do for which in prev new; do
  if [ $which = prev ]; then cp tools/probes/libcseg_hip_prev.so contrastiveseg_amd/libcseg_hip.so; else cp /tmp/lib_new.so contrastiveseg_amd/libcseg_hip.so; fi
  CSEG_BENCH_GUARD=0 timeout 300 python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done; done
cp /tmp/lib_new.so contrastiveseg_amd/libcseg_hip.so
