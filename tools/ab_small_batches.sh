S="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4"
for b in 4 2 1; do for g in 1 0 1 0; do
  r=$(CSEG_BENCH_GUARD=0 CSEG_STEP_GRAPH=0 CSEG_BLOCK_GROUP=$g timeout 200 python bench.py $S --global-batch $b 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
  echo "batch $b eager BLOCK_GROUP=$g: $r ms/step"
done; done
