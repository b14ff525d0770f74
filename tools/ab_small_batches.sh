# GPU box: batch 1 / 2 step time, hipGraph replay (the default there) with the exchange unit on the grouped launches or per site, A/B/A/B
S="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4"
for b in 1 2; do for g in 1 0 1 0; do
  r=$(CSEG_BENCH_GUARD=0 CSEG_EXCHANGE_GROUPED=$g timeout 300 python bench.py $S --global-batch $b 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("config",{}).get("step_graph"))')
  echo "batch $b EXCHANGE_GROUPED=$g: $r ms/step"
done; done
