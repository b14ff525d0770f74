S="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4"
O=gpurun_out/${TAG:-r06_final2}; mkdir -p $O
for b in 8 1; do
  CSEG_BENCH_GUARD=0 timeout 400 python bench.py $S --dist-single-rank --global-batch $b 2>$O/dist_b$b.err | tail -1 > $O/dist_b$b.json; python -c "import json; d=json.loads(open('$O/dist_b$b.json').read()); print('dist-single-rank batch $b:', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done
for w in cfg4 cfg5; do
  CSEG_BENCH_GUARD=0 timeout 600 python bench.py $S --workload $w 2>$O/$w.err | tail -1 > $O/$w.json; python -c "import json; d=json.loads(open('$O/$w.json').read()); print('$w:', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done
CSEG_BENCH_GUARD=0 timeout 400 python bench.py $S 2>/dev/null | tail -1 > $O/cfg2_short.json; python -c "import json; d=json.loads(open('$O/cfg2_short.json').read()); print('cfg2 short:', d['ms_per_step'], d['roofline'].get('traffic'), d['roofline'].get('traffic_source'))"
