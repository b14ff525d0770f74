# GPU box: A/B/A/B/A/B of the short bench between an older tree unpacked (and built) under _prev_tree/ and this tree.
#   at home:  mkdir _prev_tree && git archive <commit> | tar -x -C _prev_tree && (cd _prev_tree && python -c "from contrastiveseg_amd.csrc import build; build.build()")
#   gpurun -- 'bash tools/ab_tree.sh <tag>'
S="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ab_tree}; mkdir -p $O
for r in 1 2 3; do for which in prev new; do
  if [ $which = prev ]; then D=$R/_prev_tree; else D=$R; fi
  (cd $D && CSEG_BENCH_GUARD=0 timeout 300 python bench.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which run $r:', d['ms_per_step'], 'ms/step', d['value'], 'img/s')")
done; done | tee $O/ab_tree.txt
