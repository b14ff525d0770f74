#!/bin/bash
# ONE parametrised script for everything this repo puts on an MI355X box through `gpurun` (replaces the 79 one-shot
# tools/gpu_jobs/r0N_job*.sh lab-notebook scripts of rounds 2-4; the numbers they produced are under profiles/ and in DESIGN.md).
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh <tag> <step> [<step> ...]'
# writes to gpurun_out/<tag>/ . Steps (run in the order given; each bounded by its own `timeout`):
#   env:<VAR=VAL> / unenv:<VAR>   environment for the steps that follow
#   smoke            __graft_entry__.smoke()
#   suite[:<expr>]   pytest -m gpu (optionally -k <expr>)
#   bench[:<args>]   bench.py with extra args (comma-separated, e.g. bench:--steps,20,--warmup,5); default = the short form
#   ab:<ENV=VAL>     short bench without / with / without / with the environment setting (A/B/A/B on one box)
#   stats[:<args>]   rocprofv3 --kernel-trace of 6 steady steps (extra bench.py args, comma-separated) -> step_steady_kernel_stats.csv + idle gaps
#   pmc              rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) -> step_pmc.json
#   usepmc           installs that step_pmc.json as profiles/r06_step_pmc.json on the box (for the bench steps of the same call)
#   longrun[:<n>]    tools/long_run_arith.py (default 200 steps): default arithmetic vs strict fp32 vs a one-ulp perturbation
#   probe:<args>     tools/probes/conv_probe with the given arguments (';' separates arguments)
#   gprobe:<args>    tools/probes/group_probe (grouped launches vs the one-layer launches they replace; ';' separates arguments)
#   gpmc:<ctrs>|<args>  the same probe under rocprofv3 --pmc <ctrs> (comma separated) -> gpmc_summary.csv (per-kernel averages)
#   py:<script,args> any python script of tools/ (comma-separated arguments), output kept as py_<script>.txt
#   host             tools/host_profile.py 8
#   hpmc:<ctrs>|<args>  tools/probes/conv_probe under rocprofv3 --pmc <ctrs> -> hpmc_summary.csv (per-kernel averages; the head kernels: --nt 265 --wrw)
#   contrast         tools/contrast_probe.py under rocprofv3 --kernel-trace --stats: fused vs three-launch contrastive forward
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:?tag}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SHORT="--no-kernels --no-cpu-baseline --no-fp32-pass --steps 12 --warmup 4"
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  case $name in
    env) export "$arg"; echo "export $arg" ;;
    unenv) unset "$arg"; echo "unset $arg" ;;
    smoke) timeout 150 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    suite)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "$arg" > $O/tests.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/tests.log 2>&1; fi
      grep -E "passed|failed|Error|Fatal|CSEG_ZZ|^FAILED" $O/tests.log | cut -c1-1500 | tail -12 ;;
    bench)
      a=${arg//,/ }; [ -z "$a" ] && a=$SHORT
      CSEG_BENCH_GUARD=0 timeout 900 python bench.py $a > $O/bench.log 2> $O/bench.err; grep '^{"metric"' $O/bench.log | tail -1 | cut -c1-1800 | tee -a $O/bench_lines.txt
      cp bench_detail.json $O/bench_detail.json 2>/dev/null ;;
    ab)
      for r in 1 2; do for on in 0 1; do
        if [ $on = 1 ]; then env "$arg" CSEG_BENCH_GUARD=0 timeout 200 python bench.py $SHORT > $O/ab_${on}_$r.log 2> $O/ab_${on}_$r.err
        else CSEG_BENCH_GUARD=0 timeout 200 python bench.py $SHORT > $O/ab_${on}_$r.log 2> $O/ab_${on}_$r.err; fi
        echo "$arg on=$on run $r: $(tail -1 $O/ab_${on}_$r.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "img/s")' 2>&1 | tail -1)"
      done; done | tee $O/ab.txt ;;
    stats)
      cd /tmp
      CSEG_BENCH_GUARD=0 timeout 500 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-fp32-pass ${arg//,/ } > $O/bench_under_rocprof.json 2> $O/trace.err
      cd $R
      T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
      MS=$(grep '^{"metric"' $O/bench_under_rocprof.json | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
      echo "under rocprof: $MS ms/step"
      python tools/trace_window_stats.py $T $(python -c "print(5*$MS/1000.0)") > $O/step_steady_kernel_stats.csv 2> $O/window.txt; cat $O/window.txt
      python tools/trace_gaps.py $T $(python -c "print(3*$MS/1000.0)") 30 > $O/step_trace_gaps.txt; head -12 $O/step_trace_gaps.txt | cut -c1-200
      [ -n "$CSEG_KEEP_TRACE" ] && gzip -c $T > $O/kernel_trace.csv.gz          # (env:CSEG_KEEP_TRACE=1: the raw dispatch list, ~1 MB, for timeline questions)
      rm -rf $O/trace ;;
    pmc)
      cd /tmp
      for ctr in FETCH_SIZE WRITE_SIZE; do
        CSEG_BENCH_GUARD=0 timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_$ctr -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernels --no-fp32-pass > $O/pmc_$ctr.out 2> $O/pmc_$ctr.err
        c=$(find $O/pmc_$ctr -name '*counter_collection.csv' | head -1)
        [ -n "$c" ] && python $R/tools/step_pmc_summary.py $c $ctr 2 > $O/step_pmc_$ctr.json 2> $O/step_pmc_$ctr.err
        rm -rf $O/pmc_$ctr
      done
      cd $R
      python tools/merge_step_pmc.py $O/step_pmc_FETCH_SIZE.json $O/step_pmc_WRITE_SIZE.json "tree of $TAG" > $O/step_pmc.json 2> $O/step_pmc.err
      head -5 $O/step_pmc.json | cut -c1-200 ;;
    usepmc) cp $O/step_pmc.json $R/profiles/r06_step_pmc.json && echo "profiles/r06_step_pmc.json <- $O/step_pmc.json" ;;      # (the bench steps that follow report it as roofline.traffic; copy the same file into profiles/ at home)
    longrun)
      timeout 600 python tools/long_run_arith.py --steps ${arg:-200} > $O/long_run_arith.json 2> $O/long_run_arith.err
      tail -3 $O/long_run_arith.err; python -c "import json; d=json.load(open('$O/long_run_arith.json')); print(d['finite']); print(d['window_mean_rel_dev'])" ;;
    probe)
      export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
      P=tools/probes/conv_probe
      [ -x $P ] || g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/conv_probe.cpp -o $P -L/opt/rocm/lib -lamdhip64 -ldl
      IFS=';' read -ra PA <<< "$arg"
      timeout 120 $P "${PA[@]}" > $O/probe_$(date +%s%N).jsonl 2>> $O/probe.err; cat $O/probe_*.jsonl | tail -40 | cut -c1-400 ;;
    gprobe)
      export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
      P=tools/probes/group_probe
      [ -x $P ] || g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/group_probe.cpp -o $P -L/opt/rocm/lib -lamdhip64 -ldl
      IFS=';' read -ra PA <<< "$arg"
      timeout 120 $P "${PA[@]}" >> $O/group_probe.jsonl 2>> $O/group_probe.err; tail -3 $O/group_probe.jsonl | cut -c1-600; tail -2 $O/group_probe.err ;;
    gpmc)          # gpmc:<counters, comma separated>|<probe args, ';' separated>: PMC counters of the grouped-launch probe's kernels
      export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
      ctrs=${arg%%|*}; pargs=${arg#*|}
      IFS=';' read -ra PA <<< "$pargs"
      cd /tmp
      CSEG_LIB=$R/contrastiveseg_amd/libcseg_hip.so timeout 200 rocprofv3 --pmc ${ctrs//,/ } --kernel-trace -d $O/gpmc -o g --output-format csv -- $R/tools/probes/group_probe "${PA[@]}" > $O/gpmc_probe.out 2> $O/gpmc.err
      cd $R
      c=$(find $O/gpmc -name '*counter_collection.csv' | head -1)
      [ -n "$c" ] && python tools/pmc_kernel_summary.py $c group >> $O/gpmc_summary.csv
      rm -rf $O/gpmc; tail -4 $O/gpmc_summary.csv | cut -c1-700 ;;
    hpmc)          # hpmc:<counters, comma separated>|<conv_probe args, ';' separated>: PMC counters of the one-layer 3x3 kernels (the head: --nt 265)
      export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
      P=tools/probes/conv_probe
      [ -x $P ] || g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/conv_probe.cpp -o $P -L/opt/rocm/lib -lamdhip64 -ldl
      ctrs=${arg%%|*}; pargs=${arg#*|}
      IFS=';' read -ra PA <<< "$pargs"
      cd /tmp
      CSEG_LIB=$R/contrastiveseg_amd/libcseg_hip.so timeout 300 rocprofv3 --pmc ${ctrs//,/ } --kernel-trace -d $O/hpmc -o h --output-format csv -- $R/$P "${PA[@]}" > $O/hpmc_probe.out 2> $O/hpmc.err
      cd $R
      c=$(find $O/hpmc -name '*counter_collection.csv' | head -1)
      [ -n "$c" ] && python tools/pmc_kernel_summary.py $c conv3x3 >> $O/hpmc_summary.csv
      rm -rf $O/hpmc; tail -6 $O/hpmc_summary.csv | cut -c1-900 ;;
    contrast)
      cd /tmp
      timeout 300 rocprofv3 --kernel-trace --stats -d $O/ctrace -o c --output-format csv -- python $R/tools/contrast_probe.py > $O/contrast_probe.json 2> $O/contrast_probe.err
      cd $R
      cat $O/contrast_probe.json | cut -c1-900
      st=$(find $O/ctrace -name "*kernel_stats.csv" | head -1)
      [ -n "$st" ] && grep -E "Name|s_gemm|row_pass|mean_kernel|contrast_fused" $st | cut -c1-220 > $O/contrast_fused_kernels.txt; cat $O/contrast_fused_kernels.txt
      rm -rf $O/ctrace ;;
    py) a=${arg//,/ }; timeout 600 python $a 2>&1 | tee $O/py_$(basename ${a%% *} .py).txt | cut -c1-1500 | tail -20 ;;
    host) a=${arg//,/ }; [ -z "$a" ] && a=8
      timeout 200 python tools/host_profile.py $a > $O/host_profile_${a// /_}.txt 2>&1; grep "host enqueue" $O/host_profile_${a// /_}.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
