# GPU box: A/B/A/B of the grouped-launch probe between the library in the tree and a previous build kept as tools/probes/libcseg_hip_prev.so
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
B="CSEG_GROUP_PC=2;CSEG_GROUP_TILE=8"
for r in 1 2; do for lib in tools/probes/libcseg_hip_prev.so contrastiveseg_amd/libcseg_hip.so; do
  for br in 4 3; do echo -n "$lib branches $br: "; CSEG_LIB=$lib timeout 200 tools/probes/group_probe --batch 8 --branches $br --iters 10 --variant "default:$B" | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["group_us"], "us, mismatched", d["mismatched_outputs"], d["mismatched_stats"])'; done
done; done
