# GPU box: the grouped-launch probe (tools/probes/group_probe.cpp) on the default form, and -- when a library built with the cycle
# account is present -- the kernel's own s_memtime laps (conv3x3_group.hip, CSEG_GROUP_TIMERS; profiles/r06_group_cycles.txt).
# The timed library is built in the container, next to the product library, and travels with the snapshot:
#   cd contrastiveseg_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DCSEG_GROUP_TIMERS -I ../../include -I . \
#       conv3x3_group.hip -o /tmp/group_timed.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/libcseg_hip_timers.so \
#       $(ls _obj/*.o | grep -v conv3x3_group.hip.o) /tmp/group_timed.o
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
P=tools/probes/group_probe
[ -x $P ] || g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/group_probe.cpp -o $P -L/opt/rocm/lib -lamdhip64 -ldl
B="CSEG_GROUP_PC=2;CSEG_GROUP_TILE=8"
for br in 4 3 2; do timeout 200 $P --batch 8 --branches $br --iters 10 --variant "default:$B" | cut -c1-420; done
if [ -f tools/probes/libcseg_hip_timers.so ]; then
  CSEG_LIB=tools/probes/libcseg_hip_timers.so timeout 200 $P --batch 8 --branches 4 --iters 10 --variant "timed:$B;CSEG_GROUP_ABLATE=128" | cut -c1-520
fi
