# GPU box: the grouped-launch probe over the forms / experiments of the 12-wave kernel (conv3x3_group.hip)
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
P=tools/probes/group_probe
[ -x $P ] || g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/probes/group_probe.cpp -o $P -L/opt/rocm/lib -lamdhip64 -ldl
B="CSEG_GROUP_PC=2;CSEG_GROUP_TILE=8"
for br in 4 3; do timeout 200 $P --batch 8 --branches $br --iters 10 --variant "plain:$B" --variant "dma_staging:$B;CSEG_GROUP_ABLATE=1024" --variant "plain2:$B" --variant "dma_staging2:$B;CSEG_GROUP_ABLATE=1024" | cut -c1-190; done
