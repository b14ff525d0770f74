"""Host-side cost of one train step of the benched configuration: cProfile over 3 steady steps (the GPU runs asynchronously; what
is measured is Python + launch time on the enqueueing threads), top functions by own time and by cumulative time. Optional arg:
global batch (default 8; 1 = the per-GPU batch of the 8-GPU strong-scaling point); second arg `dist1` = inside a one-rank RCCL process group
with the multi-rank code paths on (what a rank of an N-GPU job pays on the host)."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import bench

gb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dist1 = len(sys.argv) > 2 and sys.argv[2] == "dist1"       # the multi-rank code paths inside a one-rank RCCL process group
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
if dist1:
    from contrastiveseg_amd.lib.utils import distributed as D
    os.environ.update(CSEG_DIST_SINGLE_RANK="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(D.free_port()))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
tr, cfg, batch = bench.build_trainer(args, 1, dev, gb)
for _ in range(4):
    tr.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    tr.train_step(batch)
t_enq = (time.perf_counter() - t0) / 3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 3
print("global batch %d: host enqueue %.1f ms/step, with final sync %.1f ms/step" % (gb, 1e3 * t_enq, 1e3 * t_all))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.train_step(batch)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:190] for l in s.getvalue().splitlines()[:70]))
