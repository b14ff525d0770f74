"""Where the forward of the benched step spends its GPU time WITHOUT a profiler attached: HIP events around the two halves of every
HighResolutionModule -- the residual chains of the branches (grouped launches) and the exchange unit (forked 1x1 / stride-2 paths) --
next to the host time spent enqueueing the same region. A region whose GPU span is close to its host time is fed just in time (host-bound);
one whose GPU span is far above it is device-bound. Usage: python tools/region_probe.py [global batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import bench

gb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
tr, cfg, batch = bench.build_trainer(args, 1, dev, gb)
from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB

records = []          # (name, n_branches, event0, event1, host seconds)
on = [False]


def wrap(name):
    orig = getattr(HB.HighResolutionModule, name)

    def inner(self, x, *a, **k):
        if not on[0]:
            return orig(self, x, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        out = orig(self, x, *a, **k)
        dt = time.perf_counter() - t0
        e1.record()
        records.append((name, self.num_branches, e0, e1, dt))
        return out
    setattr(HB.HighResolutionModule, name, inner)


for n in ("_branches_grouped", "_exchange_forked", "_exchange_lockstep"):
    wrap(n)

# step boundary: [end of the backward launches -> optimizer done] and [optimizer done -> first kernel of the next forward has run]
bound = []            # (event after backward, event after optimizer.step, event after the stem of the next forward)
_opt_step = tr.optimizer.step


def opt_step(*a, **k):
    if not on[0]:
        return _opt_step(*a, **k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = _opt_step(*a, **k)
    e1.record()
    bound.append([e0, e1, None])
    return out


tr.optimizer.step = opt_step
net = tr.seg_net.module if hasattr(tr.seg_net, "module") else tr.seg_net
backbone = getattr(net, "backbone", None)
if backbone is not None:
    def after_stem(mod, inp, out):
        if on[0] and bound and bound[-1][2] is None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            bound[-1][2] = e
    stem = getattr(backbone, "conv1", None)
    if stem is not None:
        stem.register_forward_hook(after_stem)
for _ in range(4):
    tr.train_step(batch)
torch.cuda.synchronize()
steps = 5
on[0] = True
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(steps):
    tr.train_step(batch)
ev1.record()
torch.cuda.synchronize()
print("global batch %d: %.2f ms/step with the events in" % (gb, ev0.elapsed_time(ev1) / steps))
agg = {}
for name, nb, e0, e1, dt in records:
    a = agg.setdefault((name, nb), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += dt * 1e3
tot = {}
for (name, nb), (n, gpu, host) in sorted(agg.items()):
    print("%-20s %d branches: %5.1f regions/step, GPU span %7.3f ms/step (%.3f each), host %7.3f ms/step (%.3f each)"
          % (name, nb, n / steps, gpu / steps, gpu / n, host / steps, host / n))
    t = tot.setdefault(name, [0.0, 0.0])
    t[0] += gpu / steps
    t[1] += host / steps
for name, (gpu, host) in tot.items():
    print("%-20s total: GPU span %.2f ms/step, host %.2f ms/step" % (name, gpu, host))

done = [b for b in bound if b[2] is not None]
if done:
    print("step boundary, GPU time: end of backward -> optimizer done %.3f ms; optimizer done -> first stem convolution of the next step done %.3f ms "
          "(the optimizer's kernels: ~0.7 ms, the stem convolution: ~0.12 ms; the rest is the GPU waiting for the host)"
          % (sum(b[0].elapsed_time(b[1]) for b in done) / len(done), sum(b[1].elapsed_time(b[2]) for b in done) / len(done)))
