"""Fused (Sync)BatchNorm + residual + ReLU HIP kernels (csrc/bn.hip, lib/models/tools/fused_bn.py) against
torch.nn.BatchNorm2d (+ add + ReLU) -- the reference's ModuleHelper.BatchNorm2d / BNReLU and residual-block tails
(lib/models/tools/module_helper.py:29-68, hrnet_backbone.py:49-105). The comparison target is torch's own CPU fp64
evaluation of the same module chain; bars: outputs and input gradients 1e-5, parameter gradients 1e-5 relative to the
gradient norm, running statistics 1e-6, batch counter exact."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [(8, 48, 128, 256), (2, 720, 32, 64), (3, 5, 7, 9), (6, 512, 1, 1), (2, 64, 65, 129), (1, 19, 33, 17),
          (4, 2048, 13, 17)]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _reference(x, r, g, w, b, rm, rv, relu, res, train):
    """torch fp64 on the CPU"""
    bn = nn.BatchNorm2d(x.shape[1]).double()
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(train)
    x = x.double().requires_grad_(True)
    r = r.double().requires_grad_(True)
    y = bn(x)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    y.backward(g.double())
    return y.detach(), x.grad, (r.grad if res else None), bn


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True), (False, True)])
@pytest.mark.parametrize("train", [True, False])
def test_fused_bn_matches_torch_fp64(shape, relu, res, train):
    dev = _dev()
    from contrastiveseg_amd.lib.models.tools.fused_bn import FusedBatchNorm2d
    gen = torch.Generator().manual_seed(sum(shape) + 2 * relu + res)
    C = shape[1]
    x = torch.randn(shape, generator=gen) * 1.7 + torch.randn(1, C, 1, 1, generator=gen) * 3.0   # |mean| up to ~5 std
    r = torch.randn(shape, generator=gen)
    g = torch.randn(shape, generator=gen)
    w, b = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    rm, rv = torch.randn(C, generator=gen), torch.rand(C, generator=gen) + 0.5
    y_ref, dx_ref, dr_ref, bn_ref = _reference(x, r, g, w, b, rm, rv, relu, res, train)

    m = FusedBatchNorm2d(C).to(dev)
    with torch.no_grad():
        m.weight.copy_(w); m.bias.copy_(b); m.running_mean.copy_(rm); m.running_var.copy_(rv)
    m.train(train)
    xd = x.to(dev).requires_grad_(True)
    rd = r.to(dev).requires_grad_(True)
    y = m(xd, residual=rd if res else None, relu=relu)
    y.backward(g.to(dev))
    torch.cuda.synchronize()

    def err(a, ref):
        return float((a.detach().cpu().double() - ref).abs().max())
    scale = max(1.0, float(y_ref.abs().max()))
    assert err(y, y_ref) <= 1e-5 * scale, err(y, y_ref)
    # ReLU mask flips are measure-zero events; allow none on these seeds
    gs = max(1.0, float(dx_ref.abs().max()))
    assert err(xd.grad, dx_ref) <= 2e-5 * gs, (err(xd.grad, dx_ref), gs)
    if res:
        assert err(rd.grad, dr_ref) <= 1e-6 * max(1.0, float(dr_ref.abs().max()))
    for name in ("weight", "bias"):
        ref = getattr(bn_ref, name).grad
        assert err(getattr(m, name).grad, ref) <= 1e-5 * max(1.0, float(ref.norm())), name
    assert err(m.running_mean, bn_ref.running_mean) <= 1e-6 * max(1.0, float(bn_ref.running_mean.abs().max()))
    assert err(m.running_var, bn_ref.running_var) <= 1e-6 * max(1.0, float(bn_ref.running_var.abs().max()))
    assert int(m.num_batches_tracked) == int(bn_ref.num_batches_tracked)


def test_fused_bn_is_deterministic_and_large_mean_safe():
    dev = _dev()
    from contrastiveseg_amd.lib.models.tools.fused_bn import FusedBatchNorm2d
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(8, 96, 64, 128, generator=gen) * 0.01 + 100.0).to(dev)       # mean = 1e4 std
    m = FusedBatchNorm2d(96).to(dev).train()
    y0 = m(x, relu=False)
    y1 = m(x, relu=False)
    assert torch.equal(y0, y1)
    ref = F.batch_norm(x.double().cpu(), None, None, training=True)
    assert float((y0.double().cpu() - ref).abs().max()) < 2e-3      # fp32 input quantisation at 100 +- 0.01 dominates


def _sync_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CSEG_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from contrastiveseg_amd.lib.utils.distributed import setup_process_group
    setup_process_group()                          # gloo: both ranks share cuda:0 (RCCL refuses one device twice)
    from contrastiveseg_amd.lib.models.tools.fused_bn import FusedSyncBatchNorm
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(4, 24, 20, 36, generator=gen) * 2 + 1
    r = torch.randn(4, 24, 20, 36, generator=gen)
    g = torch.randn(4, 24, 20, 36, generator=gen)
    sl = slice(rank * 2, rank * 2 + 2)
    m = FusedSyncBatchNorm(24).to(dev).train()
    with torch.no_grad():
        m.weight.copy_(torch.linspace(0.5, 1.5, 24)); m.bias.copy_(torch.linspace(-1, 1, 24))
    xd = x[sl].to(dev).requires_grad_(True)
    rd = r[sl].to(dev).requires_grad_(True)
    y = m(xd, residual=rd, relu=True)
    y.backward(g[sl].to(dev))
    torch.cuda.synchronize()
    q.put((rank, y.detach().cpu().numpy(), xd.grad.cpu().numpy(), rd.grad.cpu().numpy(), m.weight.grad.cpu().numpy(),
           m.bias.grad.cpu().numpy(), m.running_mean.cpu().numpy(), m.running_var.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_fused_syncbn_two_ranks_on_one_gpu_equal_single_process():
    """Two ranks (gloo transport, both on cuda:0) with half the batch each == one process with the whole batch: the
    packed fp64 moment / gradient-sum all-reduces of FusedSyncBatchNorm. d_weight/d_bias are rank-local sums whose
    total equals the single-process gradient (DDP averages them)."""
    dev = _dev()
    import torch.multiprocessing as mp
    from contrastiveseg_amd.lib.models.tools.fused_bn import FusedBatchNorm2d
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(4, 24, 20, 36, generator=gen) * 2 + 1
    r = torch.randn(4, 24, 20, 36, generator=gen)
    g = torch.randn(4, 24, 20, 36, generator=gen)
    m = FusedBatchNorm2d(24).to(dev).train()
    with torch.no_grad():
        m.weight.copy_(torch.linspace(0.5, 1.5, 24)); m.bias.copy_(torch.linspace(-1, 1, 24))
    xd = x.to(dev).requires_grad_(True)
    rd = r.to(dev).requires_grad_(True)
    y = m(xd, residual=rd, relu=True)
    y.backward(g.to(dev))
    y_all = np.concatenate([res[0][1], res[1][1]])
    dx_all = np.concatenate([res[0][2], res[1][2]])
    dr_all = np.concatenate([res[0][3], res[1][3]])
    assert np.abs(y_all - y.detach().cpu().numpy()).max() <= 1e-6
    assert np.abs(dx_all - xd.grad.cpu().numpy()).max() <= 1e-6
    assert np.array_equal(dr_all, rd.grad.cpu().numpy())
    assert np.abs(res[0][4] + res[1][4] - m.weight.grad.cpu().numpy()).max() <= 1e-4
    assert np.abs(res[0][5] + res[1][5] - m.bias.grad.cpu().numpy()).max() <= 1e-4
    for k in (0, 1):
        assert np.abs(res[k][6] - m.running_mean.cpu().numpy()).max() <= 1e-6
        assert np.abs(res[k][7] - m.running_var.cpu().numpy()).max() <= 1e-6
