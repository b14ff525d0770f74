"""TEST INFRASTRUCTURE. Rank bodies of the two-rank (gloo) tests that run the multi-rank code paths with the HIP sources on
the emulator as the device half (tests/test_emu_cabi.py): what RCCL would exchange goes through gloo, what the GPU would
compute is computed by the emulated library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Patch(object):
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _join(rank, world, port):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CSEG_DIST_BACKEND="gloo", CSEG_EMU_THREADS="4")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu import inject
    inject.install(_Patch())
    return dist


def syncbn_worker(rank, world, port, q):
    """FusedSyncBatchNorm (+ residual + ReLU) on this rank's half of the batch: cseg_bn_stats -> all-reduce of the packed
    fp64 moments -> cseg_bn_finalize -> cseg_bn_apply; backward: cseg_bn_bwd_reduce -> all-reduce -> cseg_bn_bwd_apply."""
    dist = _join(rank, world, port)
    import torch
    from contrastiveseg_amd.lib.models.tools.fused_bn import FusedSyncBatchNorm
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(4, 24, 20, 36, generator=gen) * 2 + 1
    r = torch.randn(4, 24, 20, 36, generator=gen)
    g = torch.randn(4, 24, 20, 36, generator=gen)
    sl = slice(rank * 2, rank * 2 + 2)
    m = FusedSyncBatchNorm(24).train()
    with torch.no_grad():
        m.weight.copy_(torch.linspace(0.5, 1.5, 24))
        m.bias.copy_(torch.linspace(-1, 1, 24))
    xd = x[sl].clone().requires_grad_(True)
    rd = r[sl].clone().requires_grad_(True)
    y = m(xd, residual=rd, relu=True)
    y.backward(g[sl])
    q.put((rank, y.detach().numpy(), xd.grad.numpy(), rd.grad.numpy(), m.weight.grad.numpy(), m.bias.grad.numpy(),
           m.running_mean.numpy(), m.running_var.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def cross_rank_worker(rank, world, port, q):
    """The cross-rank contrast set (lib/loss/loss_contrast.py:_forward_cross_rank): counts all-gather -> global plan -> local
    cseg_gather_anchors -> anchor all-gather -> cseg_contrast_fwd/bwd on the global set -> cseg_scatter_anchor_grad."""
    dist = _join(rank, world, port)
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_distributed_gloo import _case, _configer
    from contrastiveseg_amd.lib.loss.loss_contrast import PixelContrastLoss
    c, (target, seg, embed, _) = _case()
    B = c["B"] // world
    sl = slice(rank * B, (rank + 1) * B)
    crit = PixelContrastLoss(_configer(c, "global", 256))
    e = torch.from_numpy(embed[sl]).requires_grad_(True)
    torch.manual_seed(11)
    loss = crit(e, torch.from_numpy(target[sl]), seg=torch.from_numpy(seg[sl]))
    loss.backward()
    q.put((rank, float(loss.detach()), e.grad.numpy(), crit.last_selection["plan"].N))
    dist.barrier()
    dist.destroy_process_group()
