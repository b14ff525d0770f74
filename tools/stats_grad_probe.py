"""First-step gradients of the hrnet18 trainer with the convolution-epilogue BN statistics on vs off: which tensors differ, by how much."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(stats, case):
    import test_gpu_step_graph as T
    from contrastiveseg_amd import kernels as K
    K.CONV_EPILOGUE_STATS = stats
    tr, data = T._trainer(*case)
    torch.manual_seed(17)
    acts = {}
    hooks = []
    for name, m in tr.seg_net.named_modules():
        if name.endswith(("bn1", "bn2", "bn3")) or name.split(".")[-1].isdigit():
            def hook(mod, inp, out, name=name):
                if torch.is_tensor(out) and name not in acts:
                    acts[name] = out.detach().float().cpu().numpy().copy()
                return None
            hooks.append(m.register_forward_hook(hook))
    l0 = float(tr.train_step(data))
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    grads = {k: p.grad.detach().float().cpu().numpy().copy() for k, p in tr.seg_net.named_parameters() if p.grad is not None}
    bufs = {k: v.detach().float().cpu().numpy().copy() for k, v in tr.seg_net.named_buffers()}
    del tr, data
    torch.cuda.empty_cache()
    return l0, grads, acts, bufs


def main():
    import test_gpu_step_graph as T
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB
    from contrastiveseg_amd.segmentor.tools import step_graph
    step_graph.ENABLED = False
    HB.EAGER_FORKS = False
    K.CONV3X3_SB_MIN_TILES = 1
    K.CONV1X1_SB_MIN_TILES = 1
    torch.backends.cudnn.deterministic = True
    case = T.CASES[0]
    a = run(False, case)
    b = run(True, case)
    print("loss", a[0], b[0])
    rows = []
    for k in a[2]:
        if k in b[2] and a[2][k].shape == b[2][k].shape:
            rows.append((float(np.abs(a[2][k] - b[2][k]).max()) / max(float(np.abs(a[2][k]).max()), 1e-9), k))
    rows.sort(reverse=True)
    print("activations (max rel dev), first in network order that exceed 1e-5:")
    order = [k for k in a[2]]
    shown = 0
    for k in order:
        d = [r for r in rows if r[1] == k]
        if d and d[0][0] > 1e-5 and shown < 12:
            print("   %.2e %s" % d[0]); shown += 1
    g = []
    for k in a[1]:
        den = max(float(np.linalg.norm(a[1][k])), 1e-12)
        g.append((float(np.linalg.norm(a[1][k] - b[1][k])) / den, k))
    g.sort(reverse=True)
    print("gradients (rel L2):", ["%.1e %s" % r for r in g[:10]])
    bb = sorted(((float(np.abs(a[3][k] - b[3][k]).max()) / max(float(np.abs(a[3][k]).max()), 1e-9), k) for k in a[3]), reverse=True)
    print("buffers:", ["%.1e %s" % r for r in bb[:6]])


if __name__ == "__main__":
    main()
