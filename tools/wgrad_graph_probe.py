"""Diagnosis helper (GPU): does a step-graph replay deliver parameter gradients when the weight-gradient stream is on?
Runs the configuration of tests/test_gpu_step_graph.py (HRNet-W18 contrast, batch 2) in the orders given on the command line, each
item = <eager|graph>:<wgrad 0|1>, and prints the gradient norm of every run and its distance from the first run.
Usage: python tools/wgrad_graph_probe.py graph:1 eager:0    |    eager:1 graph:1 eager:0"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_step_graph as T
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.segmentor.tools import step_graph
    torch.backends.cudnn.deterministic = True
    K.CONV3X3_SB_MIN_TILES = 1
    K.CONV1X1_SB_MIN_TILES = 1
    step_graph.MODE = "1"
    step_graph.BRANCH_STREAMS = False
    first = None
    for item in sys.argv[1:]:
        mode, w = item.split(":")
        step_graph.ENABLED = mode == "graph"
        K.WGRAD_STREAM = w == "1"
        tr, data = T._trainer(*T.CASES[0])
        torch.manual_seed(17)
        l0 = float(tr.train_step(data))
        torch.cuda.synchronize()
        g = {k: p.grad.detach().float().cpu().numpy().copy() for k, p in tr.seg_net.named_parameters() if p.grad is not None}
        norm = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in g.values()))
        zeros = sum(1 for v in g.values() if not v.any())
        if first is None:
            first = g
        dist = np.sqrt(sum(float(((g[k] - first[k]).astype(np.float64) ** 2).sum()) for k in first))
        zn = [k for k, v in g.items() if not v.any()]
        worst = sorted(first, key=lambda k: -float(np.abs(g[k] - first[k]).max()))[:4]
        print("   all-zero: %s; largest differences: %s" % (zn[:6], [(k, float(np.abs(g[k] - first[k]).max()), float(np.abs(first[k]).max()))
                                                                         for k in worst]), flush=True)
        print("%-8s wgrad %s: loss %.6f, |grad| %.4e, %d of %d gradient tensors all-zero, distance from the first run %.3e; graph state: %s"
              % (mode, w, l0, norm, zeros, len(g), dist, os.environ.get("CSEG_STEP_GRAPH_STATE", "-")[:60]), flush=True)
        del tr, data
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
