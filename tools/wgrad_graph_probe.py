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
    orig_on, orig_group = K._on_wgrad_stream, K._group_wrw
    for item in sys.argv[1:]:
        parts = item.split(":")
        mode, w = parts[0], parts[1]
        flt = parts[2] if len(parts) > 2 else "all"          # bisect: which weight gradients go to the side stream / which record_stream calls are made
        step_graph.ENABLED = mode == "graph"
        K.WGRAD_STREAM = w == "1"
        names = {"c1": "conv1x1_sb_wrw", "c3": "conv3x3_sb_wrw", "s2": "conv3x3_s2_wrw"}

        def on_side(fn, *inputs, flt=flt):
            if flt in names and names[flt] not in fn.__code__.co_names:
                return fn()
            if flt == "group":
                return fn()
            if flt in ("norec", "norec_amax") and K._WGRAD["on"] and inputs[0].is_cuda and not torch.cuda.is_current_stream_capturing():
                cur = torch.cuda.current_stream(inputs[0].device)
                side = K._WGRAD["stream"]
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    out = fn()
                for t in inputs:
                    if t is not None and flt == "norec_amax" and t.numel() != K.AMAX_WORDS:
                        t.record_stream(side)
                out.record_stream(K._WGRAD["main"])
                K._WGRAD["used"] = True
                return out
            return orig_on(fn, *inputs)

        def group_side(items, flt=flt):
            if flt in names:
                return K.conv3x3_group_wrw(items)
            if flt in ("norec", "norec_amax") and K._WGRAD["on"] and items[0][0].is_cuda and not torch.cuda.is_current_stream_capturing():
                cur = torch.cuda.current_stream(items[0][0].device)
                side = K._WGRAD["stream"]
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    dws = K.conv3x3_group_wrw(items)
                for it in items:
                    for t in it:
                        if flt == "norec_amax" and t.numel() != K.AMAX_WORDS:
                            t.record_stream(side)
                for dw in dws:
                    dw.record_stream(K._WGRAD["main"])
                K._WGRAD["used"] = True
                return dws
            return orig_group(items)

        K._on_wgrad_stream, K._group_wrw = on_side, group_side
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
        print("%-8s wgrad %s (%s): loss %.6f, |grad| %.4e, %d of %d gradient tensors all-zero, distance from the first run %.3e; graph state: %s"
              % (mode, w, flt, l0, norm, zeros, len(g), dist, os.environ.get("CSEG_STEP_GRAPH_STATE", "-")[:60]), flush=True)
        del tr, data
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
