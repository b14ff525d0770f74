"""Diagnosis helper (GPU): the first two losses of the streams test's configuration under combinations of switches, to find what makes
the second step of a forked run differ from the single-stream run. Usage: python tools/stream_bisect.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


class MP(object):
    def __init__(self):
        self.undo = []

    def setattr(self, obj, name, val):
        self.undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, val)

    def done(self):
        for obj, name, val in reversed(self.undo):
            setattr(obj, name, val)
        self.undo = []


def main():
    import test_gpu_streams as T
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB
    torch.backends.cudnn.deterministic = True
    ref_sel = set()
    cases = [("single", False, {}), ("single again", False, {}), ("forks", True, {}), ("forks again", True, {}),
             ("forks, per-branch nodes", True, {"BLOCK_GROUP": False}), ("forks, sync after backward", True, {"sync": True}),
             ("single, per-branch nodes", False, {"BLOCK_GROUP": False})]
    for name, forks, extra in cases:
        mp = MP()
        if "BLOCK_GROUP" in extra:
            mp.setattr(K, "BLOCK_GROUP", extra["BLOCK_GROUP"])
        if extra.get("sync"):
            from contrastiveseg_amd.segmentor import trainer_contrastive as TC
            orig = torch.Tensor.backward

            def backward(self, *a, **k):
                r = orig(self, *a, **k)
                torch.cuda.synchronize()
                return r
            mp.setattr(torch.Tensor, "backward", backward)
        losses, grads = T._run(forks, False, mp)
        mp.done()
        terms, sel = T._run.last_second_step
        if name == "single":
            ref_sel = set(sel.tolist())
        print("%-32s losses %.7f %.7f | second step: CE %.7f contrast %.7f, %d of %d mined anchors differ from 'single'"
              % (name, losses[0], losses[1], terms[0], terms[1], len(ref_sel ^ set(sel.tolist())), len(sel)), flush=True)


if __name__ == "__main__":
    main()
