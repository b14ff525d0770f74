"""One process, the hrnet18 trainer of tests/test_gpu_step_graph.py (MIOpen convolutions almost everywhere), two train steps per run:
single stream / forked streams / convolution-epilogue statistics on and off, interleaved and repeated -- do the first two losses depend
on anything but the arithmetic?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_step_graph as T
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.backbones import hrnet_backbone as HB
    from contrastiveseg_amd.segmentor.tools import step_graph
    step_graph.ENABLED = False
    K.CONV3X3_SB_MIN_TILES = 1
    K.CONV1X1_SB_MIN_TILES = 1
    torch.backends.cudnn.deterministic = True
    case = T.CASES[int(os.environ.get("PROBE_CASE", "0"))]
    plan = [("one", False, True), ("one", False, True), ("forks", True, True), ("forks", True, True), ("one/nostats", False, False),
            ("forks/nostats", True, False), ("one", False, True), ("forks", True, True)]
    for name, forks, stats in plan:
        HB.EAGER_FORKS = forks
        K.CONV_EPILOGUE_STATS = stats
        tr, data = T._trainer(*case)
        torch.manual_seed(17)
        losses = [float(tr.train_step(data)) for _ in range(3)]
        torch.cuda.synchronize()
        print("%-14s %s" % (name, ["%.6f" % v for v in losses]), flush=True)
        del tr, data
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
