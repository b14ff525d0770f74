"""a >= 200-step run of BASELINE configs[1] (HRNet-W48 + contrast_ce_loss, 3x512x1024, batch 8) in the default
arithmetic (f16x3 split-operand convolutions, one power-of-two scale per tensor) beside the strict-fp32 route
(CSEG_CONV3X3_SPLIT_BF16=0 / CSEG_CONV1X1_SPLIT_BF16=0: MIOpen / fp32-MFMA convolutions), same seed, same batches.
Training from a random initialisation is chaotic (a 1e-7 perturbation grows ~1000x per handful of steps, tests/test_gpu_step_graph.py),
so a THIRD run gives the yardstick: strict fp32 again with the input images scaled by (1 + 2^-23) -- one rounding step of fp32.
What the default arithmetic must show: finite losses for all steps, a curve that falls like the fp32 one, and a deviation from the
fp32 curve of the same size as the deviation the one-ulp perturbation produces.
Usage (GPU):  python tools/long_run_arith.py [--steps 200] [--batch 8] [--batches 8] > profiles/r05_long_run_arith.json"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, split, perturb, steps, batch, n_batches):
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    K.CONV3X3_SPLIT_BF16 = split
    K.CONV1X1_SPLIT_BF16 = split
    cfg = Configer(configs=os.path.join(ROOT, "configs", "cityscapes", "H_48_D_4.json"))
    cfg.update(["train", "batch_size"], batch)
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["solver", "max_iters"], 40000)
    cfg.update(["solver", "display_iter"], 10 ** 9)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    torch.manual_seed(304)
    tr = Trainer(cfg, train_loader=[])
    data = list(SyntheticLoader(cfg, tr.module_runner.device(), length=n_batches, seed=304, mode="blocky", fixed=False))
    if perturb:
        for d in data:
            d["img"] = d["img"] * (1.0 + 2.0 ** -23)
    tr.seg_net.train()
    tr.pixel_loss.train()
    torch.manual_seed(17)                      # the anchor draws (CPU generator)
    losses = []
    for i in range(steps):
        losses.append(float(tr.train_step(data[i % n_batches])))
    torch.cuda.synchronize()
    amax = max(float(p.detach().abs().max()) for p in tr.seg_net.parameters())
    del tr, data
    torch.cuda.empty_cache()
    return losses, amax


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--batches", type=int, default=8)
    a = ap.parse_args()
    runs = {}
    for name, split, perturb in (("default_f16x3", True, False), ("strict_fp32", False, False), ("strict_fp32_one_ulp_input", False, True)):
        runs[name] = run(name, split, perturb, a.steps, a.batch, a.batches)
        print(name, "done: first %.5f last %.5f" % (runs[name][0][0], runs[name][0][-1]), file=sys.stderr)
    d, f, p = (np.array(runs[k][0]) for k in ("default_f16x3", "strict_fp32", "strict_fp32_one_ulp_input"))
    w = 20                                     # windowed means: single steps differ by which batch's anchors flipped

    def win(v):
        return [float(v[i:i + w].mean()) for i in range(0, len(v) - w + 1, w)]
    rel_df = np.abs(d - f) / np.abs(f)
    rel_pf = np.abs(p - f) / np.abs(f)
    out = {
        "what": "BASELINE configs[1], batch %d, %d steps over %d synthetic batches (blocky labels), SGD lr 0.01 poly, seed 304" % (a.batch, a.steps, a.batches),
        "finite": {k: bool(np.isfinite(v[0]).all()) for k, v in runs.items()},
        "max_abs_parameter_after_run": {k: v[1] for k, v in runs.items()},
        "loss_window_means_%d_steps" % w: {"default_f16x3": win(d), "strict_fp32": win(f), "strict_fp32_one_ulp_input": win(p)},
        "first_10_steps": {"default_f16x3": d[:10].tolist(), "strict_fp32": f[:10].tolist()},
        "rel_dev_first_10_steps_default_vs_fp32": rel_df[:10].tolist(),
        "rel_dev_first_10_steps_one_ulp_vs_fp32": rel_pf[:10].tolist(),
        "max_rel_dev_by_window": {"default_vs_fp32": [float(rel_df[i:i + w].max()) for i in range(0, len(d) - w + 1, w)],
                                  "one_ulp_vs_fp32": [float(rel_pf[i:i + w].max()) for i in range(0, len(d) - w + 1, w)]},
        "window_mean_rel_dev": {"default_vs_fp32": [abs(x - y) / abs(y) for x, y in zip(win(d), win(f))],
                                "one_ulp_vs_fp32": [abs(x - y) / abs(y) for x, y in zip(win(p), win(f))]},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
