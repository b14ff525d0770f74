"""End-to-end train steps of every model family / criterion of the hot path on the GPU (small widths and images so
the whole file runs in seconds): trainer -> registry -> HIP criterion -> backward -> SGD, memory-bank update included."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    ("hrnet_w48_contrast", "hrnet18", "contrast_ce_loss", "cityscapes/H_48_D_4.json", {}),
    ("hrnet_w48_ocr_contrast", "hrnet18", "contrast_auxce_loss", "coco_stuff/H_48_D_4.json", {}),
    ("hrnet_w48_mem", "hrnet18", "mem_contrast_ce_loss", "cityscapes/H_48_D_4_MEM.json", {"memory_size": 64}),
    ("deeplab_v3_contrast", "deepbase_resnet18_dilated8", "contrast_auxce_loss", "cityscapes/R_101_D_8.json", {}),
    # buildable forms of BASELINE.json configs[3] / [4] (memory bank on DeepLab / OCR), small widths
    ("deeplab_v3_mem", "deepbase_resnet18_dilated8", "mem_contrast_auxce_loss", "cityscapes/R_101_D_8_MEM.json",
     {"memory_size": 64}),
    ("hrnet_w48_ocr_mem", "hrnet18", "mem_contrast_auxce_loss", "coco_stuff/H_48_D_4_MEM.json", {"memory_size": 64}),
]


@pytest.mark.parametrize("model,backbone,loss,cfg_file,contrast", CASES)
def test_two_train_steps(model, backbone, loss, cfg_file, contrast):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    cfg = Configer(configs=os.path.join(ROOT, "configs", cfg_file))
    cfg.update(["network", "backbone"], backbone)
    cfg.update(["network", "model_name"], model)
    cfg.update(["loss", "loss_type"], loss)
    cfg.update(["data", "num_classes"], 7)
    cfg.get("loss", "params").pop("ce_weight", None)
    cfg.update(["train", "batch_size"], 2)
    cfg.get("train", "data_transformer")["input_size"] = [256, 128]
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["contrast", "max_views"], 1 if "mem" in loss else 12)
    for k, v in contrast.items():
        cfg.update(["contrast", k], v)
    cfg.update(["solver", "max_iters"], 2)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    torch.manual_seed(304)
    tr = Trainer(cfg, train_loader=[])
    loader = SyntheticLoader(cfg, tr.module_runner.device(), length=2, mode="blocky")
    tr.seg_net.train()
    w0 = next(tr.seg_net.parameters()).detach().clone()
    losses = [float(tr.train_step(b)) for b in loader]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    assert not torch.equal(w0, next(tr.seg_net.parameters()).detach()), "SGD did not update the weights"
    if "mem" in loss:
        net = tr.seg_net
        assert int(net.segment_queue_ptr.sum()) > 0 and int(net.pixel_queue_ptr.sum()) > 0
        n = torch.linalg.norm(net.pixel_queue, dim=2)
        assert torch.allclose(n, torch.ones_like(n), atol=1e-4)


FULL_SIZE = [
    # BASELINE.json configs[3]: DeepLabV3-R101-d8 at 3x512x1024 (features 65x129), K=19, with and without the bank
    ("cityscapes/R_101_D_8.json", 2, "uniform", {}),
    ("cityscapes/R_101_D_8_MEM.json", 2, "uniform", {}),
    # BASELINE.json configs[4]: HRNet-W48-OCR at 3x520x520 (features 130x130), K=171, blocky labels, region memory
    ("coco_stuff/H_48_D_4.json", 2, "blocky", {}),
    ("coco_stuff/H_48_D_4_MEM.json", 2, "blocky", {}),
]


@pytest.mark.parametrize("cfg_file,batch,labels,contrast", FULL_SIZE)
def test_full_size_train_steps(cfg_file, batch, labels, contrast):
    """The real widths, input sizes and class counts of BASELINE.json configs[3] / [4] (no
    resnet18/hrnet18 stand-ins): two train steps through the trainer with the shipped config files."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.data_helper import SyntheticLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    cfg = Configer(configs=os.path.join(ROOT, "configs", cfg_file))
    cfg.update(["train", "batch_size"], batch)
    cfg.update(["contrast", "warmup_iters"], 0)
    for k, v in contrast.items():
        cfg.update(["contrast", k], v)
    cfg.update(["solver", "max_iters"], 2)
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    torch.manual_seed(304)
    tr = Trainer(cfg, train_loader=[])
    loader = SyntheticLoader(cfg, tr.module_runner.device(), length=2, mode=labels)
    tr.seg_net.train()
    losses = [float(tr.train_step(b)) for b in loader]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    if cfg.exists("contrast", "with_memory"):
        net = tr.seg_net
        assert int(net.segment_queue_ptr.sum()) > 0 and int(net.pixel_queue_ptr.sum()) > 0
