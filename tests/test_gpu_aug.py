"""Device half of the GPU data pipeline (csrc/augment.hip) against the forward numpy restatement of the reference chain
(oracle/aug_oracle.py): labels bit-exact, and -- since round 4, when both sides restate OpenCV's 11-bit FIXED-POINT cubic (integer
arithmetic after the float coefficient tables) -- every image pixel on the same grey level: what is left is the fp32 normalisation
(value / div - mean) / std, compared to 1e-5. Also end to end through GPUBatchTransform with the reference's config schema."""
import random

import numpy as np
import pytest
import torch

from oracle import aug_oracle as A

pytestmark = pytest.mark.gpu
MEAN, STD, DIV = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], 255.0


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _smooth_images(rs, B, H, W):
    """Band-limited images (sums of a few sinusoids + noise): like photographs, unlike white noise, the cubic result
    rarely sits within rounding of x.5."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.zeros((B, H, W, 3))
    for b in range(B):
        for c in range(3):
            v = 127.0
            for _ in range(4):
                fy, fx, ph = rs.uniform(0.01, 0.2), rs.uniform(0.01, 0.2), rs.uniform(0, 6.28)
                v = v + rs.uniform(10, 40) * np.sin(fy * yy + fx * xx + ph)
            out[b, :, :, c] = v + rs.normal(0, 6, size=(H, W))
    return np.clip(np.around(out), 0, 255).astype(np.uint8)


def _rows(B, Hs, Ws, Wt, Ht, rs):
    rows = []
    for b in range(B):
        if b == 0:                      # resize skipped, crop smaller than the target: exercises the collate padding
            Wr, Hr = Ws, Hs
        else:
            s = rs.uniform(0.5, 2.0)
            Wr, Hr = int(Ws * s * rs.uniform(0.95, 1.05)), int(Hs * s)
        tw, th = min(Wt, Wr), min(Ht, Hr) if b else Ht // 2
        x, y = rs.randint(0, Wr - tw + 1), rs.randint(0, Hr - th + 1)
        left, up = rs.randint(0, Wt - tw + 1), rs.randint(0, Ht - th + 1)
        rows.append([Wr, Hr, x, y, tw, th, int(rs.rand() < 0.5), int(rs.randint(-10, 11)), left, up, 0, 0])
    return np.array(rows, dtype=np.int32)


@pytest.mark.parametrize("B,Hs,Ws,Ht,Wt", [(4, 96, 160, 64, 128), (3, 61, 83, 48, 80), (2, 128, 256, 128, 256)])
def test_augment_kernel_matches_forward_chain(B, Hs, Ws, Ht, Wt):
    dev = _dev()
    from contrastiveseg_amd import kernels as K
    rs = np.random.RandomState(Hs + Wt)
    img = _smooth_images(rs, B, Hs, Ws)
    lab = rs.randint(0, 34, size=(B, Hs, Ws)).astype(np.uint8)
    lab[rs.rand(B, Hs, Ws) < 0.05] = 255
    lut = np.full(256, 255, np.int16)
    for i, cid in enumerate([7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]):
        lut[cid] = i
    rows = _rows(B, Hs, Ws, Wt, Ht, rs)
    got_img, got_lab = K.augment_batch(torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev),
                                       torch.from_numpy(lut).to(dev), torch.from_numpy(rows), (Ht, Wt), DIV, MEAN, STD)
    got_img, got_lab = got_img.cpu().numpy(), got_lab.cpu().numpy()
    lsb = 1.0 / DIV / min(STD)
    for b in range(B):
        want_img, want_lab = A.apply_chain(img[b], lab[b], rows[b], (Wt, Ht), DIV, MEAN, STD, lut)
        assert np.array_equal(got_lab[b], want_lab), "labels differ for image %d" % b
        diff = np.abs(got_img[b] - want_img)
        assert diff.max() <= 1e-5, (diff.max(), lsb, float((diff > 1e-5).mean()))     # no pixel is a grey level (lsb) off


def test_gpu_batch_transform_end_to_end():
    dev = _dev()
    from contrastiveseg_amd.lib.datasets.tools.gpu_aug import GPUBatchTransform
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    cfg = Configer(config_dict={
        "data": {"num_classes": 19, "label_list": [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]},
        "train": {"data_transformer": {"size_mode": "fix_size", "input_size": [128, 64], "align_method": "only_pad",
                                       "pad_mode": "random"}},
        "train_trans": {"trans_seq": ["random_resize", "random_crop", "random_hflip", "random_brightness"],
                        "random_brightness": {"ratio": 1.0, "shift_value": 10},
                        "random_hflip": {"ratio": 0.5, "swap_pair": []},
                        "random_resize": {"ratio": 1.0, "method": "random", "scale_range": [0.5, 2.0],
                                          "aspect_range": [0.9, 1.1]},
                        "random_crop": {"ratio": 1.0, "crop_size": [128, 64], "method": "random",
                                        "allow_outside_center": False}},
        "normalize": {"div_value": DIV, "mean": MEAN, "std": STD}})
    t = GPUBatchTransform(cfg)
    rs = np.random.RandomState(3)
    img = _smooth_images(rs, 4, 100, 180)
    lab = rs.choice([7, 8, 11, 26, 33, 0, 255], size=(4, 100, 180)).astype(np.uint8)
    random.seed(99)
    out = t(torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev))
    assert out["img"].shape == (4, 3, 64, 128) and out["labelmap"].shape == (4, 64, 128)
    assert out["labelmap"].dtype == torch.int64 and int(out["labelmap"].min()) >= -1 and int(out["labelmap"].max()) <= 18
    rows = out["aug_params"].cpu().numpy()
    lut = t.lut.cpu().numpy()
    for b in range(4):
        want_img, want_lab = A.apply_chain(img[b], lab[b], rows[b], (128, 64), DIV, MEAN, STD, lut)
        assert np.array_equal(out["labelmap"][b].cpu().numpy(), want_lab)
        assert np.abs(out["img"][b].cpu().numpy() - want_img).max() <= 1.0 / DIV / min(STD) * 1.001 + 1e-6


def test_folder_loader_feeds_the_trainer_on_gpu(tmp_path):
    """Files -> PIL decode -> pinned upload on a side stream -> cseg_augment_batch -> Trainer (train + validation)."""
    _dev()
    from test_gpu_aug_host import run_folder_trainer
    run_folder_trainer(tmp_path, cpu=False)
