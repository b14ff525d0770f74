"""Host half of the GPU data pipeline: the random decisions of
contrastiveseg_amd/lib/datasets/tools/gpu_aug.py against the REFERENCE'S OWN transform classes
(lib/datasets/tools/cv2_aug_transforms.py, cv2 stubbed -- it is not installed) under the same `random.seed`: resized size,
crop origin and size, flip and brightness shift of every sample must be identical, i.e. Python's `random` stream is
consumed in the reference's order. Runs where /root/reference exists. The collate pads follow collate.py:108-121."""
import random
import sys
import types

import numpy as np
import pytest

from oracle import ref_shim

TRANS = {"trans_seq": ["random_resize", "random_crop", "random_hflip", "random_brightness"],
         "random_brightness": {"ratio": 0.7, "shift_value": 10},
         "random_hflip": {"ratio": 0.5, "swap_pair": []},
         "random_resize": {"ratio": 0.9, "method": "random", "scale_range": [0.5, 2.0], "aspect_range": [0.9, 1.1]},
         "random_crop": {"ratio": 1.0, "crop_size": [200, 120], "method": "random", "allow_outside_center": False}}


def _configer(cls):
    return cls(config_dict={
        "data": {"num_classes": 19, "input_mode": "BGR", "image_tool": "cv2"},
        "train": {"data_transformer": {"size_mode": "fix_size", "input_size": [200, 120], "align_method": "only_pad",
                                       "pad_mode": "random"}},
        "train_trans": TRANS, "val_trans": {"trans_seq": []},
        "normalize": {"div_value": 255.0, "mean_value": [0.485, 0.456, 0.406], "mean": [0.485, 0.456, 0.406],
                      "std": [0.229, 0.224, 0.225]}})


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_decisions_match_reference_transform_classes():
    ref_shim.install()
    cv2 = sys.modules["cv2"]
    cv2.INTER_CUBIC, cv2.INTER_NEAREST = 2, 0
    cv2.resize = lambda x, size, interpolation=None: np.full((size[1], size[0]) + x.shape[2:], 128, dtype=x.dtype)
    cv2.flip = lambda x, code: x
    import collections
    import collections.abc
    import importlib
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable      # the reference predates Python 3.10 (:519)
    aug = importlib.import_module("lib.datasets.tools.cv2_aug_transforms")
    from lib.utils.tools.configer import Configer as RefConfiger
    log = []
    orig = aug._BaseTransform._process

    def spy(self, img, data_dict, skip, *args, **kw):
        out_img, out = orig(self, img, data_dict, skip, *args, **kw)
        log[-1].append((type(self).__name__, bool(skip), args, out_img.shape, int(out_img.reshape(-1)[0])))
        return out_img, out
    aug._BaseTransform._process = spy
    try:
        compose = aug.CV2AugCompose(_configer(RefConfiger), split="train")
        sizes = [(256, 160), (200, 120), (333, 217), (150, 100), (512, 256)] * 6
        random.seed(1234)
        for (w, h) in sizes:
            log.append([])
            compose(np.full((h, w, 3), 128, np.uint8), labelmap=np.zeros((h, w), np.float32))
    finally:
        aug._BaseTransform._process = orig
    from contrastiveseg_amd.lib.datasets.tools.gpu_aug import GPUAugCompose
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    mine = GPUAugCompose(_configer(Configer), split="train")
    random.seed(1234)
    n_flip = n_skip = 0
    for (w, h), entries in zip(sizes, log):
        p = mine.draw(w, h)
        by = {e[0]: e for e in entries}
        _, skip, args, shape, _ = by["RandomResize"]
        Wr, Hr = (w, h) if skip else args[0]
        assert (p.Wr, p.Hr) == (Wr, Hr)
        n_skip += skip
        _, skip, args, shape, _ = by["RandomCrop"]
        assert not skip and (args[0], args[1]) == (p.y_off, p.x_off) and tuple(args[2]) == (p.tw, p.th)
        assert shape[:2] == (p.th, p.tw)
        assert by["RandomHFlip"][1] == (not p.flip)
        n_flip += p.flip
        _, skip, _, _, value = by["RandomBrightness"]
        assert value - 128 == p.shift and (skip == (p.shift == 0) or not skip)
    assert 0 < n_flip < len(sizes) and n_skip > 0          # both branches of the coins were exercised


def test_collate_pads_and_unsupported_configs():
    from contrastiveseg_amd.lib.datasets.tools.gpu_aug import GPUBatchTransform
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    t = GPUBatchTransform(_configer(Configer))
    random.seed(7)
    rows = t.plan([(150, 100), (512, 256)]).numpy()
    for r in rows:
        Wr, Hr, x, y, tw, th, flip, shift, left, up = r[:10]
        assert 0 <= x and x + tw <= Wr and 0 <= y and y + th <= Hr and tw <= 200 and th <= 120
        assert 0 <= left <= 200 - tw and 0 <= up <= 120 - th
    bad = _configer(Configer)
    bad.get("train_trans")["trans_seq"] = ["random_rotate"]
    with pytest.raises(NotImplementedError):
        GPUBatchTransform(bad)


def test_oracle_resampling_rules():
    """Properties of the restated cv2 rules: identity size is a copy, cubic weights sum to 1 and reproduce constants,
    nearest picks floor(dst * scale), a 2x integer upscale of a ramp stays monotone."""
    from oracle import aug_oracle as A
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, size=(9, 13, 3)).astype(np.uint8)
    assert np.array_equal(A.resize_cubic_u8(img, (13, 9)), img)
    const = np.full((7, 5, 3), 77, np.uint8)
    assert np.array_equal(A.resize_cubic_u8(const, (11, 16)), np.full((16, 11, 3), 77, np.uint8))
    lab = np.arange(6 * 4).reshape(6, 4).astype(np.uint8)
    up = A.resize_nearest(lab, (8, 12))
    assert up.shape == (12, 8) and np.array_equal(up[::2, ::2], lab)
    w = A.cubic_coeffs_f32(np.linspace(0, 1, 11).astype(np.float32))
    assert np.allclose(w.sum(-1), 1.0, atol=1e-6)
    q = np.rint(w * np.float32(A.COEF_SCALE)).astype(int)           # the Q11 tables OpenCV's 8-bit path works with
    assert (np.abs(q.sum(-1) - A.COEF_SCALE) <= 1).all()


def make_folder_dataset(tmp_path, ids, rs):
    from PIL import Image
    for split, n in (("train", 5), ("val", 2)):
        (tmp_path / split / "image").mkdir(parents=True)
        (tmp_path / split / "label").mkdir(parents=True)
        for i in range(n):
            Image.fromarray(rs.randint(0, 256, size=(96, 160, 3)).astype(np.uint8)).save(
                str(tmp_path / split / "image" / ("f%02d.png" % i)))
            lab = np.full((96, 160), 255, np.uint8)
            for _ in range(8):
                y, x = rs.randint(0, 96), rs.randint(0, 160)
                lab[y:y + 40, x:x + 60] = ids[rs.randint(0, len(ids))]
            Image.fromarray(lab).save(str(tmp_path / split / "label" / ("f%02d.png" % i)))


def folder_trainer_config(tmp_path, ids, cpu):
    import os
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Configer(configs=os.path.join(root, "configs", "synthetic", "R_18_D_8_tiny.json"))
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    if cpu:
        cfg.add(["gpu"], None)
    cfg.update(["network", "bn_type"], "torchbn")
    cfg.add(["data", "data_dir"], str(tmp_path))
    cfg.add(["data", "label_list"], ids)
    cfg.update(["data", "num_classes"], 5)
    cfg.get("loss", "params").pop("ce_weight", None)
    cfg.update(["train", "batch_size"], 2)
    cfg.get("train", "data_transformer").update({"input_size": [96, 64], "align_method": "only_pad", "pad_mode": "random"})
    cfg.update(["val"], {"batch_size": 2, "data_transformer": {"size_mode": "fix_size", "input_size": [160, 96],
                                                               "align_method": "only_pad"}})
    cfg.add(["val_trans"], {"trans_seq": []})
    cfg.add(["train_trans"], {"trans_seq": ["random_resize", "random_crop", "random_hflip", "random_brightness"],
                              "random_brightness": {"ratio": 1.0, "shift_value": 10},
                              "random_hflip": {"ratio": 0.5, "swap_pair": []},
                              "random_resize": {"ratio": 1.0, "method": "random", "scale_range": [0.75, 1.5],
                                                "aspect_range": [0.9, 1.1]},
                              "random_crop": {"ratio": 1.0, "crop_size": [96, 64], "method": "random",
                                              "allow_outside_center": False}})
    cfg.update(["contrast", "max_views"], 3)
    cfg.update(["contrast", "warmup_iters"], 0)
    cfg.update(["solver", "max_iters"], 2)
    cfg.update(["solver", "test_interval"], 10 ** 9)
    cfg.get("checkpoints")["checkpoints_root"] = str(tmp_path)      # validation saves checkpoints: keep them out of the repo
    return cfg


def run_folder_trainer(tmp_path, cpu):
    import torch
    from contrastiveseg_amd.lib.datasets.data_loader import GPUAugLoader
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    ids = [7, 8, 11, 12, 13]
    make_folder_dataset(tmp_path, ids, np.random.RandomState(0))
    cfg = folder_trainer_config(tmp_path, ids, cpu)
    random.seed(5)
    torch.manual_seed(304)
    tr = Trainer(cfg)
    assert isinstance(tr.train_loader, GPUAugLoader) and len(tr.train_loader) == 2
    assert isinstance(tr.val_loader, GPUAugLoader)
    batch = next(iter(tr.train_loader))
    assert batch["img"].shape == (2, 3, 64, 96) and batch["labelmap"].shape == (2, 64, 96)
    assert int(batch["labelmap"].min()) >= -1 and int(batch["labelmap"].max()) <= 4
    vb = next(iter(tr.val_loader))
    assert vb["img"].shape == (2, 3, 96, 160)          # validation: no augmentation, normalise + encode only
    tr.train()                                          # 2 SGD steps, then the final validation pass
    assert cfg.get("iters") == 2 and 0.0 <= cfg.get("performance") <= 1.0
    return tr


def test_folder_loader_feeds_the_trainer_on_cpu(tmp_path, monkeypatch):
    """Reference directory layout (default_loader.py:108-200) -> FolderSource (PIL decode) -> GPUAugLoader -> Trainer:
    the loaders are picked up from `data.data_dir`, batches have the fixed input size and encoded labels, two SGD steps and
    a validation pass run. Device half = oracle/cpu_port.py."""
    from oracle import cpu_port
    cpu_port.install(monkeypatch)
    run_folder_trainer(tmp_path, cpu=True)


# ---- sharding of the file list across ranks (private, epoch-seeded generator; equal shard lengths) ----------------
def test_folder_source_shards_like_distributed_sampler(monkeypatch):
    import torch
    from contrastiveseg_amd.lib.datasets import data_loader as DL

    def source(rank, world, n, bs, shuffle, drop_last):
        src = DL.FolderSource.__new__(DL.FolderSource)
        src.pairs, src.batch_size, src.shuffle, src.drop_last = [("i%d" % k, "l%d" % k) for k in range(n)], bs, shuffle, drop_last
        src.epoch, src.seed = 0, 304
        monkeypatch.setattr(DL, "get_rank", lambda: rank)
        monkeypatch.setattr(DL, "get_world_size", lambda: world)
        return src

    n, world, bs = 2975, 4, 2                                  # Cityscapes train on 4 ranks: 2975 % 8 != 0
    shards = []
    for rank in range(world):
        src = source(rank, world, n, bs, True, True)
        torch.manual_seed(rank * 7919)                         # the global generator differs per rank (anchor sampling, bank)
        torch.randperm(1000 + rank)
        shards.append(src.shard(epoch=3))
        assert len(src) == n // (world * bs) == len(shards[-1]) // bs
    assert len({len(s) for s in shards}) == 1                  # every rank: the same number of batches
    flat = [k for s in shards for k in s]
    assert len(flat) == len(set(flat)) == (n // (world * bs)) * world * bs          # disjoint, only the tail is dropped
    src = source(0, world, n, bs, True, True)
    assert src.shard(epoch=3) == shards[0] and src.shard(epoch=4) != shards[0]     # function of (seed, epoch) only
    g = torch.Generator()
    g.manual_seed(304 + 3)
    assert shards[1] == torch.randperm(n, generator=g).tolist()[:(n // (world * bs)) * world * bs][1::world]
    # validation: nothing dropped, ranks padded to equal length by wrapping around (DistributedSampler's default)
    val = [source(r, 4, 10, 2, False, False).shard() for r in range(4)]
    assert [len(v) for v in val] == [3, 3, 3, 3] and set(k for v in val for k in v) == set(range(10))
    assert all(len(source(r, 4, 10, 2, False, False)) == 2 for r in range(4))
    it = source(0, 1, 5, 2, True, False)
    it.set_epoch(9)
    assert it.epoch == 9
