"""
ORACLE -- test infrastructure only. Imports the *reference itself* (read-only, /root/reference) on CPU so the
restatement in cseg_oracle.py can be pinned against it. Works only in the build container; /root/reference
does not exist on the GPU box, so nothing under `-m gpu`, smoke() or bench.py may import this module.

Accommodations: no bytecode written into the reference tree, Tensor.cuda -> identity,
stub modules for third-party imports pulled in by unrelated reference modules (yacs, timm, cv2, torchcontrib).
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lib", "loss"))


def install():
    import torch
    import torch.nn as nn

    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True
    torch.Tensor.cuda = lambda self, *a, **k: self

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class CN(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    if "yacs" not in sys.modules:
        _mod("yacs")
        _mod("yacs.config", CfgNode=CN)
        _mod("timm")
        _mod("timm.models")
        _mod("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda x: (x, x),
             trunc_normal_=lambda *a, **k: None)
        _mod("timm.models.registry", register_model=lambda f: f)
        _mod("timm.models.vision_transformer", _cfg=lambda **k: {}, Block=nn.Module, Attention=nn.Module)
        _mod("torchcontrib")
        _mod("cv2")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def configer(num_classes=19, model_name="hrnet_w48_contrast", backbone="hrnet48", loss_type="contrast_ce_loss",
             contrast=None, ce_weight=None, multi_grid=(1, 1, 1)):
    install()
    from lib.utils.tools.configer import Configer
    c = {"proj_dim": 256, "temperature": 0.1, "base_temperature": 0.07, "max_samples": 1024, "max_views": 100,
         "loss_weight": 0.1, "use_rmi": False, "use_lovasz": False, "warmup_iters": 0}
    c.update(contrast or {})
    params = {"ce_ignore_index": -1, "ce_reduction": "elementwise_mean"}
    if ce_weight is not None:
        params["ce_weight"] = list(ce_weight)
    return Configer(config_dict={
        "data": {"num_classes": num_classes}, "gpu": None,
        "network": {"backbone": backbone, "model_name": model_name, "bn_type": "torchsyncbn", "resume": None,
                    "pretrained": None, "stride": 8, "loss_balance": False,
                    "multi_grid": None if multi_grid is None else list(multi_grid),
                    "loss_weights": {"aux_loss": 0.4, "seg_loss": 1.0}},
        "contrast": c,
        "loss": {"loss_type": loss_type, "params": params}})
