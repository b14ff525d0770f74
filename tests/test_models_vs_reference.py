"""Drop-in check of the model registry against the reference itself (build container only: skipped where
/root/reference is absent): identical state_dict keys/shapes and, under the same torch seed, identical values --
i.e. same construction order and initialisers -- so reference checkpoints interchange. CPU only, no forward."""
import pytest
import torch

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")

CASES = [("hrnet_w48_contrast", "hrnet48", {}),
         ("hrnet_w48_ocr_contrast", "hrnet48", {}),
         ("hrnet_w48_mem", "hrnet48", {"memory_size": 7}),
         ("deeplab_v3_contrast", "deepbase_resnet101_dilated8", {})]


def _mine(model, backbone, contrast, num_classes=19):
    from contrastiveseg_amd.lib.models.model_manager import ModelManager
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    c = {"proj_dim": 256}
    c.update(contrast)
    cfg = Configer(config_dict={"data": {"num_classes": num_classes},
                                "network": {"backbone": backbone, "model_name": model, "bn_type": "torchsyncbn",
                                            "resume": None, "pretrained": None, "multi_grid": [1, 1, 1]},
                                "contrast": c})
    return ModelManager(cfg).semantic_segmentor()


@pytest.mark.parametrize("model,backbone,contrast", CASES)
def test_state_dict_identical_under_same_seed(model, backbone, contrast):
    ref_shim.install()
    from lib.models.model_manager import ModelManager as RefManager
    cfg = ref_shim.configer(model_name=model, backbone=backbone, contrast=contrast)
    torch.manual_seed(304)
    ref = RefManager(cfg).semantic_segmentor().state_dict()
    torch.manual_seed(304)
    mine = _mine(model, backbone, contrast).state_dict()
    assert list(ref.keys()) == list(mine.keys())
    for k in ref:
        assert ref[k].shape == mine[k].shape, k
        assert torch.equal(ref[k], mine[k]), k
