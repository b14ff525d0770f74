"""Model registry with the reference's keys for the contrastive hot path (lib/models/model_manager.py:48-98)."""
from contrastiveseg_amd.lib.models.nets.deeplab import DeepLabV3_MEM, DeepLabV3Contrast
from contrastiveseg_amd.lib.models.nets.hrnet import (HRNet_W48_CONTRAST, HRNet_W48_MEM, HRNet_W48_OCR_CONTRAST,
                                                      HRNet_W48_OCR_MEM)
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

SEG_MODEL_DICT = {
    'hrnet_w48_contrast': HRNet_W48_CONTRAST,
    'hrnet_w48_ocr_contrast': HRNet_W48_OCR_CONTRAST,
    'hrnet_w48_mem': HRNet_W48_MEM,
    'deeplab_v3_contrast': DeepLabV3Contrast,
    # not in the reference's table: buildable forms of BASELINE.json configs[3] / [4] (nets/hrnet.py:ContrastMemoryModel)
    'deeplab_v3_mem': DeepLabV3_MEM,
    'hrnet_w48_ocr_mem': HRNet_W48_OCR_MEM,
}


class ModelManager(object):
    def __init__(self, configer):
        self.configer = configer

    def semantic_segmentor(self):
        model_name = self.configer.get('network', 'model_name')
        if model_name not in SEG_MODEL_DICT:
            Log.error('Model: {} not valid!'.format(model_name))
            exit(1)
        from contrastiveseg_amd.lib.models.tools.module_helper import mark_conv_bn_pairs
        # conv -> BatchNorm neighbours: the split kernels of those convolutions emit the BN statistics in their epilogue
        return mark_conv_bn_pairs(SEG_MODEL_DICT[model_name](self.configer))
