"""DeepLabV3Contrast with the reference's parameter names, output dict and initialisation
(lib/models/nets/deeplab.py:8-41): dilated ResNet encoder, projection head on layer4, DeepLab head with DSN
auxiliary output. Head widths follow the encoder (reference hard-codes 1024/2048, i.e. Bottleneck encoders)."""
import torch.nn as nn

from contrastiveseg_amd.lib.models.backbones.backbone_selector import BackboneSelector
from contrastiveseg_amd.lib.models.modules.decoder_block import DeepLabHead
from contrastiveseg_amd.lib.models.modules.projection import ProjectionHead


class DeepLabV3Contrast(nn.Module):
    def __init__(self, configer):
        super(DeepLabV3Contrast, self).__init__()
        self.configer = configer
        self.num_classes = self.configer.get('data', 'num_classes')
        self.backbone = BackboneSelector(configer).get_backbone()
        self.proj_dim = self.configer.get('contrast', 'proj_dim')
        top = self.backbone.get_num_features()
        in_channels = [top // 2, top]
        self.proj_head = ProjectionHead(dim_in=in_channels[1], proj_dim=self.proj_dim)
        self.decoder = DeepLabHead(num_classes=self.num_classes, bn_type=self.configer.get('network', 'bn_type'),
                                   in_channels=in_channels)
        for modules in [self.proj_head, self.decoder]:
            for m in modules.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight.data)
                    if m.bias is not None:
                        m.bias.data.zero_()

    def forward(self, x_, with_embed=False, is_eval=False):
        x = self.backbone(x_)
        embedding = self.proj_head(x[-1])
        seg, seg_aux = self.decoder(x[-4:])
        return {'embed': embedding, 'seg_aux': seg_aux, 'seg': seg}


def _memory_model():
    from contrastiveseg_amd.lib.models.nets.hrnet import ContrastMemoryModel

    class DeepLabV3_MEM(ContrastMemoryModel):
        """DeepLabV3Contrast + per-class pixel / segment queues: the buildable form of BASELINE.json configs[3]."""
        ENCODER = DeepLabV3Contrast
    return DeepLabV3_MEM


DeepLabV3_MEM = _memory_model()
