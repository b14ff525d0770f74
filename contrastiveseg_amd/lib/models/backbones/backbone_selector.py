"""BackboneSelector with the reference's contract (lib/models/backbones/backbone_selector.py): picks the encoder
family from network.backbone. Only the two families of the hot path exist here."""
from contrastiveseg_amd.lib.models.backbones.hrnet_backbone import HRNetBackbone
from contrastiveseg_amd.lib.models.backbones.resnet_backbone import ResNetBackbone
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


class BackboneSelector(object):
    def __init__(self, configer):
        self.configer = configer

    def get_backbone(self, **params):
        backbone = self.configer.get('network', 'backbone')
        if 'hrne' in backbone:
            return HRNetBackbone(self.configer)(**params)
        if 'resnet' in backbone and 'wide' not in backbone and 'resnext' not in backbone:
            return ResNetBackbone(self.configer)(**params)
        Log.error('Backbone {} is invalid.'.format(backbone))
        exit(1)
