"""ProjectionHead of the reference (lib/models/modules/projection.py:8-24): 1x1 conv -> BN+ReLU -> 1x1 conv
('convmlp') or a single 1x1 conv ('linear'), then L2 normalisation over channels. Convs/BN run on MIOpen."""
import torch.nn as nn
import torch.nn.functional as F

from contrastiveseg_amd.lib.models.tools.module_helper import ModuleHelper
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


class ProjectionHead(nn.Module):
    def __init__(self, dim_in, proj_dim=256, proj='convmlp', bn_type='torchsyncbn'):
        super(ProjectionHead, self).__init__()
        Log.info('proj_dim: {}'.format(proj_dim))
        if proj == 'linear':
            self.proj = nn.Conv2d(dim_in, proj_dim, kernel_size=1)
        elif proj == 'convmlp':
            self.proj = nn.Sequential(nn.Conv2d(dim_in, dim_in, kernel_size=1),
                                      ModuleHelper.BNReLU(dim_in, bn_type=bn_type),
                                      nn.Conv2d(dim_in, proj_dim, kernel_size=1))
        else:
            raise ValueError('unknown projection {!r}'.format(proj))

    def forward(self, x):
        return F.normalize(self.proj(x), p=2, dim=1)
