"""Pixel-embedding projection head (reference: lib/models/modules/projection.py:8-24).

`proj='convmlp'`: 1x1 conv (dim_in -> dim_in) -> BN + ReLU -> 1x1 conv (dim_in -> proj_dim); `proj='linear'`: a single
1x1 conv. The output is L2-normalised over channels (eps 1e-12), which is what makes anchor . contrast a cosine
similarity in the contrastive kernels. Parameter names (`proj.0`, `proj.1.0`, `proj.2` / `proj`) and creation order
match the reference so checkpoints and seeds interchange; the 1x1 convolutions are GEMMs on MIOpen/rocBLAS."""
import torch.nn as nn
import torch.nn.functional as F

from contrastiveseg_amd.lib.models.tools.module_helper import Conv1x1, ModuleHelper
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

_KINDS = ('linear', 'convmlp')


def _pointwise(cin, cout):
    return Conv1x1(cin, cout)            # an nn.Conv2d(cin, cout, 1); split-bf16 kernel when switched on


class ProjectionHead(nn.Module):
    def __init__(self, dim_in, proj_dim=256, proj='convmlp', bn_type='torchsyncbn'):
        super(ProjectionHead, self).__init__()
        if proj not in _KINDS:
            raise ValueError('unknown projection {!r}; expected one of {}'.format(proj, _KINDS))
        Log.info('proj_dim: {}'.format(proj_dim))
        self.dim_in, self.proj_dim, self.kind = dim_in, proj_dim, proj
        if proj == 'linear':
            self.proj = _pointwise(dim_in, proj_dim)
        else:
            stages = [_pointwise(dim_in, dim_in)]
            stages.append(ModuleHelper.BNReLU(dim_in, bn_type=bn_type))
            stages.append(_pointwise(dim_in, proj_dim))
            self.proj = nn.Sequential(*stages)

    def extra_repr(self):
        return 'dim_in={}, proj_dim={}, kind={}'.format(self.dim_in, self.proj_dim, self.kind)

    def forward(self, x):
        return F.normalize(self.proj(x), p=2, dim=1)
