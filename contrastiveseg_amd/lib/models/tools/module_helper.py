"""Norm plug-in of the reference (lib/models/tools/module_helper.py:29-121), restricted to the two branches the
hot-path configs reach: 'torchbn' -> FusedBatchNorm2d (an nn.BatchNorm2d), 'torchsyncbn' -> FusedSyncBatchNorm (an
nn.SyncBatchNorm: global-batch statistics over RCCL when a process group exists, plain batch norm otherwise). Both are
computed by the cseg_bn_* HIP kernels and can fuse the ReLU / residual add that follows (fused_bn.py); parameters,
buffers and state_dict keys are those of the torch classes. Pretrained loading mirrors :124-235 for the two backbone
families of the hot path."""
import torch
import torch.nn as nn

from contrastiveseg_amd.lib.models.tools.fused_bn import FusedBatchNorm2d, FusedSyncBatchNorm
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

_NORMS = {'torchbn': FusedBatchNorm2d, 'torchsyncbn': FusedSyncBatchNorm}


class ModuleHelper(object):
    @staticmethod
    def BatchNorm2d(bn_type='torch', ret_cls=False):
        if bn_type not in _NORMS:
            Log.error('Not support BN type: {}.'.format(bn_type))
            exit(1)
        return _NORMS[bn_type]

    @staticmethod
    def BNReLU(num_features, bn_type=None, **kwargs):
        # reference: nn.Sequential(BN, nn.ReLU()); the ReLU has no state, so folding it into the norm kernel keeps the
        # state_dict keys ('0.weight', ...) unchanged
        return nn.Sequential(ModuleHelper.BatchNorm2d(bn_type)(num_features, act='relu', **kwargs))

    @staticmethod
    def load_model(model, pretrained=None, all_match=True, network='resnet101'):
        if pretrained is None:
            return model
        Log.info('Loading pretrained model:{}'.format(pretrained))
        src = torch.load(pretrained, map_location='cpu')
        dst = model.state_dict()
        if all_match:
            # torchvision-style stems are stored without the `resinit.` prefix (reference :130-139)
            load = {('resinit.' + k if 'resinit.' + k in dst else k): v for k, v in src.items()}
            model.load_state_dict(load)
        else:
            if network != 'hrnet':
                raise NotImplementedError('pretrained loading for {!r} is outside the hot path'.format(network))
            load = {k: v for k, v in src.items() if k in dst}
            Log.info('Missing keys: {}'.format(list(set(dst) - set(load))))
            dst.update(load)
            model.load_state_dict(dst)
        return model
