"""Norm plug-in of the reference (lib/models/tools/module_helper.py:29-121), restricted to the two branches the
hot-path configs reach: 'torchbn' -> nn.BatchNorm2d, 'torchsyncbn' -> nn.SyncBatchNorm (global-batch statistics
over RCCL when a process group exists, plain batch norm otherwise). Pretrained loading mirrors :124-235 for the
two backbone families of the hot path."""
import torch
import torch.nn as nn

from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

_NORMS = {'torchbn': nn.BatchNorm2d, 'torchsyncbn': nn.SyncBatchNorm}


class ModuleHelper(object):
    @staticmethod
    def BatchNorm2d(bn_type='torch', ret_cls=False):
        if bn_type not in _NORMS:
            Log.error('Not support BN type: {}.'.format(bn_type))
            exit(1)
        return _NORMS[bn_type]

    @staticmethod
    def BNReLU(num_features, bn_type=None, **kwargs):
        return nn.Sequential(ModuleHelper.BatchNorm2d(bn_type)(num_features, **kwargs), nn.ReLU())

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
