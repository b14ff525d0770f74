"""Loss registry with the reference's keys (lib/loss/loss_manager.py:27-68). Only the criteria of the contrastive
hot path are registered; every other key of the reference's table is out of scope."""
from contrastiveseg_amd.lib.loss.loss_contrast import ContrastAuxCELoss, ContrastCELoss
from contrastiveseg_amd.lib.loss.loss_contrast_mem import ContrastAuxCELoss as MemContrastAuxCELoss
from contrastiveseg_amd.lib.loss.loss_contrast_mem import ContrastCELoss as MemContrastCELoss
from contrastiveseg_amd.lib.loss.loss_helper import FSAuxCELoss, FSCELoss
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

SEG_LOSS_DICT = {
    'fs_ce_loss': FSCELoss,
    'fs_auxce_loss': FSAuxCELoss,
    'contrast_auxce_loss': ContrastAuxCELoss,
    'contrast_ce_loss': ContrastCELoss,
    'mem_contrast_ce_loss': MemContrastCELoss,
    'mem_contrast_auxce_loss': MemContrastAuxCELoss,      # not in the reference's table (loss_contrast_mem.py)
}


class LossManager(object):
    def __init__(self, configer):
        self.configer = configer

    def get_seg_loss(self, loss_type=None):
        key = self.configer.get('loss', 'loss_type') if loss_type is None else loss_type
        if key not in SEG_LOSS_DICT:
            Log.error('Loss: {} not valid!'.format(key))
            exit(1)
        Log.info('use loss: {}.'.format(key))
        # one process per GPU: no DataParallelCriterion wrap (reference :49-59 only wraps the legacy path)
        return SEG_LOSS_DICT[key](self.configer)
