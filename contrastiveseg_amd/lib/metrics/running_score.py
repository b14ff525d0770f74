"""RunningScore with the reference's interface and arithmetic (lib/metrics/running_score.py:120-215): a K x K confusion
matrix accumulated over (prediction, ground truth) label maps, mean IoU = nanmean over classes of
diag / (row + column - diag), pixel accuracy, per-class accuracy, frequency-weighted IoU.

What changes underneath: the matrix is an int64 tensor that lives where the label maps live (on the GPU during
validation: no per-image .cpu().numpy() round trip, one bincount per batch instead of one per image), integer counts
are exact, and reduce_scores() is an all-reduce over RCCL when a process group exists (reference :163-168 goes through
numpy). Scores are evaluated in fp64 on the host from the (tiny) matrix, with numpy semantics for empty classes (nan,
skipped by nanmean) exactly like the reference."""
import numpy as np
import torch

from contrastiveseg_amd.lib.utils import distributed as D


class RunningScore(object):
    def __init__(self, configer=None, num_classes=None, ignore_index=None):
        self.configer = configer
        self.n_classes = configer.get('data', 'num_classes') if num_classes is None else num_classes
        self.ignore_index = ignore_index
        self.confusion_matrix = None            # int64 [K, K], rows = ground truth, columns = prediction
        self.reduced_confusion_matrix = None

    def _fast_hist(self, label_true, label_pred):
        n = self.n_classes
        mask = (label_true >= 0) & (label_true < n) & (label_pred >= 0) & (label_pred < n)     # reference :143-144
        if self.ignore_index is not None:
            mask = mask & (label_true != self.ignore_index)
        idx = n * label_true[mask].long() + label_pred[mask].long()
        return torch.bincount(idx, minlength=n * n).reshape(n, n)

    def update(self, label_preds, label_trues):
        """label_preds / label_trues: integer tensors of identical shape (any leading batch dimension)."""
        self.reduced_confusion_matrix = None
        hist = self._fast_hist(label_trues.reshape(-1), label_preds.reshape(-1))
        self.confusion_matrix = hist if self.confusion_matrix is None else self.confusion_matrix + hist

    def reduce_scores(self):
        hist = self.confusion_matrix
        if hist is None:
            hist = torch.zeros(self.n_classes, self.n_classes, dtype=torch.int64)
        if D.is_distributed() and D.get_world_size() > 1:
            import torch.distributed as dist
            # RCCL reduces device tensors only; gloo takes either
            hist = hist.cuda() if dist.get_backend() == "nccl" and not hist.is_cuda else hist.clone()
            dist.all_reduce(hist)
        self.reduced_confusion_matrix = hist.cpu().numpy().astype(np.float64)

    def _get_scores(self):
        if self.reduced_confusion_matrix is None:
            self.reduce_scores()
        hist = self.reduced_confusion_matrix
        with np.errstate(divide='ignore', invalid='ignore'):
            acc = np.diag(hist).sum() / hist.sum()
            acc_cls_list = np.diag(hist) / hist.sum(axis=1)
            iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
            mean_iu = np.nanmean(iu)
            freq = hist.sum(axis=1) / hist.sum()
            fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        return acc, acc_cls_list, fwavacc, mean_iu, dict(zip(range(self.n_classes), iu))

    def get_mean_iou(self):
        return self._get_scores()[3]

    def get_pixel_acc(self):
        return self._get_scores()[0]

    def get_mean_acc(self):
        return self._get_scores()[1]

    def get_cls_iou(self):
        return self._get_scores()[-1]

    def reset(self):
        self.confusion_matrix = None
        self.reduced_confusion_matrix = None
