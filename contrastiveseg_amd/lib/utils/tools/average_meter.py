"""Running meter with the attribute names the reference's log line formats (`val`, `avg`, `sum`, `count`;
lib/utils/tools/average_meter.py, used at segmentor/trainer_contrastive.py:28-34, 270-289)."""


class AverageMeter(object):
    __slots__ = ('val', 'avg', 'sum', 'count')

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.count += n
        self.sum += val * n
        self.avg = self.sum / max(self.count, 1)

    def __repr__(self):
        return 'AverageMeter(val={:.6g}, avg={:.6g}, n={})'.format(self.val, self.avg, self.count)
