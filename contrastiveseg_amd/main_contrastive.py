"""Command line of the contrastive trainer, same flags and override scheme as the reference's main_contrastive.py
(:31-222): `--configs X.json --phase train --model_name ... --loss_type ... [section.key value ...]`; argparse
destinations of the form 'section:key' override the JSON when given, trailing free-form pairs are literal_eval'ed.

Launch, MI355X-first: one process per GPU on the 'nccl' (= RCCL) backend. Either the reference's way --
`python -m contrastiveseg_amd.main_contrastive --distributed --gpu 0 1 2 3 ...` respawns itself with one rank per
listed GPU (lib/utils/distributed.py:handle_distributed, reference lib/utils/distributed.py:27-69) -- or ranks started
by torchrun (`python -m torch.distributed.run --nproc-per-node N -m contrastiveseg_amd.main_contrastive ...`), which are
recognised by RANK/LOCAL_RANK/WORLD_SIZE in the environment. Without `--distributed` the process is restricted to the
GPUs listed by `--gpu` (first one used), like the reference.

`--cudnn` (reference :155, default True there = cudnn.benchmark) maps to MIOpen's exhaustive find on ROCm, which costs
a 20+ minute warm-up per fresh machine for this network; the default here is False: MIOpen immediate mode plus the
tuned solver records shipped in contrastiveseg_amd/miopen_db (what bench.py measures). Passing `--cudnn true` turns the
find on and says so."""
import argparse
import os
import random

import torch

from contrastiveseg_amd.lib.utils.distributed import handle_distributed
from contrastiveseg_amd.lib.utils.tools.configer import Configer
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


def str2bool(v):
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Unsupported value encountered.')


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--configs', default=None, type=str, dest='configs')
    p.add_argument('--phase', default='train', type=str, dest='phase')
    p.add_argument('--gpu', default=[0], nargs='+', type=int, dest='gpu')
    p.add_argument('--seed', default=304, type=int)
    p.add_argument('--cudnn', type=str2bool, nargs='?', const=True, default=False)
    p.add_argument('--distributed', action='store_true', dest='distributed')
    p.add_argument('--local_rank', type=int, default=-1, dest='local_rank')
    # section:key overrides (None = keep the JSON value)
    for flag, dest, typ in [('--train_batch_size', 'train:batch_size', int), ('--val_batch_size', 'val:batch_size', int),
                            ('--model_name', 'network:model_name', str), ('--backbone', 'network:backbone', str),
                            ('--bn_type', 'network:bn_type', str), ('--pretrained', 'network:pretrained', str),
                            ('--resume', 'network:resume', str), ('--resume_strict', 'network:resume_strict', str2bool),
                            ('--resume_continue', 'network:resume_continue', str2bool),
                            ('--resume_val', 'network:resume_val', str2bool),
                            ('--base_lr', 'lr:base_lr', float), ('--nbb_mult', 'lr:nbb_mult', float),
                            ('--lr_policy', 'lr:lr_policy', str), ('--is_warm', 'lr:is_warm', str2bool),
                            ('--loss_type', 'loss:loss_type', str), ('--max_iters', 'solver:max_iters', int),
                            ('--display_iter', 'solver:display_iter', int), ('--test_interval', 'solver:test_interval', int),
                            ('--checkpoints_root', 'checkpoints:checkpoints_root', str),
                            ('--checkpoints_name', 'checkpoints:checkpoints_name', str),
                            ('--log_file', 'logging:log_file', str), ('--stdout_level', 'logging:stdout_level', str),
                            ('--optim_method', 'optim:optim_method', str), ('--group_method', 'optim:group_method', str)]:
        p.add_argument(flag, default=None, type=typ, dest=dest)
    p.add_argument('REMAIN', nargs='*')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    handle_distributed(args, module='contrastiveseg_amd.main_contrastive', argv=argv)
    if args.seed is not None:
        random.seed(args.seed)
        torch.manual_seed(args.seed)      # every rank seeds identically, like the reference (:169-171)
    torch.backends.cudnn.enabled = True
    torch.backends.cudnn.benchmark = bool(args.cudnn)   # MIOpen find mode on ROCm
    configer = Configer(args_parser=args)
    configer.add(['project_dir'], os.getcwd())
    lg = configer.get('logging') if configer.exists('logging') else {}
    Log.init(logfile_level=lg.get('logfile_level', 'info'), stdout_level=lg.get('stdout_level', 'info'),
             log_file=lg.get('log_file'), log_format=lg.get('log_format'), rewrite=lg.get('rewrite', False))
    if args.cudnn:
        Log.warn('--cudnn true: MIOpen exhaustive find is ON (20+ min of warm-up on a fresh machine for HRNet-W48); the '
                 'default (off) uses immediate mode + the tuned records in contrastiveseg_amd/miopen_db')
    if configer.get('phase') != 'train':
        Log.error('Phase: {} is outside the accelerated hot path (train only).'.format(configer.get('phase')))
        raise SystemExit(1)
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    Trainer(configer).train()


if __name__ == '__main__':
    main()
