"""Optimizer / lr-policy factory with the reference's config keys (segmentor/tools/optim_scheduler.py:46-143):
SGD / Adam / AdamW and the step, multistep, lambda_poly, lambda_cosine policies. SWA (torchcontrib) is outside the
hot path. SGD uses torch's fused update on the GPU (foreach on the CPU): 926 parameter tensors in a handful of launches."""
import math
import os

from torch.optim import SGD, Adam, AdamW, lr_scheduler

from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log


class OptimScheduler(object):
    def __init__(self, configer):
        self.configer = configer

    def init_optimizer(self, net_params):
        c = self.configer
        method = c.get('optim', 'optim_method')
        lr = c.get('lr', 'base_lr')
        if method == 'sgd':
            p = c.get('optim', 'sgd')
            # same update rule either way (torch.optim.SGD). Default where every parameter lives on the GPU: torch's fused implementation
            # (one kernel family per parameter group instead of three foreach passes over 926 tensors: 99.25 -> 96.7 ms per step of the
            # benched configuration, A/B/A/B on one MI355X, profiles/r03_fused_sgd_ab.txt); CSEG_FUSED_SGD=0 = the foreach form
            params = list(net_params)
            flat = [q for g in params for q in (g['params'] if isinstance(g, dict) else [g])]
            fused = os.environ.get('CSEG_FUSED_SGD', '1') == '1' and all(q.is_cuda for q in flat)
            optimizer = SGD(params, lr=lr, momentum=p['momentum'], weight_decay=p['weight_decay'], nesterov=p['nesterov'],
                            **({'fused': True} if fused else {'foreach': True}))
        elif method == 'adam':
            p = c.get('optim', 'adam')
            optimizer = Adam(net_params, lr=lr, betas=p['betas'], eps=p['eps'], weight_decay=p['weight_decay'])
        elif method == 'adamw':
            p = c.get('optim', 'adamw')
            optimizer = AdamW(net_params, lr=lr, betas=p['betas'], eps=p['eps'], weight_decay=p['weight_decay'])
        else:
            Log.error('Optimizer {} is not valid.'.format(method))
            exit(1)

        policy = c.get('lr', 'lr_policy')
        max_iters = c.get('solver', 'max_iters')
        if policy == 'step':
            scheduler = lr_scheduler.StepLR(optimizer, c.get('lr', 'step')['step_size'], gamma=c.get('lr', 'step')['gamma'])
        elif policy == 'multistep':
            scheduler = lr_scheduler.MultiStepLR(optimizer, c.get('lr', 'multistep')['stepvalue'],
                                                 gamma=c.get('lr', 'multistep')['gamma'])
        elif policy == 'lambda_poly':
            if os.environ.get('lambda_poly_power'):
                power = float(os.environ.get('lambda_poly_power'))
            elif c.exists('lr', 'lambda_poly'):
                power = c.get('lr', 'lambda_poly')['power']
            else:
                power = 0.9
            Log.info('Use lambda_poly policy with power {}'.format(power))
            scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda it: pow(1.0 - it / max_iters, power))
        elif policy == 'lambda_cosine':
            scheduler = lr_scheduler.LambdaLR(optimizer,
                                              lr_lambda=lambda it: (math.cos(math.pi * it / max_iters) + 1.0) / 2)
        else:
            Log.error('Policy:{} is not valid.'.format(policy))
            exit(1)
        return optimizer, scheduler
