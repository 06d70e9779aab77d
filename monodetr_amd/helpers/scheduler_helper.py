"""Learning-rate schedule -- mirror of ``lib/helpers/scheduler_helper.py``: step decay by epoch
(``decay_list`` / ``decay_rate``) and an optional 5-epoch cosine warm-up from 1e-5.  (The batch-norm momentum
scheduler of that file has nothing to act on: every BatchNorm of this model is frozen.)"""
import math

import torch.optim.lr_scheduler as lr_sched


class CosineWarmupLR(lr_sched.LRScheduler):
    """lr(e) = init + (base - init) * (1 - cos(pi e / num_epoch)) / 2."""

    def __init__(self, optimizer, num_epoch, init_lr=0.0, last_epoch=-1):
        self.num_epoch, self.init_lr = num_epoch, init_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        ramp = (1 - math.cos(math.pi * self.last_epoch / self.num_epoch)) / 2
        return [self.init_lr + (base - self.init_lr) * ramp for base in self.base_lrs]


class LinearWarmupLR(lr_sched.LRScheduler):
    def __init__(self, optimizer, num_epoch, init_lr=0.0, last_epoch=-1):
        self.num_epoch, self.init_lr = num_epoch, init_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [self.init_lr + (base - self.init_lr) * self.last_epoch / self.num_epoch for base in self.base_lrs]


def build_lr_scheduler(cfg, optimizer, last_epoch):
    def decay(epoch):
        passed = sum(1 for step in cfg['decay_list'] if epoch >= step)
        return cfg['decay_rate'] ** passed

    lr_scheduler = lr_sched.LambdaLR(optimizer, decay, last_epoch=last_epoch)
    warmup = CosineWarmupLR(optimizer, num_epoch=5, init_lr=0.00001) if cfg['warmup'] else None
    return lr_scheduler, warmup
