"""Optimizer factory and the reference's AdamW variant -- mirror of
lib/helpers/optimizer_helper.py (``build_optimizer`` :7-27, ``AdamW`` :30-129).

Semantics kept exactly (they are NOT torch.optim.AdamW's):
    m <- b1 m + (1-b1) g ;  v <- b2 v + (1-b2) g^2
    step = lr * sqrt(1 - b2^t) / (1 - b1^t)
    p <- p - step * (wd * p + m / (sqrt(v) + eps))            (reference :129)
i.e. the decoupled decay is scaled by the bias-corrected ``step``, eps is added outside the
bias correction, parameters whose name contains 'bias' get wd = 0 (:8-16), parameters without a
gradient are skipped (:79-80).

What changes is how it runs: the reference loops over a few hundred tensors in Python with ~8 tiny
kernels each (and deprecated ``add_(Number, Tensor)`` overloads, :111-112,129).  Here every update
is a handful of multi-tensor ``torch._foreach_*`` launches per parameter group, with the scalar
step size computed on the host (no device sync).  bf16 parameters (helpers/precision.py) are
updated through fp32 master copies kept in the optimizer state and rounded back after the step.

``capturable=True`` keeps the step count (and, if ``group['lr']`` is a tensor, the learning rate) on
the device and computes the bias-corrected step size there, so that ``step()`` can be recorded into
a hipGraph and replayed: a replay advances the count and applies the right bias correction without
any host code running.
"""
import math

import torch
import torch.optim as optim
from torch.optim.optimizer import Optimizer


def build_optimizer(cfg_optimizer, model):
    weights, biases = [], []
    for name, param in model.named_parameters():
        (biases if 'bias' in name else weights).append(param)
    parameters = [{'params': biases, 'weight_decay': 0},
                  {'params': weights, 'weight_decay': cfg_optimizer['weight_decay']}]
    kind = cfg_optimizer['type']
    if kind == 'sgd':
        return optim.SGD(parameters, lr=cfg_optimizer['lr'], momentum=0.9)
    if kind == 'adam':
        return optim.Adam(parameters, lr=cfg_optimizer['lr'])
    if kind == 'adamw':
        return AdamW(parameters, lr=cfg_optimizer['lr'], capturable=bool(cfg_optimizer.get('capturable', False)))
    raise NotImplementedError("%s optimizer is not supported" % kind)


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                 capturable=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad,
                                      capturable=capturable))

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault('amsgrad', False)
            group.setdefault('capturable', False)

    def _device_step_size(self, group, device, first_step):
        """lr * sqrt(1 - b2^t) / (1 - b1^t) as a device scalar.  t lives on the device, one counter
        per cohort of parameters that started stepping together (``group['step_dev']``, keyed by the
        host-side count at which the counter was created minus the cohort's own count)."""
        beta1, beta2 = group['betas']
        counters = group.setdefault('step_dev', {})
        key = group['calls'] - first_step          # constant over the life of a cohort
        t = counters.get(key)
        if t is None:
            t = counters[key] = torch.full((), float(first_step - 1), dtype=torch.float64, device=device)
        t.add_(1.0)
        lr = group['lr']
        lr = lr.to(device=device, dtype=torch.float64) if torch.is_tensor(lr) else float(lr)
        size = lr * torch.sqrt(1.0 - torch.pow(beta2, t)) / (1.0 - torch.pow(beta1, t))
        return size.to(torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            group['calls'] = group.get('calls', 0) + 1
            # parameters of one group may have taken different numbers of steps (a parameter that
            # first gets a gradient later): bucket by step count so the bias correction stays scalar
            buckets = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    wide = p.dtype if p.dtype in (torch.float32, torch.float64) else torch.float32
                    st['exp_avg'] = torch.zeros_like(p, dtype=wide)
                    st['exp_avg_sq'] = torch.zeros_like(p, dtype=wide)
                    if group['amsgrad']:
                        st['max_exp_avg_sq'] = torch.zeros_like(p, dtype=wide)
                    if wide != p.dtype:                   # bf16 / fp16 parameter: fp32 master + staging for the gradient
                        st['master'] = p.detach().to(wide)
                        st['grad32'] = torch.zeros_like(p, dtype=wide)
                st['step'] += 1
                buckets.setdefault(st['step'], []).append(p)
            for t, ps_model in buckets.items():
                # bf16 parameters are updated through their fp32 master copies
                low = [p for p in ps_model if p.dtype != torch.float32 and p.dtype != torch.float64]
                ps = [self.state[p]['master'] if p.dtype not in (torch.float32, torch.float64) else p for p in ps_model]
                grads = [p.grad for p in ps_model]
                if low:
                    g32 = [self.state[p]['grad32'] for p in low]
                    torch._foreach_copy_(g32, [p.grad for p in low])
                    it = iter(g32)
                    grads = [next(it) if p.dtype not in (torch.float32, torch.float64) else p.grad for p in ps_model]
                m = [self.state[p]['exp_avg'] for p in ps_model]
                v = [self.state[p]['exp_avg_sq'] for p in ps_model]
                torch._foreach_mul_(m, beta1)
                torch._foreach_add_(m, grads, alpha=1 - beta1)
                torch._foreach_mul_(v, beta2)
                torch._foreach_addcmul_(v, grads, grads, value=1 - beta2)
                if group['amsgrad']:
                    vmax = [self.state[p]['max_exp_avg_sq'] for p in ps_model]
                    torch._foreach_maximum_(vmax, v)
                    denom = torch._foreach_sqrt(vmax)
                else:
                    denom = torch._foreach_sqrt(v)
                torch._foreach_add_(denom, group['eps'])
                upd = torch._foreach_div(m, denom)
                if group['weight_decay'] != 0:
                    torch._foreach_add_(upd, ps, alpha=group['weight_decay'])
                if group['capturable']:
                    torch._foreach_mul_(upd, self._device_step_size(group, ps[0].device, t))
                    torch._foreach_sub_(ps, upd)
                else:
                    step = group['lr'] * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
                    torch._foreach_add_(ps, upd, alpha=-step)
                if low:
                    torch._foreach_copy_(low, [self.state[p]['master'] for p in low])
        return loss
