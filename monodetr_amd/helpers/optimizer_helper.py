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
import os

import torch

from .. import _tune
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
        cls = FusedAdamW if cfg_optimizer.get('fused', False) else AdamW
        return cls(parameters, lr=cfg_optimizer['lr'], capturable=bool(cfg_optimizer.get('capturable', False)))
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

    def load_state_dict(self, state_dict):
        """torch's loader casts every floating-point state tensor to its parameter's dtype, which would
        round the fp32 moments and master copy of a bf16 parameter to bf16: put the tensors back as saved."""
        from itertools import chain
        super().load_state_dict(state_dict)
        saved_ids = chain.from_iterable(g['params'] for g in state_dict['param_groups'])
        own = chain.from_iterable(g['params'] for g in self.param_groups)
        for pid, p in zip(saved_ids, own):
            for k, v in state_dict['state'].get(pid, {}).items():
                if torch.is_tensor(v) and v.is_floating_point() and k != 'step':
                    self.state[p][k] = v.detach().clone().to(device=p.device)

    # ---- graph replay (helpers/step_helper.TrainIteration): a replayed step advances the device counters only -------------
    def note_replay(self):
        """A captured ``step()`` was replayed: the host-side step counts (what ``state_dict()`` saves) are brought up to date
        lazily by ``flush_replays``."""
        self._pending_replays = getattr(self, "_pending_replays", 0) + 1

    def uncount_step(self):
        """The host bookkeeping of one ``step()`` whose kernels did not run (it was being captured) is taken back."""
        self._advance_host_counts(-1)

    def flush_replays(self):
        n, self._pending_replays = getattr(self, "_pending_replays", 0), 0
        if n:
            self._advance_host_counts(n)

    def _advance_host_counts(self, n):
        for group in self.param_groups:
            if 'calls' in group:
                group['calls'] += n
        for st in self.state.values():
            if 'step' in st:
                st['step'] += n

    def state_dict(self):
        """Device-only bookkeeping stays out of checkpoints (they remain interchangeable with the reference's): a
        device-resident learning rate is saved as a float, the device step counters are rebuilt from the saved counts."""
        self.flush_replays()
        sd = super().state_dict()
        for g in sd['param_groups']:
            g.pop('step_dev', None)
            if torch.is_tensor(g.get('lr')):
                g['lr'] = float(g['lr'])
        return sd

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
        self.flush_replays()
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


def _same_layout(g, p):
    """Do g and p place their elements in the same memory order?  Strides of size-1 axes say nothing (a [256, 512, 1, 1] weight
    is the same bytes with contiguous and with channels-last strides)."""
    return g.shape == p.shape and all(a == b for n, a, b in zip(p.shape, g.stride(), p.stride()) if n != 1)


class FusedAdamW(AdamW):
    """The same update as :class:`AdamW` with ONE kernel launch per (parameter group, dtype)
    (csrc/adamw.hip through ``mdetr_adamw_step``) instead of ~40 multi-tensor launches per group.

    At the first ``step()`` the parameters that receive gradients are re-homed into one flat buffer per
    (group, dtype) -- each ``p.data`` becomes a view with its original shape and strides, segments
    start on 256-byte boundaries -- together with flat fp32 ``exp_avg`` / ``exp_avg_sq`` (and fp32
    master copies of bf16 parameters); ``state[p]`` holds views of those, so ``state_dict()`` keeps the
    reference's per-parameter layout.  Each step copies the gradients into a flat staging buffer with one
    multi-tensor copy and launches the update.  CUDA parameters only; amsgrad, sparse gradients and CPU
    parameters take the parent's path.  """

    _GATHER_CHUNK = 32768        # bytes of one tensor a workgroup of the gradient gather copies
    _PAD = 64                                    # elements: 256 B for fp32 segments, 128 B for bf16

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._flat = None
        self._lib = None                         # tests substitute a host build of the same kernel arithmetic
        self._allow_cpu = False
        # _gather = True: the update reads every gradient where autograd left it (csrc/adamw.hip adamw_gathered_kernel over a block
        # table, base addresses as kernel arguments) -- no flat gradient buffer and no multi-tensor copy into one (7 launches, 0.12 ms
        # per iteration).  Off = the flat form.  (Round 3's gather INTO the flat buffer ahead of the flat update moved the same bytes
        # as the copy it replaced and measured the same: profiles/r03g2_bench_*.json, r06p_.)
        self._gather = _tune.get("adamw_gather", "1") != "0"           # (MDETR_TUNE=adamw_gather=0: A-B runs)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # AdamW.load_state_dict: keeps the saved dtypes
        self._flat = None                        # loaded state tensors are not views of the flat buffers: rebuild

    def _advance_host_counts(self, n):
        super()._advance_host_counts(n)
        for buf in (self._flat[1] if self._flat is not None else ()):
            buf['step'] += n

    def _eligible(self):
        for group in self.param_groups:
            if group['amsgrad']:
                return False
            for p in group['params']:
                if p.grad is not None and ((not p.is_cuda and not self._allow_cpu) or p.grad.is_sparse
                                           or p.dtype not in (torch.float32, torch.bfloat16)):
                    return False
        return True

    @staticmethod
    def _dense(t):
        return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))

    def _signature(self):
        return tuple(tuple(id(p) for p in g['params'] if p.grad is not None) for g in self.param_groups)

    @torch.no_grad()
    def _build_flat(self):
        flat = []
        for gi, group in enumerate(self.param_groups):
            # one flat buffer per (dtype, number of steps already taken): a parameter that first receives a
            # gradient later has its own bias correction, as in the parent class
            by_dtype = {}
            for p in group['params']:
                if p.grad is not None:
                    by_dtype.setdefault((p.dtype, int(self.state[p].get('step', 0))), []).append(p)
            for (dtype, _), plist in by_dtype.items():
                offs, total = [], 0
                for p in plist:
                    offs.append(total)
                    total += -(-p.numel() // self._PAD) * self._PAD
                dev = plist[0].device
                buf = dict(group=gi, dtype=dtype, params=plist, n=total,
                           param=torch.zeros(total, dtype=dtype, device=dev),
                           grad=torch.zeros(total, dtype=dtype, device=dev),
                           exp_avg=torch.zeros(total, dtype=torch.float32, device=dev),
                           exp_avg_sq=torch.zeros(total, dtype=torch.float32, device=dev))
                buf['master'] = buf['param'] if dtype == torch.float32 else torch.zeros(total, dtype=torch.float32, device=dev)
                grad_views, steps = [], set()
                for p, off in zip(plist, offs):
                    if not self._dense(p.data):
                        p.data = p.data.contiguous()
                    shape, stride, n = tuple(p.shape), tuple(p.stride()), p.numel()

                    def view(t, off=off, shape=shape, stride=stride):
                        return t.as_strided(shape, stride, off)
                    st = self.state[p]
                    view(buf['param']).copy_(p.data)
                    if dtype != torch.float32:
                        view(buf['master']).copy_(st['master'] if 'master' in st else p.data)
                    if 'exp_avg' in st:                         # carry an existing state over (load_state_dict, late switch)
                        view(buf['exp_avg']).copy_(st['exp_avg'])
                        view(buf['exp_avg_sq']).copy_(st['exp_avg_sq'])
                    p.data = view(buf['param'])
                    st['exp_avg'], st['exp_avg_sq'] = view(buf['exp_avg']), view(buf['exp_avg_sq'])
                    if dtype != torch.float32:
                        st['master'] = view(buf['master'])
                        st.pop('grad32', None)
                    st.setdefault('step', 0)
                    steps.add(int(st['step']))
                    grad_views.append(view(buf['grad']))
                if len(steps) != 1:
                    raise RuntimeError("FusedAdamW: the parameters of a flat group must have taken the same number of steps")
                buf['grad_views'], buf['step'] = grad_views, steps.pop()
                if True:
                    # tables of the gathered update (csrc/adamw.hip adamw_gathered_kernel): one 32 KB chunk of one tensor per workgroup
                    esz = buf['grad'].element_size()
                    chunk, bt, bs = self._GATHER_CHUNK, [], []
                    for i, (p, off) in enumerate(zip(plist, offs)):
                        for s0 in range(0, p.numel() * esz, chunk):
                            bt.append(i)
                            bs.append(s0)
                    begin, nb = [], 0
                    for i, p in enumerate(plist):
                        begin.append(nb)
                        nb += -(-(p.numel() * esz) // chunk)
                    begin.append(nb)
                    import ctypes
                    buf['gather'] = dict(
                        dst_off=torch.tensor([o * esz for o in offs], dtype=torch.int64, device=dev),
                        nbytes=torch.tensor([p.numel() * esz for p in plist], dtype=torch.int64, device=dev),
                        blk_tensor=torch.tensor(bt, dtype=torch.int32, device=dev), blk_start=torch.tensor(bs, dtype=torch.int64, device=dev),
                        begin=(ctypes.c_int * len(begin))(*begin), ptrs=(ctypes.c_void_p * len(plist))())
                flat.append(buf)
        self._flat = (self._signature(), flat)

    @torch.no_grad()
    def step(self, closure=None):
        if not self._eligible():
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.flush_replays()
        if self._flat is None or self._flat[0] != self._signature():
            self._build_flat()
        from .. import _capi
        lib = self._lib if self._lib is not None else _capi.lib()
        for buf in self._flat[1]:
            group = self.param_groups[buf['group']]
            beta1, beta2 = group['betas']
            grads = []
            for p in buf['params']:
                g = p.grad
                if g.stride() != p.stride():          # (the multi-tensor copy leaves its fused path on any stride mismatch)
                    g = g.as_strided(p.shape, p.stride(), g.storage_offset()) if _same_layout(g, p) else torch.empty_like(p).copy_(g)
                grads.append(g)
            buf['step'] += 1
            t = buf['step']
            for p in buf['params']:
                self.state[p]['step'] = t
            dev = buf['param'].device
            dtype_code = _capi.MDETR_BF16 if buf['dtype'] == torch.bfloat16 else _capi.MDETR_F32
            dev_index = dev.index if dev.type == "cuda" else -1
            stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
            n_no_decay = buf['n'] if group['weight_decay'] == 0 else 0
            # where the step size comes from: a device-resident step count (capturable: one launch, nothing else), a device scalar,
            # or the host
            tdev = step_dev = None
            lr = group['lr']
            lr_dev = lr if torch.is_tensor(lr) and lr.dtype == torch.float64 and lr.device == dev else None
            counted = getattr(lib, "mdetr_adamw_step_counted", None)
            if group['capturable']:
                group['calls'] = group.get('calls', 0)            # the device counters are keyed like the parent's
                if counted is not None:
                    tdev = buf.get('step_dev')
                    if tdev is None:
                        tdev = buf['step_dev'] = torch.full((), float(t - 1), dtype=torch.float64, device=dev)
                    tdev.add_(1.0)
                else:
                    step_dev = self._fused_step_size(group, buf, t)
            step = float(lr) * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t) if (tdev is None and step_dev is None) else 0.0
            lr_host = 0.0 if (lr_dev is not None or tdev is None) else float(lr)
            gather = buf.get('gather') if self._gather else None
            gathered = getattr(lib, "mdetr_adamw_step_gathered", None) if gather is not None else None
            if gathered is not None:
                # the kernel reads every gradient where autograd left it (base addresses as kernel arguments: autograd allocates them anew
                # each iteration, a captured graph bakes them into its node): no flat gradient buffer, no copy into one
                ptrs = gather['ptrs']
                for i, g in enumerate(grads):
                    ptrs[i] = g.data_ptr()
                rc = gathered(dtype_code, buf['param'].data_ptr(), buf['master'].data_ptr(), ptrs, len(grads), gather['begin'],
                              gather['dst_off'].data_ptr(), gather['nbytes'].data_ptr(), gather['blk_tensor'].data_ptr(), gather['blk_start'].data_ptr(),
                              self._GATHER_CHUNK, buf['exp_avg'].data_ptr(), buf['exp_avg_sq'].data_ptr(), n_no_decay,
                              beta1, beta2, group['eps'], group['weight_decay'], step, step_dev.data_ptr() if step_dev is not None else None,
                              tdev.data_ptr() if tdev is not None else None, lr_host, lr_dev.data_ptr() if (lr_dev is not None and tdev is not None) else None,
                              dev_index, stream)
                if rc != 0:
                    _capi.check(rc, "mdetr_adamw_step_gathered")
                buf['_keep'] = grads                              # the launch reads them: alive until the next step replaces the list
                continue
            torch._foreach_copy_(buf['grad_views'], grads)
            if tdev is not None:
                rc = counted(dtype_code, buf['param'].data_ptr(), buf['master'].data_ptr(), buf['grad'].data_ptr(),
                             buf['exp_avg'].data_ptr(), buf['exp_avg_sq'].data_ptr(), buf['n'], n_no_decay, beta1, beta2, group['eps'],
                             group['weight_decay'], tdev.data_ptr(), lr_host, lr_dev.data_ptr() if lr_dev is not None else None, dev_index, stream)
                if rc != 0:
                    _capi.check(rc, "mdetr_adamw_step_counted")
                continue
            rc = lib.mdetr_adamw_step(dtype_code, buf['param'].data_ptr(), buf['master'].data_ptr(), buf['grad'].data_ptr(),
                                      buf['exp_avg'].data_ptr(), buf['exp_avg_sq'].data_ptr(), buf['n'], n_no_decay, beta1, beta2,
                                      group['eps'], group['weight_decay'], step, step_dev.data_ptr() if step_dev is not None else None,
                                      dev_index, stream)
            if rc != 0:
                _capi.check(rc, "mdetr_adamw_step")
        return loss

    def _fused_step_size(self, group, buf, first_step):
        """Device-resident bias-corrected step size of one flat buffer (capturable mode)."""
        beta1, beta2 = group['betas']
        t = buf.get('step_dev')
        if t is None:
            t = buf['step_dev'] = torch.full((), float(first_step - 1), dtype=torch.float64, device=buf['param'].device)
        t.add_(1.0)
        lr = group['lr']
        lr = lr.to(device=t.device, dtype=torch.float64) if torch.is_tensor(lr) else float(lr)
        return (lr * torch.sqrt(1.0 - torch.pow(beta2, t)) / (1.0 - torch.pow(beta1, t))).to(torch.float32)
