"""``residual_layernorm(a, b, norm, dropout)``: ``norm(a + dropout(b))`` in one launch each way
(csrc/add_ln.hip through ``mdetr_add_layernorm_forward / _backward``) -- the model's ten residual sites.
The dropout decision is a hash of (device-resident seed, element index); no mask tensor exists."""
import os

import torch
import torch.nn.functional as F

from . import _capi
from .attn_ext import _next_seed

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# MDETR_FUSED_LN=1 routes the residual sites through the kernel; on the committed list of kernel_families.py
ENABLED = os.environ.get("MDETR_FUSED_LN") == "1"


_seed_counter = None


def _host_seed():
    """A new 64-bit seed per call: a Weyl sequence started from torch's CPU generator (reproducible under
    torch.manual_seed), as attn_ext._next_seed does on the device."""
    global _seed_counter
    if _seed_counter is None:
        _seed_counter = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
    _seed_counter = (_seed_counter + 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF      # 63 bits: autograd Function arguments must fit int64 (torch.profiler records them)
    return _seed_counter


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(a, b, norm):
    return ((a.is_cuda or _backend is not None) and a.dtype in (torch.float32, torch.bfloat16) and b.dtype == a.dtype
            and a.shape == b.shape and a.shape[-1] in (128, 256, 512) and len(norm.normalized_shape) == 1
            and norm.normalized_shape[0] == a.shape[-1] and norm.elementwise_affine and norm.bias is not None)


def _rows(t, C):
    """[rows, C] contiguous with a 16-byte aligned base (views at odd storage offsets are copied)."""
    t = t.reshape(-1, C).contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _dev(t):
    cuda = t.is_cuda
    return (t.device.index if cuda else -1), (torch.cuda.current_stream(t.device).cuda_stream if cuda else None)


def _column_sum_f32(x, out_dtype=torch.float32):
    """fp32 [rows, n] -> [n] (the partial gamma / beta sums; at most 1024 rows), written as ``out_dtype``."""
    if x.is_cuda and _backend is None:
        from .colsum_ext import column_sum
        return column_sum(x, out_dtype if out_dtype in (torch.float32, torch.bfloat16) else torch.float32)
    return x.sum(0).to(out_dtype)


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps, p, seed, seed_dev):
        shape = a.shape
        C = shape[-1]
        a2, b2 = _rows(a, C), _rows(b, C)
        rows = a2.shape[0]
        if gamma.dtype == beta.dtype and gamma.dtype in (torch.float32, torch.bfloat16):
            gp, bp = gamma.contiguous(), beta.contiguous()           # read as they are: no cast launches
        else:
            gp, bp = gamma.float().contiguous(), beta.float().contiguous()
        pcode = _capi.MDETR_BF16 if gp.dtype == torch.bfloat16 else _capi.MDETR_F32
        y, s = torch.empty_like(a2), torch.empty_like(a2)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=a.device)
        io = _capi.MDETR_BF16 if a.dtype == torch.bfloat16 else _capi.MDETR_F32
        dev, stream = _dev(a2)
        rc = _lib().mdetr_add_layernorm_forward(io, pcode, a2.data_ptr(), b2.data_ptr(), gp.data_ptr(), bp.data_ptr(), y.data_ptr(), s.data_ptr(),
                                                stats.data_ptr(), rows, C, float(eps), float(p), int(seed),
                                                seed_dev.data_ptr() if seed_dev is not None else None, dev, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_add_layernorm_forward")
        ctx.save_for_backward(s, gp, stats)
        ctx.seed_dev = seed_dev
        ctx.meta = (io, pcode, rows, C, float(p), int(seed), shape, gamma.dtype, beta.dtype)
        return y.view(shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        s, gp, stats = ctx.saved_tensors
        io, pcode, rows, C, p, seed, shape, g_dtype, b_dtype = ctx.meta
        dy2 = _rows(dy, C)
        da, db = torch.empty_like(s), torch.empty_like(s)
        lib = _lib()
        nb = lib.mdetr_add_layernorm_partial_rows(rows)
        partial = torch.empty((nb, 2 * C), dtype=torch.float32, device=s.device)
        dev, stream = _dev(s)
        rc = lib.mdetr_add_layernorm_backward(io, pcode, dy2.data_ptr(), s.data_ptr(), gp.data_ptr(), stats.data_ptr(), da.data_ptr(), db.data_ptr(),
                                              partial.data_ptr(), rows, C, p, seed,
                                              ctx.seed_dev.data_ptr() if ctx.seed_dev is not None else None, dev, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_add_layernorm_backward")
        sums = _column_sum_f32(partial, g_dtype if g_dtype == b_dtype else torch.float32)   # written in the parameters' dtype
        return da.view(shape), db.view(shape), sums[:C].to(g_dtype), sums[C:].to(b_dtype), None, None, None, None


def fused_add_layernorm(a, b, gamma, beta, eps=1e-5, dropout_p=0.0, seed=None):
    """LayerNorm(a + dropout(b)) over the last dimension (128, 256 or 512 wide); ``seed`` fixes the mask (tests),
    otherwise every call draws a fresh device-resident seed."""
    seed_dev = None
    if dropout_p > 0.0 and seed is None:
        if a.is_cuda and torch.cuda.is_current_stream_capturing():
            from .attn_ext import site_seed
            seed, seed_dev = site_seed(a.device)            # a replayed graph needs a seed that lives on the device
        else:
            seed = _host_seed()                             # eager: a host integer costs no launch
    return _AddLayerNorm.apply(a, b, gamma, beta, eps, dropout_p, seed or 0, seed_dev)


def residual_layernorm(a, b, norm, dropout):
    """``norm(a + dropout(b))`` for an ``nn.LayerNorm`` and an ``nn.Dropout`` module -- the fused kernel when
    ``MDETR_FUSED_LN=1`` and the shapes allow, the three framework operators otherwise."""
    if ENABLED and supported(a, b, norm):
        p = dropout.p if (dropout is not None and dropout.training) else 0.0
        return fused_add_layernorm(a, b, norm.weight, norm.bias, norm.eps, p)
    return norm(a + (dropout(b) if dropout is not None else b))
