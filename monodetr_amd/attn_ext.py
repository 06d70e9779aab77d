"""Host binding of the fused attention kernels (C ABI: mdetr_attn_forward / mdetr_attn_backward).

``fused_attention(q, k, v, num_heads, dropout_p, key_padding_mask)`` takes batch-first
``[B, L, H*32]`` tensors whose last dimension is contiguous (row / batch strides are free, so the
slices of a packed in-projection output are consumed in place) and returns ``[B, Lq, H*32]``.
Differentiable; the dropout mask is regenerated in the backward from (seed, b, h, q, k).
"""
import torch

from . import _capi


def _codes(t):
    if t.dtype == torch.float32:
        return _capi.MDETR_F32
    if t.dtype == torch.bfloat16:
        return _capi.MDETR_BF16
    raise RuntimeError("fused attention supports float32 and bfloat16, got %s" % t.dtype)


def _strided(t):
    """[B, L, E] with unit stride in E -> (tensor, batch stride, row stride); copies if it must."""
    if t.stride(-1) != 1 or (t.data_ptr() % 16) or (t.stride(1) * t.element_size()) % 16 or (t.stride(0) * t.element_size()) % 16:
        t = t.contiguous()
    return t, t.stride(0), t.stride(1)


_seed_state = {}
_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _where(t):
    return (t.device.index, torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else (-1, None)


_WEYL = 0x9E3779B97F4A7C15 - (1 << 64)      # 64-bit golden-ratio increment as a signed int64


def _next_seed(device):
    """Device-resident dropout counter: returns a snapshot tensor (int64 [1]) for this call and
    advances the counter, all with device ops -- so a captured hipGraph draws a new mask on every
    replay.  Initialised once from torch's CPU generator (reproducible under torch.manual_seed).
    The counter advances by the 64-bit golden ratio (a Weyl sequence), not by 1: the kernels' hash mixes
    the seed with a single multiply, and masks drawn from seeds that differ only in their lowest bit are
    correlated (-0.11); golden-ratio steps flip about half of the bits and give uncorrelated masks
    (tests/test_dropout_hash_cpu.py)."""
    st = _seed_state.get(device)
    if st is None:
        st = _seed_state[device] = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
    snap = st.clone()
    st.add_(_WEYL)                             # wraps modulo 2^64
    return snap


# ---- one device-side seed bump per training ITERATION instead of one snapshot + bump per dropout site -----------------------------
# Inside an iteration scope (helpers/step_helper.TrainIteration opens one around forward + backward) every dropout site draws
# (a host constant, the device-resident base): the kernels hash seed = constant + *base, the base advances ONCE per iteration
# (one launch; captured, so every replay advances it), the constant is the site's ordinal times a 64-bit odd multiplier (flips
# about half of the bits: see the caveat at `_next_seed`).  24 sites per step were 24 clones + 24 adds of an 8-byte tensor.
# The backward of a site re-reads the base: valid as long as the iteration's backward runs before the next `begin_iteration`
# (one forward + backward per scope, which is what TrainIteration does).
_scope = {}
_SITE = 0xD1B54A32D192ED03


def begin_iteration(device):
    device = torch.device(device)
    st = _seed_state.get(device)
    if st is None:
        st = _seed_state[device] = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
    st.add_(_WEYL)
    _scope[device] = 0


def end_iteration(device):
    _scope.pop(torch.device(device), None)


def site_seed(device):
    """(host seed, device seed tensor) of the next dropout site on `device`."""
    k = _scope.get(device)
    if k is None:
        return 0, _next_seed(device)
    _scope[device] = k + 1
    return ((k + 1) * _SITE) & 0x7FFFFFFFFFFFFFFF, _seed_state[device]


class _FusedAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, num_heads, scale, dropout_p, seed, key_padding_mask, seed_dev=None):
        assert q.is_cuda or _backend is not None, "fused attention runs on the GPU only"
        B, Lq, E = q.shape
        Lk = k.shape[1]
        assert E == num_heads * 32, "head_dim must be 32"
        assert k.shape == (B, Lk, E) and v.shape == (B, Lk, E) and k.dtype == q.dtype and v.dtype == q.dtype
        code = _codes(q)
        q, qb, qr = _strided(q)
        k, kb, kr = _strided(k)
        v, vb, vr = _strided(v)
        kpm = None
        if key_padding_mask is not None:
            kpm = key_padding_mask.to(torch.uint8).contiguous()
            assert kpm.shape == (B, Lk)
        out = torch.empty((B, Lq, E), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, num_heads, Lq), dtype=torch.float32, device=q.device)
        rc = _lib().mdetr_attn_forward(
            code, q.data_ptr(), k.data_ptr(), v.data_ptr(), kpm.data_ptr() if kpm is not None else None,
            out.data_ptr(), lse.data_ptr(), B, num_heads, Lq, Lk, qb, kb, vb, qr, kr, vr,
            float(scale), float(dropout_p), int(seed), seed_dev.data_ptr() if seed_dev is not None else None,
            *_where(q))
        _capi.check(rc, "mdetr_attn_forward")
        ctx.seed_dev = seed_dev
        ctx.save_for_backward(q, k, v, out, lse, kpm if kpm is not None else torch.empty(0, device=q.device))
        ctx.meta = (code, num_heads, float(scale), float(dropout_p), int(seed), kpm is not None, (qb, kb, vb, qr, kr, vr))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        q, k, v, out, lse, kpm = ctx.saved_tensors
        code, H, scale, dropout_p, seed, has_kpm, (qb, kb, vb, qr, kr, vr) = ctx.meta
        B, Lq, E = q.shape
        Lk = k.shape[1]
        d_out = d_out.contiguous()
        if d_out.data_ptr() % 16:
            d_out = d_out.clone()
        dq = torch.empty((B, Lq, E), dtype=q.dtype, device=q.device)
        dk = torch.empty((B, Lk, E), dtype=q.dtype, device=q.device)
        dv = torch.empty((B, Lk, E), dtype=q.dtype, device=q.device)
        dsum = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
        rc = _lib().mdetr_attn_backward(
            code, q.data_ptr(), k.data_ptr(), v.data_ptr(), kpm.data_ptr() if has_kpm else None,
            out.data_ptr(), d_out.data_ptr(), lse.data_ptr(), dsum.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Lq, Lk, qb, kb, vb, qr, kr, vr,
            scale, dropout_p, seed, ctx.seed_dev.data_ptr() if ctx.seed_dev is not None else None,
            *_where(q))
        _capi.check(rc, "mdetr_attn_backward")
        return dq, dk, dv, None, None, None, None, None, None


def fused_attention(q, k, v, num_heads, dropout_p=0.0, key_padding_mask=None, scale=None, seed=None):
    if scale is None:
        scale = (q.shape[-1] // num_heads) ** -0.5
    seed_dev = None
    if dropout_p > 0.0 and seed is None:
        seed, seed_dev = site_seed(q.device)     # explicit `seed` (tests) keeps the host-scalar path
    return _FusedAttention.apply(q, k, v, num_heads, scale, dropout_p, seed or 0, key_padding_mask, seed_dev)
