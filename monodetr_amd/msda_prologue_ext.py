"""``msda_prologue(offsets, logits, reference_points, spatial_shapes)`` -> fp32 ``(sampling_locations,
attention_weights)``: the softmax and location arithmetic of MSDeformAttn.forward in one launch each way
(csrc/msda_prologue.hip through ``mdetr_msda_prologue_forward / _backward``)."""
import torch

from . import _capi

_backend = None               # tests substitute the host build of the same arithmetic (tests/native)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _io_code(dtype):
    return _capi.MDETR_BF16 if dtype == torch.bfloat16 else _capi.MDETR_F32


def supported(offsets, logits, reference_points):
    return ((offsets.is_cuda or _backend is not None) and offsets.dtype in (torch.float32, torch.bfloat16) and logits.dtype == offsets.dtype
            and reference_points.shape[-1] in (2, 6) and reference_points.stride(-1) == 1)


class _Prologue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, offsets, logits, ref, spatial_shapes):
        B, Lq, M, L, P, _ = offsets.shape
        R = ref.shape[-1]
        io = offsets.dtype
        off, lg = offsets.contiguous(), logits.contiguous()
        refc = ref if ref.dtype in (torch.float32, torch.bfloat16) else ref.float()     # read in its own dtype: fp32 coordinates stay fp32
        if refc.stride(-1) != 1:
            refc = refc.contiguous()
        rcode = _io_code(refc.dtype)
        dev = off.device
        loc = torch.empty((B, Lq, M, L, P, 2), dtype=torch.float32, device=dev)
        attn = torch.empty((B, Lq, M, L, P), dtype=torch.float32, device=dev)
        geom = (B, Lq, M, L, P, R, refc.stride(0), refc.stride(1), refc.stride(2))
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        rc = _lib().mdetr_msda_prologue_forward(_io_code(io), rcode, off.data_ptr(), lg.data_ptr(), refc.data_ptr(),
                                                spatial_shapes.data_ptr(), loc.data_ptr(), attn.data_ptr(), *geom,
                                                dev.index if dev.type == "cuda" else -1, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_msda_prologue_forward")
        ctx.save_for_backward(off, refc, spatial_shapes, attn)
        ctx.geom, ctx.io, ctx.ref_dtype, ctx.rcode = geom, io, ref.dtype, rcode
        ctx.ref_shape = tuple(ref.shape)
        ctx.mark_non_differentiable(spatial_shapes)
        return loc, attn

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loc, g_attn):
        off, refc, shapes, attn = ctx.saved_tensors
        B, Lq, M, L, P, R = ctx.geom[:6]
        dev = off.device
        g_off = torch.empty_like(off)
        g_lg = torch.empty((B, Lq, M, L * P), dtype=ctx.io, device=dev)
        g_ref = torch.empty((B, Lq, L, R), dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        gl, ga = g_loc.contiguous().float(), g_attn.contiguous().float()      # named: they must outlive the call that reads their pointers
        rc = _lib().mdetr_msda_prologue_backward(
            _io_code(ctx.io), ctx.rcode, off.data_ptr(), refc.data_ptr(), shapes.data_ptr(), attn.data_ptr(),
            gl.data_ptr(), ga.data_ptr(), g_off.data_ptr(), g_lg.data_ptr(),
            g_ref.data_ptr() if g_ref is not None else None, *ctx.geom, dev.index if dev.type == "cuda" else -1, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_msda_prologue_backward")
        if g_ref is not None:
            g_ref = g_ref.to(ctx.ref_dtype)
            if ctx.ref_shape != tuple(g_ref.shape):           # the caller passed a broadcast view: reduce back
                g_ref = g_ref.sum_to_size(ctx.ref_shape)
        return g_off, g_lg.view(B, Lq, M, L * P), g_ref, None


def packed_supported(packed, reference_points, L, P):
    """The one-GEMM form: `packed` [B, Lq, M L P 3] = per query the M L P 2 offsets, then the M L P logits (L = P = 4)."""
    return (L == 4 and P == 4 and (packed.is_cuda or _backend is not None) and packed.dtype in (torch.float32, torch.bfloat16)
            and packed.is_contiguous() and packed.data_ptr() % 16 == 0
            and reference_points.shape[-1] in (2, 6) and reference_points.stride(-1) == 1
            and getattr(_lib(), "mdetr_msda_prologue_forward_packed", None) is not None)


class _ProloguePacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed, ref, spatial_shapes, M, L, P):
        B, Lq = packed.shape[:2]
        R = ref.shape[-1]
        io = packed.dtype
        refc = ref if ref.dtype in (torch.float32, torch.bfloat16) else ref.float()
        if refc.stride(-1) != 1:
            refc = refc.contiguous()
        rcode = _io_code(refc.dtype)
        dev = packed.device
        loc = torch.empty((B, Lq, M, L, P, 2), dtype=torch.float32, device=dev)
        attn = torch.empty((B, Lq, M, L, P), dtype=torch.float32, device=dev)
        geom = (B, Lq, M, L, P, R, refc.stride(0), refc.stride(1), refc.stride(2))
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        rc = _lib().mdetr_msda_prologue_forward_packed(_io_code(io), rcode, packed.data_ptr(), refc.data_ptr(), spatial_shapes.data_ptr(),
                                                       loc.data_ptr(), attn.data_ptr(), *geom, dev.index if dev.type == "cuda" else -1, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_msda_prologue_forward_packed")
        ctx.save_for_backward(packed, refc, spatial_shapes, attn)
        ctx.geom, ctx.io, ctx.ref_dtype, ctx.rcode = geom, io, ref.dtype, rcode
        ctx.ref_shape = tuple(ref.shape)
        return loc, attn

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loc, g_attn):
        packed, refc, shapes, attn = ctx.saved_tensors
        B, Lq, M, L, P, R = ctx.geom[:6]
        dev = packed.device
        g_packed = torch.empty_like(packed)
        g_ref = torch.empty((B, Lq, L, R), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        gl, ga = g_loc.contiguous().float(), g_attn.contiguous().float()
        rc = _lib().mdetr_msda_prologue_backward_packed(
            _io_code(ctx.io), ctx.rcode, packed.data_ptr(), refc.data_ptr(), shapes.data_ptr(), attn.data_ptr(),
            gl.data_ptr(), ga.data_ptr(), g_packed.data_ptr(), g_ref.data_ptr() if g_ref is not None else None, *ctx.geom,
            dev.index if dev.type == "cuda" else -1, stream)
        if rc != 0:
            _capi.check(rc, "mdetr_msda_prologue_backward_packed")
        if g_ref is not None:
            g_ref = g_ref.to(ctx.ref_dtype)
            if ctx.ref_shape != tuple(g_ref.shape):
                g_ref = g_ref.sum_to_size(ctx.ref_shape)
        return g_packed, g_ref, None, None, None, None


def msda_prologue_packed(packed, reference_points, spatial_shapes, M, L, P):
    """packed [B, Lq, M L P 3] (offsets | logits per query: the output of ONE projection whose weight is the sampling-offset rows
    followed by the attention-weight rows) -> the same pair as `msda_prologue`."""
    return _ProloguePacked.apply(packed, reference_points, spatial_shapes, M, L, P)


def msda_prologue(offsets, logits, reference_points, spatial_shapes):
    """offsets [B,Lq,M,L,P,2], logits [B,Lq,M,L*P] (f32 or bf16), reference_points [B,Lq,L,2|6] (may be an
    expanded view), spatial_shapes int64 [L,2] -> (sampling_locations fp32 [B,Lq,M,L,P,2], attention_weights
    fp32 [B,Lq,M,L,P])."""
    return _Prologue.apply(offsets, logits, reference_points, spatial_shapes)
