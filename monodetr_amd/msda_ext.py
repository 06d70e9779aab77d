"""Drop-in for the reference's extension module ``MultiScaleDeformableAttention``.

The reference builds a pybind11/ATen module with two functions
(lib/models/monodetr/ops/src/vision.cpp:13-16, ops/src/ms_deform_attn.h:20-60):

    ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step)

Same names, argument meaning, return shapes and error behaviour here, implemented by the gfx950
kernels behind the C ABI (include/monodetr_amd.h).  ``monodetr_amd.install()`` registers this
module as ``sys.modules['MultiScaleDeformableAttention']`` so ``import MultiScaleDeformableAttention
as MSDA`` (ops/functions/ms_deform_attn_func.py:18) resolves to it unchanged.
"""
import os

import torch

from . import _capi

_NAMES5 = ("value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight")

# ``allow_cpu(True)``: host tensors go to the C ABI's host entry points mdetr_msda_forward_cpu /
# _backward_cpu instead of raising.  OFF by default: the reference raises for CPU tensors (ops/src/ms_deform_attn.h:38,
# cpu/ms_deform_attn_cpu.cpp:26,39) and so does this module -- the switch exists so that BASELINE configs[0] (the yaml on a
# CPU, one training iteration: plumbing) can run at all.  CUDA tensors never take this route, switch or no switch.
_ALLOW_CPU = False


def allow_cpu(on=True):
    global _ALLOW_CPU
    _ALLOW_CPU = bool(on)


def _check_inputs(tensors, names):
    # ms_deform_attn.h:26-38: CUDA tensors dispatch, anything else is an error
    if not tensors[0].is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    for t, n in zip(tensors, names):                      # ms_deform_attn_cuda.cu:28-38, :93-105
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % n)
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % n)
        if t.device != tensors[0].device:
            raise RuntimeError("%s is on %s but value is on %s" % (n, t.device, tensors[0].device))


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected value[B,S,M,D], sampling_loc[B,Lq,M,L,P,2], attn_weight[B,Lq,M,L,P]")
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (B, Lq, M, L, P, 2) or tuple(attn_weight.shape) != (B, Lq, M, L, P):
        raise RuntimeError("sampling_loc / attn_weight shapes do not match value / spatial_shapes")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 (ms_deform_attn_cuda.cu:67-68)")
    if tuple(spatial_shapes.shape) != (L, 2) or tuple(level_start_index.shape) != (L,):
        raise RuntimeError("spatial_shapes must be [L,2] and level_start_index [L]")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("value, sampling_loc and attn_weight must share one dtype")
    step = min(B, int(im2col_step))                       # ms_deform_attn_cuda.cu:50-52
    if B > 0 and (step <= 0 or B % step != 0):
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (B, step))
    return B, S, M, D, L, Lq, P


def _aligned(t):
    # the C ABI wants 16-byte aligned bases; contiguous views at odd storage offsets are copied
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


# Host copies of (spatial_shapes, level_start_index) for launch planning, kept ON the device tensor they were read from
# (the model reuses one tensor per resolution -- DepthAwareTransformer._level_tensors -- so the device->host copy happens
# once per resolution); a caller that builds fresh tensors every call pays one small synchronising copy per call.  (An
# earlier version keyed a dict on data_ptr / _version: the caching allocator hands the address of a freed shapes tensor
# to the next one, whose contents may differ.)  One scratch buffer per device for the workspace-based backward.


def _geometry_on_host(spatial_shapes, level_start_index):
    hit = getattr(spatial_shapes, "_mdetr_host", None)
    if hit is None or hit[0] != spatial_shapes._version or hit[1] is not level_start_index or hit[2] != level_start_index._version:
        host = (spatial_shapes.cpu().contiguous(), level_start_index.cpu().contiguous())
        hit = (spatial_shapes._version, level_start_index, level_start_index._version, host)
        try:
            spatial_shapes._mdetr_host = hit
        except AttributeError:                               # a tensor subclass without a __dict__: no caching
            pass
    return hit[3]


def _workspace(device, nbytes):
    # per (device, stream); zero on (re)allocation: msda_fused keeps a cookie in the header saying that its `far` region is all
    # zero between calls, and recycled pool memory could hold a stale one
    from . import _workspace as W
    return W.get("msda", device, nbytes, zero=True)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """-> Tensor [B, Lq, M*D]  (ms_deform_attn_cuda.cu:20-80)."""
    args = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if _ALLOW_CPU and not value.is_cuda:
        return _forward_cpu(*args, im2col_step)
    _check_inputs(args, _NAMES5)
    B, S, M, D, L, Lq, P = _dims(*args, im2col_step)
    code = _capi.dtype_code(value)
    value, sampling_loc, attn_weight = _aligned(value), _aligned(sampling_loc), _aligned(attn_weight)
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    dev = value.device.index
    rc = _capi.lib().mdetr_msda_forward(
        code, value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
        sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(),
        B, S, M, D, L, Lq, P, dev, _stream(value.device))
    _capi.check(rc, "mdetr_msda_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]  (ms_deform_attn_cuda.cu:83-153)."""
    args = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if _ALLOW_CPU and not value.is_cuda:
        return _backward_cpu(*args, grad_output, im2col_step)
    _check_inputs(args + (grad_output,), _NAMES5 + ("grad_output",))
    B, S, M, D, L, Lq, P = _dims(*args, im2col_step)
    if grad_output.dtype != value.dtype or grad_output.numel() != B * Lq * M * D:
        raise RuntimeError("grad_output must be [B,Lq,M*D] of value's dtype")
    code = _capi.dtype_code(value)
    value, sampling_loc, attn_weight, grad_output = (_aligned(t) for t in (value, sampling_loc, attn_weight, grad_output))
    grad_value = torch.empty_like(value)                  # written completely (or zero-filled first) by the C ABI on the stream
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    dev = value.device.index
    lib = _capi.lib()
    ws_bytes = 0
    if code == _capi.MDETR_F32 and D == 32 and B * Lq > 0:     # one-pass backward (csrc/msda_fused.hip) where the geometry qualifies
        sh_h, st_h = _geometry_on_host(spatial_shapes, level_start_index)
        ws_bytes = lib.mdetr_msda_backward_workspace_bytes(code, sh_h.data_ptr(), st_h.data_ptr(), B, S, M, D, L, Lq, P)
    if ws_bytes > 0:
        ws = _workspace(value.device, ws_bytes)
        rc = lib.mdetr_msda_backward_ex(
            code, value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
            grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
            B, S, M, D, L, Lq, P, sh_h.data_ptr(), st_h.data_ptr(), ws.data_ptr(), ws_bytes,
            dev, _stream(value.device))
        _capi.check(rc, "mdetr_msda_backward_ex")
    else:
        rc = lib.mdetr_msda_backward(
            code, value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
            grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
            B, S, M, D, L, Lq, P, dev, _stream(value.device))
        _capi.check(rc, "mdetr_msda_backward")
    return [grad_value, grad_loc, grad_attn]


def _host_args(tensors, names):
    for t, n in zip(tensors, names):
        if t.is_cuda:
            raise RuntimeError("%s must be a CPU tensor (value is)" % n)
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % n)


def _forward_cpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """Host tensors through mdetr_msda_forward_cpu (only after allow_cpu(True))."""
    args = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    _host_args(args, _NAMES5)
    B, S, M, D, L, Lq, P = _dims(*args, im2col_step)
    out = torch.empty((B, Lq, M * D), dtype=value.dtype)
    rc = _capi.lib().mdetr_msda_forward_cpu(_capi.dtype_code(value), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                            sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(), B, S, M, D, L, Lq, P)
    _capi.check(rc, "mdetr_msda_forward_cpu")
    return out


def _backward_cpu(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    args = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    _host_args(args + (grad_output,), _NAMES5 + ("grad_output",))
    B, S, M, D, L, Lq, P = _dims(*args, im2col_step)
    if grad_output.dtype != value.dtype or grad_output.numel() != B * Lq * M * D:
        raise RuntimeError("grad_output must be [B,Lq,M*D] of value's dtype")
    gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    rc = _capi.lib().mdetr_msda_backward_cpu(_capi.dtype_code(value), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                             sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                                             gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), B, S, M, D, L, Lq, P)
    _capi.check(rc, "mdetr_msda_backward_cpu")
    return [gv, gl, ga]


def _dims_mixed(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    if value.dtype != torch.bfloat16 or sampling_loc.dtype != torch.float32 or attn_weight.dtype != torch.float32:
        raise RuntimeError("bf16 variant: value must be bfloat16, sampling_loc and attn_weight float32")
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected value[B,S,M,D], sampling_loc[B,Lq,M,L,P,2], attn_weight[B,Lq,M,L,P]")
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (B, Lq, M, L, P, 2) or tuple(attn_weight.shape) != (B, Lq, M, L, P):
        raise RuntimeError("sampling_loc / attn_weight shapes do not match value / spatial_shapes")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 (ms_deform_attn_cuda.cu:67-68)")
    if tuple(spatial_shapes.shape) != (L, 2) or tuple(level_start_index.shape) != (L,):
        raise RuntimeError("spatial_shapes must be [L,2] and level_start_index [L]")
    return B, S, M, D, L, Lq, P


def bf16_supported(value, sampling_loc):
    """Shapes the mixed-precision kernels cover (D = 32, L = P = 4 -- the model's configuration)."""
    return (value.is_cuda and value.dtype == torch.bfloat16 and value.dim() == 4 and value.shape[3] == 32
            and sampling_loc.dim() == 6 and sampling_loc.shape[3] == 4 and sampling_loc.shape[4] == 4)


def ms_deform_attn_forward_bf16(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """bf16 ``value`` in, bf16 ``[B, Lq, M*D]`` out; fp32 sampling locations / weights; fp32 accumulation.
    Not part of the reference module (its kernels are fp32/fp64 only): include/monodetr_amd.h, mdetr_msda_forward_bf16."""
    args = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    _check_inputs(args, _NAMES5)
    B, S, M, D, L, Lq, P = _dims_mixed(*args)
    value, sampling_loc, attn_weight = _aligned(value), _aligned(sampling_loc), _aligned(attn_weight)
    out = torch.empty((B, Lq, M * D), dtype=torch.bfloat16, device=value.device)
    rc = _capi.lib().mdetr_msda_forward_bf16(
        value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
        sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(),
        B, S, M, D, L, Lq, P, value.device.index, _stream(value.device))
    _capi.check(rc, "mdetr_msda_forward_bf16")
    return out


def ms_deform_attn_backward_bf16(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight], all fp32, from bf16 ``value`` / ``grad_output``."""
    args = (value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    _check_inputs(args + (grad_output,), _NAMES5 + ("grad_output",))
    B, S, M, D, L, Lq, P = _dims_mixed(*args)
    if grad_output.dtype != torch.bfloat16 or grad_output.numel() != B * Lq * M * D:
        raise RuntimeError("grad_output must be bfloat16 [B,Lq,M*D]")
    value, sampling_loc, attn_weight, grad_output = (_aligned(t) for t in (value, sampling_loc, attn_weight, grad_output))
    grad_value = torch.empty(value.shape, dtype=torch.float32, device=value.device)   # zero-filled by the C ABI
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    lib = _capi.lib()
    ws_bytes, ws_ptr, sh_ptr, st_ptr = 0, 0, 0, 0
    if B * Lq > 0:
        sh_h, st_h = _geometry_on_host(spatial_shapes, level_start_index)
        ws_bytes = lib.mdetr_msda_backward_workspace_bytes(_capi.MDETR_F32, sh_h.data_ptr(), st_h.data_ptr(),
                                                           B, S, M, D, L, Lq, P)
        if ws_bytes > 0:
            ws_ptr, sh_ptr, st_ptr = _workspace(value.device, ws_bytes).data_ptr(), sh_h.data_ptr(), st_h.data_ptr()
    rc = lib.mdetr_msda_backward_bf16(
        value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
        sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
        grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
        B, S, M, D, L, Lq, P, sh_ptr, st_ptr, ws_ptr, ws_bytes, value.device.index, _stream(value.device))
    _capi.check(rc, "mdetr_msda_backward_bf16")
    return [grad_value, grad_loc, grad_attn]


def ms_deform_attn_indices(spatial_shapes, sampling_loc):
    """int32 [B,Lq,M,L,P,4] = (in_window, h_low, w_low, corner_mask): the gather indices the kernels
    use, exposed for bit-exact index parity tests (not part of the reference module)."""
    if not sampling_loc.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    if not sampling_loc.is_contiguous() or not spatial_shapes.is_contiguous():
        raise RuntimeError("sampling_loc tensor has to be contiguous")
    B, Lq, M, L, P, _ = sampling_loc.shape
    idx = torch.empty((B, Lq, M, L, P, 4), dtype=torch.int32, device=sampling_loc.device)
    rc = _capi.lib().mdetr_msda_indices(
        _capi.dtype_code(sampling_loc), spatial_shapes.data_ptr(), _aligned(sampling_loc).data_ptr(),
        idx.data_ptr(), B, M, L, Lq, P, sampling_loc.device.index, _stream(sampling_loc.device))
    _capi.check(rc, "mdetr_msda_indices")
    return idx
