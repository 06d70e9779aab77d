"""``tgemm(a, w, ...)``: y = dropout(relu(a op(w) + bias + res)) in bf16 on the matrix cores with the tail inside the product's
epilogue (csrc/tgemm.hip through ``mdetr_tgemm``): every token-wise product of the iteration -- forward ``a w^T`` with ``w = W[N, K]``,
input gradient ``a w`` with ``w = W[K, N]`` (``nn=True``: the parameter as it lies in memory), ``res is out`` for a beta = 1
accumulation.  CUDA bf16 only; callers ask ``supported`` first and keep the library GEMM for everything else."""
import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
RELU, NN, BIAS_F32, OUT_F32 = 1, 2, 4, 8


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _rows_ok(t, cols):
    return t.dim() == 2 and t.shape[1] == cols and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.stride(0) >= cols and t.data_ptr() % 16 == 0


def supported(a2, w, nn=False, res=None, bias=None, out=None):
    """a2 [T, K] bf16 with unit column stride; w bf16 [N, K] (nn: [K, N]) with unit column stride; res / out [T, N] bf16."""
    if not ((a2.is_cuda or _backend is not None) and a2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a2.dim() == 2 and w.dim() == 2):
        return False
    T, K = a2.shape
    N = w.shape[1] if nn else w.shape[0]
    if not (T > 0 and K % 8 == 0 and N % 8 == 0 and (w.shape[0] if nn else w.shape[1]) == K and _rows_ok(a2, K) and _rows_ok(w, w.shape[1])):
        return False
    if res is not None and not (res.dtype == torch.bfloat16 and tuple(res.shape) == (T, N) and _rows_ok(res, N)):
        return False
    if out is not None and not (out.dtype in (torch.bfloat16, torch.float32) and tuple(out.shape) == (T, N) and out.stride(1) == 1
                                and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0):
        return False
    if bias is not None and not (bias.dim() == 1 and bias.shape[0] == N and bias.is_contiguous() and bias.data_ptr() % 16 == 0
                                 and bias.dtype in (torch.bfloat16, torch.float32)):
        return False
    return True


def tgemm(a2, w, bias=None, res=None, relu=False, nn=False, out=None, out_dtype=torch.bfloat16, dropout_p=0.0, seed=0, seed_dev=None):
    """-> out [T, N] (allocated unless given; ``res is out`` accumulates into it)."""
    T, K = a2.shape
    N = w.shape[1] if nn else w.shape[0]
    if out is None:
        out = torch.empty((T, N), dtype=out_dtype, device=a2.device)
    flags = (RELU if relu else 0) | (NN if nn else 0) | (BIAS_F32 if bias is not None and bias.dtype == torch.float32 else 0) \
        | (OUT_F32 if out.dtype == torch.float32 else 0)
    rc = _lib().mdetr_tgemm(
        a2.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, res.data_ptr() if res is not None else None,
        out.data_ptr(), T, N, K, a2.stride(0), w.stride(0), res.stride(0) if res is not None else 0, out.stride(0), flags,
        float(dropout_p), int(seed), seed_dev.data_ptr() if seed_dev is not None else None,
        a2.device.index if a2.is_cuda else -1, torch.cuda.current_stream(a2.device).cuda_stream if a2.is_cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_tgemm")
    return out


def masked_supported(a2, w, mask, res=None):
    """`tgemm_masked`'s operands: the NN rules of `supported` plus a bf16 mask [T, N] with 16-byte aligned rows."""
    if not supported(a2, w, nn=True, res=res):
        return False
    return mask.dtype == torch.bfloat16 and tuple(mask.shape) == (a2.shape[0], w.shape[1]) and _rows_ok(mask, w.shape[1])


def tgemm_masked(a2, w, mask, res=None):
    """-> out [T, N] = where(mask <= 0, 0, a2 @ w + res): an input gradient with the ReLU backward of the layer's input inside
    (``mdetr_tgemm_masked``; w = the parameter [K, N] as it lies in memory)."""
    T, K = a2.shape
    N = w.shape[1]
    out = torch.empty((T, N), dtype=torch.bfloat16, device=a2.device)
    rc = _lib().mdetr_tgemm_masked(
        a2.data_ptr(), w.data_ptr(), res.data_ptr() if res is not None else None, mask.data_ptr(), out.data_ptr(), T, N, K,
        a2.stride(0), w.stride(0), res.stride(0) if res is not None else 0, mask.stride(0), out.stride(0),
        a2.device.index if a2.is_cuda else -1, torch.cuda.current_stream(a2.device).cuda_stream if a2.is_cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_tgemm_masked")
    return out
