"""``small_wgrad(dy, x, dtype)``: weight and bias gradient of a linear layer over a few thousand token rows in one launch plus
one chunk sum (csrc/small_wgrad.hip through ``mdetr_small_wgrad``) -- the decoder's 4 400-row layers, where the library's
single-tile GEMM and generic reductions take ~45 us and 4-5 launches per layer."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# MDETR_SMALL_WGRAD=1 routes token_linear's backward through the kernel where the shape qualifies
ENABLED = os.environ.get("MDETR_SMALL_WGRAD") == "1"
MAX_ROWS = 8192
MAX_ROWS_NARROW = 65536       # N <= 64


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(dy, x):
    """dy [T, N], x [T, K]: same dtype (f32 / bf16) and device, unit column stride, 16-byte aligned rows of x, K a multiple
    of 64, any N with N K <= 524 288, T <= 8 192 (65 536 for N <= 64)."""
    if not ((dy.is_cuda or _backend is not None) and dy.dim() == 2 and x.dim() == 2 and dy.shape[0] == x.shape[0]
            and dy.dtype == x.dtype and dy.dtype in (torch.float32, torch.bfloat16) and dy.device == x.device):
        return False
    T, N = dy.shape
    K = x.shape[1]
    return (0 < T <= (MAX_ROWS_NARROW if N <= 64 else MAX_ROWS) and K % 64 == 0 and 0 < N * K <= 2048 * 256
            and dy.stride(1) == 1 and x.stride(1) == 1 and x.stride(0) % 8 == 0
            and dy.stride(0) >= N and x.stride(0) >= K and x.data_ptr() % 16 == 0)


def small_wgrad(dy, x, out_dtype=None):
    """(dW [N, K], db [N]) = (dy^T x, column sums of dy), accumulated in fp32, written as ``out_dtype`` (default: dy's)."""
    if not supported(dy, x):
        raise RuntimeError("small_wgrad: needs two CUDA f32/bf16 matrices with <= 8192 rows (65536 for N <= 64), K a multiple of 64, N K <= 524288")
    out_dtype = out_dtype or dy.dtype
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("small_wgrad: out_dtype must be float32 or bfloat16")
    T, N = dy.shape
    K = x.shape[1]
    lib = _lib()
    need = lib.mdetr_small_wgrad_workspace_bytes(T, N, K)
    from . import _workspace as W
    ws = W.get("small_wgrad", dy.device, need, floor=8 << 20)       # per stream: calls on different streams run concurrently
    out = torch.empty((N * K + N + 3) // 4 * 4, dtype=out_dtype, device=dy.device)
    code = lambda dt: _capi.MDETR_BF16 if dt == torch.bfloat16 else _capi.MDETR_F32      # noqa: E731
    dev, stream = (dy.device.index, torch.cuda.current_stream(dy.device).cuda_stream) if dy.is_cuda else (-1, None)
    _capi.check(lib.mdetr_small_wgrad(code(dy.dtype), dy.data_ptr(), x.data_ptr(), out.data_ptr(), code(out_dtype), ws.data_ptr(), ws.numel(),
                                      T, N, K, dy.stride(0), x.stride(0), dev, stream), "mdetr_small_wgrad")
    return out[:N * K].view(N, K), out[N * K:N * K + N]
