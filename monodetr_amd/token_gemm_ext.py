"""``token_gemm(x, weight, bias, relu)``: y = x W^T + b (+ ReLU) in bf16 with the weight resident in LDS
(csrc/token_gemm.hip through ``mdetr_token_linear``).  CUDA bf16 only, K in {64, 128, 256, 512}; callers check
``supported`` and keep the library GEMM for everything else."""
import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(x2, weight, n_out=None):
    """x2 [T, K] bf16 with unit column stride, weight [N, K] contiguous bf16."""
    K = x2.shape[1]
    N = weight.shape[0] if n_out is None else n_out
    return ((x2.is_cuda or _backend is not None) and x2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x2.dim() == 2
            and K in (64, 128, 256, 512) and weight.shape[1] == K and N % 8 == 0 and x2.stride(1) == 1
            and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0 and weight.is_contiguous()
            and weight.data_ptr() % 16 == 0 and x2.shape[0] > 0)


def token_gemm(x2, weight, bias=None, relu=False):
    T, K = x2.shape
    N = weight.shape[0]
    y = torch.empty((T, N), dtype=torch.bfloat16, device=x2.device)
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
    rc = _lib().mdetr_token_linear(
        x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
        T, N, K, x2.stride(0), y.stride(0), 1 if relu else 0, x2.device.index if x2.is_cuda else -1,
        torch.cuda.current_stream(x2.device).cuda_stream if x2.is_cuda else None)
    _capi.check(rc, "mdetr_token_linear")
    return y
