"""``weight_gradient(x, dy, k, stride, dtype)``: dW of a 3x3 (stride 1 / 2, pad 1) or 1x1 (stride 2) convolution from the
channels_last bf16 activation and output gradient (csrc/conv_wgrad.hip through ``mdetr_conv_wgrad``: split-K over pixel tiles
on the matrix cores; the per-chunk partial gradients are added in a fixed order by csrc/colsum.hip, one rounding)."""
import os

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
# MDETR_CONV_WGRAD=1: the hand-written convolutions (conv3x3_ext, conv_taps_ext) take their weight gradient from this kernel
# instead of the library's (MIOpen igemm_wrw)
ENABLED = os.environ.get("MDETR_CONV_WGRAD") == "1"


def _lib():
    return _backend if _backend is not None else _capi.lib()


def supported(x, dy, k, stride):
    return ((ENABLED or _backend is not None) and (x.is_cuda or _backend is not None) and x.dim() == 4 and dy.dim() == 4 and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
            and ((k == 3 and stride in (1, 2)) or (k == 1 and stride in (1, 2))) and x.shape[1] % 64 == 0 and dy.shape[1] % 32 == 0
            and x.numel() > 0 and dy.numel() > 0 and x.is_contiguous(memory_format=torch.channels_last)
            and dy.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0)


# Weight AND bias gradients of token-wise linear layers over tens of thousands of rows (the encoder's 81 600, the backbone's 1x1
# convolutions) as the 1x1 case of this kernel, the bias gradient riding along (mdetr_token_wgrad), instead of the library's batched
# split-K product + chunk sum + two-launch column sum: 53 -> 43 us at [81 600, 256] x [81 600, 256], 49 -> 28-36 us on the backbone's
# and the depth head's shapes (profiles/r04q_wgradbench.json).
TOKEN_ROUTE = True


def token_weight_gradient(x2, dy2, dtype, bias=False):
    """(dW [N, K], db [N] or None) = (dy2^T x2, column sums of dy2) for token matrices x2 [T, K], dy2 [T, N] (bf16, contiguous rows, T a
    multiple of 8): the 1x1 case of the convolution weight-gradient kernel, the bias gradient riding along on the dy operand that
    is already in LDS -- one kernel + one sum over its chunks for both gradients (the library route: a batched split-K product, a
    chunk sum, and a two-launch column sum that reads dy a second time)."""
    T, K = x2.shape
    N = dy2.shape[1]
    lib = _lib()
    chunks = lib.mdetr_token_wgrad_chunks(T, K, N)
    if chunks <= 0:
        raise RuntimeError("token_wgrad: unsupported problem")
    cols = N * K + (N if bias else 0)
    cuda = x2.is_cuda
    from . import chunk_sums
    batched = chunk_sums.deferring() and cols % 4 == 0 and dtype in (torch.float32, torch.bfloat16) and (cuda or chunk_sums._backend is not None)
    if cuda and _backend is None and not batched:
        from . import _workspace as W_
        part = W_.get("conv_wgrad", x2.device, chunks * cols * 4).view(torch.float32)[:chunks * cols]
    else:                                            # (a deferred sum reads its partials later: they get a buffer of their own)
        part = torch.empty(chunks * cols, dtype=torch.float32, device=x2.device)
    rc = lib.mdetr_token_wgrad(x2.data_ptr(), dy2.data_ptr(), part.data_ptr(), part.numel(), T, K, N, 1 if bias else 0,
                               x2.device.index if cuda else -1, torch.cuda.current_stream(x2.device).cuda_stream if cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_token_wgrad")
    part = part.view(chunks, cols)
    if batched:
        both = chunk_sums.chunk_sum(part, dtype)     # one launch for all the iteration's weight gradients (chunk_sums.flush)
    elif cuda and _backend is None:
        from .colsum_ext import column_sum, supported as colsum_ok
        out_dt = dtype if dtype in (torch.float32, torch.bfloat16) else torch.float32
        both = (column_sum(part, out_dtype=out_dt) if colsum_ok(part) else part.sum(0)).to(dtype)
    else:
        both = part.sum(0).to(dtype)
    return both[:N * K].view(N, K), (both[N * K:] if bias else None)


def token_supported(x2, dy2):
    """bf16 token matrices with contiguous rows whose widths are multiples of 8 (csrc/twgrad.hip).  With MDETR_TUNE="twgrad=0" (tests: the
    1x1 case of csrc/conv_wgrad.hip behind the same entry point) that kernel's narrower rules apply."""
    T = x2.shape[0]
    ok = (TOKEN_ROUTE and (ENABLED or _backend is not None) and x2.dtype == torch.bfloat16 and dy2.dtype == torch.bfloat16 and x2.is_contiguous()
          and dy2.is_contiguous() and (x2.is_cuda or _backend is not None) and x2.data_ptr() % 16 == 0 and dy2.data_ptr() % 16 == 0
          and T * max(x2.shape[1], dy2.shape[1]) * 2 < (1 << 31))
    from . import _tune
    if _tune.get("twgrad") == "0":
        # (narrow outputs waste that kernel's 128-row dy block; a weight of more than 64 (128 x 64) blocks leaves too few chunks)
        return ok and T % 8 == 0 and x2.shape[1] % 64 == 0 and dy2.shape[1] % 32 == 0 and dy2.shape[1] >= 128 \
            and ((dy2.shape[1] + 127) // 128) * (x2.shape[1] // 64) <= 64
    return ok and x2.shape[1] % 8 == 0 and dy2.shape[1] % 8 == 0


def weight_gradient(x, dy, k, stride, dtype=torch.bfloat16):
    """x [B, C, H, W], dy [B, N, OH, OW] (channels_last bf16) -> dW [N, C, k, k] in ``dtype`` (channels_last strides)."""
    B, C, H, W = x.shape
    _, N, OH, OW = dy.shape
    lib = _lib()
    chunks = lib.mdetr_conv_wgrad_chunks(B, H, W, C, OH, OW, N, k, stride)
    if chunks <= 0:
        raise RuntimeError("conv_wgrad: unsupported problem")
    cols = N * k * k * C
    cuda = x.is_cuda
    from . import chunk_sums
    batched = chunk_sums.deferring() and dtype in (torch.float32, torch.bfloat16) and (cuda or chunk_sums._backend is not None)
    if cuda and _backend is None and not batched:
        from . import _workspace as W_
        part = W_.get("conv_wgrad", x.device, chunks * cols * 4).view(torch.float32)[:chunks * cols]
    else:
        part = torch.empty(chunks * cols, dtype=torch.float32, device=x.device)
    rc = lib.mdetr_conv_wgrad(x.data_ptr(), dy.data_ptr(), part.data_ptr(), part.numel(), B, H, W, C, OH, OW, N, k, stride,
                              x.device.index if cuda else -1, torch.cuda.current_stream(x.device).cuda_stream if cuda else None)
    if rc != 0:
        _capi.check(rc, "mdetr_conv_wgrad")
    part = part.view(chunks, cols)
    if batched:
        dw = chunk_sums.chunk_sum(part, dtype)
    elif cuda and _backend is None:
        from .colsum_ext import column_sum, supported as colsum_ok
        out_dt = dtype if dtype in (torch.float32, torch.bfloat16) else torch.float32
        dw = (column_sum(part, out_dtype=out_dt) if colsum_ok(part) else part.sum(0)).to(dtype)
    else:
        dw = part.sum(0).to(dtype)
    return dw.view(N, k, k, C).permute(0, 3, 1, 2)                       # [N, C, k, k] with channels_last strides
