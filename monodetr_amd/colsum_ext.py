"""``column_sum(x)``: fp32 column sums of a tall 2-D tensor on the GPU (csrc/colsum.hip) -- the bias
gradient of the token-wise linear layers.  CUDA tensors only; callers keep their own torch expression
for everything else."""
import torch

from . import _capi



def supported(x):
    return (x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.stride(1) == 1
            and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0
            and (x.stride(0) * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0 and x.shape[0] > 0)


def column_sum(x, out_dtype=torch.float32):
    """x [T, N] (f32 / bf16, unit column stride) -> [N] column sums, accumulated in fp32 and written as ``out_dtype``
    (float32 or bfloat16: one rounding, no separate cast launch)."""
    if not supported(x):
        raise RuntimeError("column_sum: needs a CUDA f32/bf16 matrix with 16-byte aligned rows")
    T, N = x.shape
    lib = _capi.lib()
    need = lib.mdetr_column_sum_workspace_bytes(T, N)
    from . import _workspace as W
    ws = W.get("colsum", x.device, need, floor=1 << 20)             # per stream: calls on different streams run concurrently
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("column_sum: out_dtype must be float32 or bfloat16")
    out = torch.empty(N, dtype=out_dtype, device=x.device)
    code = lambda dt: _capi.MDETR_BF16 if dt == torch.bfloat16 else _capi.MDETR_F32
    rc = lib.mdetr_column_sum_to(code(x.dtype), x.data_ptr(), out.data_ptr(), code(out_dtype), ws.data_ptr(), ws.numel(),
                                 T, N, x.stride(0), x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    _capi.check(rc, "mdetr_column_sum_to")
    return out
