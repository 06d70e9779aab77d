"""Grouped fp32 products on the f32-input matrix instruction (csrc/sgemm.hip through ``mdetr_sgemm_grouped``): several independent
``C = mask(relu(sum_t A_t op(B_t) + bias + res))`` in ONE launch, exact fp32 arithmetic.  Built for the fp32 prediction heads (18 small
products per decoder level and direction, monodetr/heads.py); any 2-D fp32 / bf16 operands with unit column stride qualify."""
import ctypes

import torch

from . import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)
NT, NN, TN = 0, 1, 2
MAX_PROBLEMS, MAX_TERMS = 10, 5


class _Term(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64),
                ("k", ctypes.c_int32), ("a_dtype", ctypes.c_int32), ("b_dtype", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class _Problem(ctypes.Structure):
    _fields_ = [("term", _Term * MAX_TERMS), ("c", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("colsum", ctypes.c_void_p),
                ("mask", ctypes.c_void_p), ("res", ctypes.c_void_p), ("ldc", ctypes.c_int64), ("ldm", ctypes.c_int64),
                ("ldr", ctypes.c_int64), ("nterm", ctypes.c_int32), ("m", ctypes.c_int32), ("n", ctypes.c_int32),
                ("relu_cols", ctypes.c_int32), ("c_dtype", ctypes.c_int32), ("res_dtype", ctypes.c_int32)]


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _code(t):
    if t.dtype == torch.float32:
        return _capi.MDETR_F32
    if t.dtype == torch.bfloat16:
        return _capi.MDETR_BF16
    raise TypeError("sgemm: fp32 or bf16 operands, got %s" % t.dtype)


def _mat(t, what):
    if not (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
        raise ValueError("sgemm: %s must be 2-D with unit column stride, got shape %s strides %s" % (what, tuple(t.shape), t.stride()))
    return t


def usable(*tensors):
    """Can these tensors be operands / results?  (on the GPU -- or any device with the emulation backend --, fp32 or bf16, 2-D or 1-D
    with unit innermost stride)"""
    for t in tensors:
        if t is None:
            continue
        if not ((t.is_cuda or _backend is not None) and t.dtype in (torch.float32, torch.bfloat16) and t.dim() in (1, 2) and t.stride(-1) == 1):
            return False
    return True


class Problem:
    """One product of a group.  terms: [(A, B)] -- NT: A [M, K], B [N, K];  NN: A [M, K], B [K, N];  TN: A [K, M], B [K, N].
    out [M, N] (fp32 or bf16); bias [N] fp32; relu_cols: ReLU on the first columns (True = all); mask fp32 [M, N] (zero the
    result where mask <= 0); res [M, N] added before relu / mask; colsum [M] fp32 (TN: column sums of A)."""

    def __init__(self, terms, out, bias=None, relu_cols=0, mask=None, res=None, colsum=None):
        self.terms, self.out, self.bias, self.mask, self.res, self.colsum = list(terms), out, bias, mask, res, colsum
        self.relu_cols = out.shape[1] if relu_cols is True else int(relu_cols)


def grouped(mode, problems):
    """Launch the group; every result tensor is written in place."""
    if not 0 < len(problems) <= MAX_PROBLEMS:
        raise ValueError("sgemm: 1 .. %d problems per group" % MAX_PROBLEMS)
    arr = (_Problem * len(problems))()
    dev = problems[0].out.device
    for q, pr in zip(arr, problems):
        out = _mat(pr.out, "out")
        M, N = out.shape
        if len(pr.terms) > MAX_TERMS:
            raise ValueError("sgemm: at most %d terms per problem" % MAX_TERMS)
        for i, (A, B) in enumerate(pr.terms):
            A, B = _mat(A, "A"), _mat(B, "B")
            if mode == NT:
                K = A.shape[1]
                ok = A.shape == (M, K) and B.shape == (N, K)
            elif mode == NN:
                K = A.shape[1]
                ok = A.shape == (M, K) and B.shape == (K, N)
            else:
                K = A.shape[0]
                ok = A.shape == (K, M) and B.shape == (K, N)
            if not ok:
                raise ValueError("sgemm: term %d shapes A %s B %s do not give [%d, %d] in mode %d" % (i, tuple(A.shape), tuple(B.shape), M, N, mode))
            t = q.term[i]
            t.a, t.b, t.lda, t.ldb, t.k, t.a_dtype, t.b_dtype = A.data_ptr(), B.data_ptr(), A.stride(0), B.stride(0), K, _code(A), _code(B)
        q.nterm, q.m, q.n, q.relu_cols = len(pr.terms), M, N, pr.relu_cols
        q.c, q.ldc, q.c_dtype = out.data_ptr(), out.stride(0), _code(out)
        if pr.bias is not None:
            assert pr.bias.dtype == torch.float32 and pr.bias.shape == (N,) and pr.bias.is_contiguous()
            q.bias = pr.bias.data_ptr()
        if pr.mask is not None:
            mk = _mat(pr.mask, "mask")
            assert mk.dtype == torch.float32 and mk.shape == (M, N)
            q.mask, q.ldm = mk.data_ptr(), mk.stride(0)
        if pr.res is not None:
            rs = _mat(pr.res, "res")
            assert rs.shape == (M, N)
            q.res, q.ldr, q.res_dtype = rs.data_ptr(), rs.stride(0), _code(rs)
        if pr.colsum is not None:
            assert pr.colsum.dtype == torch.float32 and pr.colsum.shape == (M,) and pr.colsum.is_contiguous()
            q.colsum = pr.colsum.data_ptr()
    parr = ctypes.cast(arr, ctypes.c_void_p)
    need = _lib().mdetr_sgemm_workspace_bytes(mode, parr, len(problems)) if mode == TN else 0
    ws = torch.empty(need, dtype=torch.uint8, device=dev) if need > 0 else None          # (scratch of this launch; the caching allocator recycles it)
    rc = -1 if need < 0 else _lib().mdetr_sgemm_grouped(mode, parr, len(problems), ws.data_ptr() if ws is not None else None, max(need, 0),
                                                        dev.index if dev.type == "cuda" else -1,
                                                        torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None)
    if rc != 0:
        msg = _lib().mdetr_last_error()
        raise RuntimeError("mdetr_sgemm_grouped failed (code %d): %s" % (rc, msg.decode() if msg else "?"))
