"""Element-wise error bound of a bf16-operand / fp32-accumulate / one-rounding product against its fp64 value (test infrastructure).

    y = RNE_bf16(s),  |s - ref| <= delta   =>   |y - ref| <= 2^-8 |ref| (1 + ...) + delta,      delta = c sqrt(K) 2^-23 (|a| |w|)

2^-8 is bf16's unit roundoff (8 significant bits); delta is the fp32 accumulation of K exact products in any order (a statistical
sqrt(K) with c = 6 for the tails of tens of millions of elements; the worst case would be K 2^-24).  An fp32 result keeps only
delta (plus its own 2^-24 rounding).  A wrong tap, a swapped fragment or a dropped k-step moves an element by ~|a||w| / sqrt(K) --
three to four orders of magnitude above the bound wherever |ref| is small."""
import torch


def product_bound(ref64, mag64, K, out_dtype=torch.bfloat16, c=6.0):
    acc = c * (K ** 0.5) * 2.0 ** -23 * mag64
    if out_dtype == torch.bfloat16:
        return (ref64.abs() + acc) * 2.0 ** -8 + acc + 1e-30
    return ref64.abs() * 2.0 ** -22 + acc + 1e-30


def assert_product_close(y, ref64, mag64, K, what=""):
    bound = product_bound(ref64, mag64, K, y.dtype)
    err = (y.double() - ref64).abs()
    bad = err > bound
    assert not bool(bad.any()), "%s: %d of %d elements beyond the ulp bound (worst ratio %.2f at %s)" % (
        what, int(bad.sum()), bad.numel(), float((err / bound).max()), tuple(int(i) for i in torch.nonzero(bad)[0]))


def conv2d_f64(x, w, bias=None, stride=1, padding=0):
    """F.conv2d in float64 as im2col + one matrix product (the box's convolution library has no fp64 kernels; unfold and matmul do):
    differentiable, so torch.autograd gives the fp64 input / weight gradients too."""
    B, C, H, W = x.shape
    N, _, kh, kw = w.shape
    OH, OW = (H + 2 * padding - kh) // stride + 1, (W + 2 * padding - kw) // stride + 1
    cols = torch.nn.functional.unfold(x, (kh, kw), padding=padding, stride=stride)          # [B, C kh kw, OH OW]
    y = (w.reshape(N, -1) @ cols).view(B, N, OH, OW)
    return y if bias is None else y + bias.view(1, -1, 1, 1)
