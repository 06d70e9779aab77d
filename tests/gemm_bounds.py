"""Element-wise error bound of a bf16-operand / fp32-accumulate / one-rounding product against its fp64 value (test infrastructure).

    |y - ref| <= 2^-8 |ref| + c sqrt(K) 2^-23 (|a| |w|)        (y rounded to bf16: half an ulp is 2^-9 relative)

The first term is the output rounding (with a factor two of slack for values that the fp32 accumulation error moves across a
rounding boundary), the second the fp32 accumulation of K exact products in any order.  An fp32 result keeps only the second
term (plus its own 2^-24 rounding).  A wrong tap, a swapped fragment or a dropped k-step moves an element by ~|a||w| / sqrt(K) --
four orders of magnitude above the bound wherever |ref| is small."""
import torch


def product_bound(ref64, mag64, K, out_dtype=torch.bfloat16, c=4.0):
    acc = c * (K ** 0.5) * 2.0 ** -23 * mag64
    if out_dtype == torch.bfloat16:
        return ref64.abs() * 2.0 ** -8 + acc + 1e-30
    return ref64.abs() * 2.0 ** -22 + acc + 1e-30


def assert_product_close(y, ref64, mag64, K, what=""):
    bound = product_bound(ref64, mag64, K, y.dtype)
    err = (y.double() - ref64).abs()
    bad = err > bound
    assert not bool(bad.any()), "%s: %d of %d elements beyond the ulp bound (worst ratio %.2f at %s)" % (
        what, int(bad.sum()), bad.numel(), float((err / bound).max()), tuple(int(i) for i in torch.nonzero(bad)[0]))
