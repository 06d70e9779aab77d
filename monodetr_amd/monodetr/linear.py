"""``token_linear`` and its relatives: every token-wise product y = x W^T + b of the model (linear layers, 1x1 convolutions of
channels-last maps) with its autograd structure, and the router that decides which kernel runs it.

bf16 operands on the GPU with MDETR_TGEMM (the committed bf16 list): forward and input gradient through csrc/tgemm.hip -- bias,
ReLU, Dropout, "+ identity" and the residual-path gradient in the product's epilogue -- and the weight + bias gradient through
csrc/twgrad.hip (>= 1 024 rows) or csrc/small_wgrad.hip, chunk partials summed by csrc/colsum.hip.  Everything else (fp32, autocast,
CPU tensors, operands the kernels' alignment rules refuse) takes the library GEMMs with the SAME autograd functions: a batched
split-K product for tall weight gradients (the library's single NT GEMM runs 16 workgroups deep at [81 600, 256]^T x [81 600, 256]:
205 us against 35), `colsum` for the bias.  Rules, not probing: nothing is timed at run time.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import colsum_ext, small_wgrad_ext

_MIN_TOKENS = 4096        # (2 048, which would hand layer4 to csrc/tgemm.hip too, measured no different: profiles/r06k_)
# MDETR_GEMM_RELU=1: "linear -> ReLU" as one library GEMM with the RELU_BIAS epilogue (torch._addmm_activation ->
# hipBLASLt) instead of a GEMM and an elementwise pass.  On the committed list (kernel_families.py); the bf16 step takes csrc/tgemm.hip's epilogue instead.
_GEMM_RELU = os.environ.get("MDETR_GEMM_RELU") == "1"
# MDETR_TGEMM=1: forward and input-gradient products of bf16 layers through csrc/tgemm.hip, their elementwise tails (bias, ReLU,
# Dropout, "+ identity", the residual-path gradient's accumulation) inside its epilogue.  kernel_families decides (committed for bf16).
_TGEMM = os.environ.get("MDETR_TGEMM") == "1"


# rows from which a bf16 weight gradient takes csrc/twgrad.hip instead of csrc/small_wgrad.hip: the decoder's 4 400 and layer4's 3 840
# rows included (421.7 -> 426.9 img/s, profiles/r05n_step_ab_twgrad_small_rows.log)
_TWGRAD_MIN_ROWS = 1024


# MDETR_RELU_PREMASK=1: the ReLU backward between two kernels of this repository is applied where the CONSUMER's input gradient leaves
# the chip (csrc/tgemm.hip's masked tail) and the producer skips its own pass -- see `ReluToken`.  kernel_families decides.
_PREMASK = os.environ.get("MDETR_RELU_PREMASK") == "1"


class ReluToken:
    """The contract between the producer of a ReLU output y and its ONE consumer.  The producer hangs a token on y
    (`y._mdetr_relu_token`) and keeps it; a consumer whose backward applies the ReLU's mask to the input gradient it returns (y <= 0 ->
    0: `mdetr_tgemm_masked`) sets `premasked`; the producer's backward then takes the arriving gradient as it is instead of running
    threshold_backward over it.  Masking is idempotent and linear, so a consumer may always mask its own contribution; the producer
    may only SKIP its pass when every consumer masks -- which is why tokens are only handed out where the graph is closed by
    construction: inside a Bottleneck (conv2 -> conv3) and between consecutive blocks of a stage (monodetr/backbone.py), never on a
    tensor that leaves the module.

    Limitation (what "closed by construction" does not see): an observer attached to the INTERMEDIATE tensor itself --
    `y.retain_grad()`, `y.register_hook(...)`, `torch.autograd.grad(loss, y)` -- receives the consumer-masked gradient (zero where
    y <= 0) instead of d loss / d y.  Parameter gradients and every gradient outside the module are unaffected (the mask would have
    been applied one node later anyway).  Module-level forward hooks switch the tokens off (backbone.py checks `_forward_hooks`);
    to observe intermediate gradients inside a bottleneck run with MDETR_RELU_PREMASK unset."""
    __slots__ = ("premasked",)

    def __init__(self):
        self.premasked = False


def relu_token_of(t):
    """The token a producer hung on `t`, if the switch is on."""
    return getattr(t, "_mdetr_relu_token", None) if _PREMASK else None


def _tgemm_ok(x2, weight, bias=None, res2=None, nn=False):
    if not _TGEMM:
        return False
    from .. import tgemm_ext
    return tgemm_ext.supported(x2, weight, nn=nn, res=res2, bias=bias)


def _drop_seed(x, p):
    """(seed, device-resident seed or None) of one dropout site: a replayed graph needs a seed that lives on the device."""
    if p <= 0.0:
        return 0, None
    if x.is_cuda and torch.cuda.is_current_stream_capturing():
        from ..attn_ext import site_seed
        return site_seed(x.device)
    from ..add_ln_ext import _host_seed
    return _host_seed(), None


def _act_backward(dy, y, scale=1.0):
    """Backward of y = dropout(relu(.)) from the saved output: dy * scale where y > 0 (kept and positive), else 0."""
    if scale == 1.0:
        return torch.ops.aten.threshold_backward(dy, y, 0.0)
    from .. import bias_act_ext
    if bias_act_ext.supported(y):
        return bias_act_ext.act_backward(dy, y, scale)
    return torch.where(y > 0, dy * scale, torch.zeros_like(dy))


def _split_count(T):
    """Chunks along the token axis: a divisor of T giving chunks of ~512-2048 rows."""
    best = 0
    for c in (64, 48, 80, 96, 60, 50, 40, 32, 128, 30, 24, 20, 16):
        if T % c == 0 and T // c >= 256:
            best = c
            break
    return best


def _weight_bias_grads(x2, dy2, weight, need_w, need_b, bias_dtype=None, out_dtype=None):
    """(dW, db) of y = x W^T + b from the [T, K] input and the [T, N] output gradient (either may be None when not needed), in the
    weight's dtype (or `out_dtype`).  bias_dtype: the bias parameter's dtype where it differs from the weight's -- an fp32 bias
    beside a bf16 weight gets its gradient from the fp32 sums, not through a bf16 rounding."""
    dt = out_dtype if out_dtype is not None else weight.dtype
    if need_b and bias_dtype is not None and bias_dtype != dt:
        dw, db = _weight_bias_grads(x2, dy2, weight, need_w, True, out_dtype=torch.float32)
        from .. import chunk_sums
        chunk_sums.flush()                           # (the conversions below READ what may be a registered, not yet computed chunk sum)
        return (dw.to(dt) if dw is not None else None), db.to(bias_dtype)
    dw = db = None
    T = x2.shape[0]
    if need_w and _TWGRAD_MIN_ROWS <= T <= small_wgrad_ext.MAX_ROWS and dt in (torch.float32, torch.bfloat16):
        # a few thousand rows of bf16 operands (the decoder's 4 400, layer4's 3 840): csrc/twgrad.hip as well (_TWGRAD_MIN_ROWS)
        from .. import conv_wgrad_ext
        if conv_wgrad_ext.token_supported(x2, dy2):
            return conv_wgrad_ext.token_weight_gradient(x2, dy2, dt, bias=need_b)
    if small_wgrad_ext.ENABLED and need_w and (T <= small_wgrad_ext.MAX_ROWS or dy2.shape[1] <= 64) \
            and dt in (torch.float32, torch.bfloat16) and small_wgrad_ext.supported(dy2, x2):
        # a few thousand rows (the decoder's 4 400): dW and db from one launch + one chunk sum (csrc/small_wgrad.hip)
        dw, db = small_wgrad_ext.small_wgrad(dy2, x2, dt)
        return dw, (db if need_b else None)
    # the batched split rounds every chunk's partial product to the activation dtype: worth it from ~8 000 rows on (43 vs
    # 110 us at 15 360), not for the decoder's 4 400 (29 + 12 vs 34 us, and 16 bf16 roundings instead of one)
    C = _split_count(T) if T > small_wgrad_ext.MAX_ROWS else 0
    if need_w:
        from .. import conv_wgrad_ext
        if T > small_wgrad_ext.MAX_ROWS and conv_wgrad_ext.token_supported(x2, dy2) and dt in (torch.float32, torch.bfloat16):
            # dW and db from ONE kernel + one chunk sum (csrc/twgrad.hip, the bias gradient riding along)
            return conv_wgrad_ext.token_weight_gradient(x2, dy2, dt, bias=need_b)
        elif C:
            parts = torch.bmm(dy2.view(C, T // C, -1).transpose(1, 2), x2.view(C, T // C, -1))    # [C, N, K]
            flat = parts.view(C, -1)
            if colsum_ext.supported(flat) and dt in (torch.float32, torch.bfloat16):
                # sum over the C chunks in one streaming pass (csrc/colsum.hip), written in the parameter's dtype
                dw = colsum_ext.column_sum(flat, dt).view(parts.shape[1], parts.shape[2])
            else:
                dw = parts.sum(0).to(dt)
        else:
            dw = (dy2.t() @ x2).to(dt)
    if need_b:
        if dy2.is_cuda and colsum_ext.supported(dy2) and dt in (torch.float32, torch.bfloat16):
            db = colsum_ext.column_sum(dy2, dt)       # csrc/colsum.hip: one HBM pass, fp32 accumulation, one rounding
        else:
            db = (dy2.view(C, T // C, -1).sum(1).sum(0) if C else dy2.sum(0)).to(dt)
    return dw, db


def _fwd_product(x2, weight, bias, relu=False, res2=None, dropout_p=0.0, seed=0, seed_dev=None, out_dtype=torch.bfloat16):
    """x2 W^T + bias (+ res2, ReLU, Dropout) through csrc/tgemm.hip (the caller has asked `_tgemm_ok`)."""
    from .. import tgemm_ext
    return tgemm_ext.tgemm(x2, weight, bias, res2, relu=relu, out_dtype=out_dtype, dropout_p=dropout_p, seed=seed, seed_dev=seed_dev)


def _input_gradient(dy2, weight, dskip2=None, mask2=None):
    """dY W (+ the gradient arriving over a residual path): csrc/tgemm.hip's NN form reads the parameter as it lies in memory and
    adds `dskip2` where the product's tile leaves the chip -- into a NEW tensor: nothing is written into the arriving gradient,
    whoever else may hold it.  The library route is a copy of the residual gradient followed by a beta = 1 GEMM.
    mask2 (the layer's input, a ReLU output): the result is zeroed where mask2 <= 0, inside the product where the kernel takes it."""
    if dy2.is_contiguous() and _tgemm_ok(dy2, weight, res2=dskip2, nn=True):
        from .. import tgemm_ext
        if mask2 is not None and tgemm_ext.masked_supported(dy2, weight, mask2, dskip2):
            return tgemm_ext.tgemm_masked(dy2, weight, mask2, dskip2)
        dx = tgemm_ext.tgemm(dy2, weight, None, dskip2, nn=True)
    elif dskip2 is None:
        dx = dy2 @ weight
    elif dskip2.dtype == dy2.dtype:
        dx = torch.addmm(dskip2, dy2, weight)
    else:
        dx = dskip2 + dy2 @ weight
    return dx if mask2 is None else torch.ops.aten.threshold_backward(dx, mask2, 0.0)


class _TokenLinearSkip(torch.autograd.Function):
    """(y, x') = (tail((x + pos) W^T + b), x): the linear layer of a residual branch together with the tensor the residual
    connection continues from.  x has ONE consumer in the graph, so the gradient arriving through x' (the residual path) and the
    layer's own input gradient dY W need no separate sum: backward adds the arriving gradient inside the input-gradient product
    (`_input_gradient`).  The encoder's 81 600-row layers read such a sum as three 42 MB tensors per residual site (reference
    depthaware_transformer.py:318-345: `src` feeds with_pos_embed, value_proj and the residual; the FFN input feeds linear1 and
    the residual); the bottleneck's conv1 shares its input with the identity connection (torchvision Bottleneck.forward behind
    backbone.py:93-106).  tail: ReLU (relu) followed by Dropout (dropout_p, csrc/tgemm.hip only).  `pos` is a constant (no gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pos, relu=False, dropout_p=0.0, premask=False, wide_out=False, out_token=None):
        ctx.premask = bool(premask) and pos is None                 # x is a ReLU output whose producer leaves the mask to this backward
        ctx.out_token = out_token                                   # (ReluToken) the consumer of y may take over THIS ReLU's backward mask
        ctx.wide_out = bool(wide_out)                               # fp32 result of bf16 operands (`_TokenLinear.forward`)
        q = x if pos is None else x + pos
        q2 = q.reshape(-1, q.shape[-1])
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.relu = bool(relu)
        ctx.scale = 1.0 / (1.0 - dropout_p) if dropout_p > 0.0 else 1.0
        if _tgemm_ok(q2, weight, bias):
            seed, seed_dev = _drop_seed(q, dropout_p)
            y = _fwd_product(q2, weight, bias, relu, None, dropout_p, seed, seed_dev,
                             torch.float32 if wide_out else torch.bfloat16).view(q.shape[:-1] + (weight.shape[0],))
        elif dropout_p > 0.0 or wide_out:
            raise RuntimeError("token_linear_skip: dropout and the fp32 result exist in csrc/tgemm.hip's epilogue only")
        elif relu:                                                   # library GEMM with the RELU_BIAS epilogue (as _TokenLinear)
            y = torch._addmm_activation(bias, q2, weight.t()).view(q.shape[:-1] + (weight.shape[0],))
        else:
            y = F.linear(q, weight, bias)
        if relu:
            ctx.save_for_backward(q, weight, y)
        else:
            ctx.save_for_backward(q, weight)
        return y, x.view_as(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, dskip):
        q, weight = ctx.saved_tensors[:2]
        if ctx.relu and not (ctx.out_token is not None and ctx.out_token.premasked and ctx.scale == 1.0):
            dy = _act_backward(dy.contiguous(), ctx.saved_tensors[2], ctx.scale)       # (premasked: the gradient arrived masked)
        if ctx.wide_out:
            dy = dy.to(q.dtype)                                        # one rounding of the fp32 gradient, as the narrow form receives it
        q2, dy2 = q.reshape(-1, q.shape[-1]), dy.reshape(-1, dy.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            ds2 = dskip.reshape(-1, q.shape[-1]) if dskip is not None else None
            dx = _input_gradient(dy2, weight, ds2, q2 if ctx.premask else None).view_as(q)
        dw, db = _weight_bias_grads(q2, dy2, weight, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], ctx.bias_dtype)
        return dx, dw, db, None, None, None, None, None, None


def token_linear_skip(x, weight, bias=None, pos=None, relu=False, dropout_p=0.0, relu_token=None, wide_out=False, out_token=None):
    """-> (token_linear(x + pos, weight, bias[, relu, dropout]), x'): use x' (== x) for everything that follows on the residual path;
    see `_TokenLinearSkip`.  Plain tensors out of it when the fused form does not apply.  relu / dropout_p: the caller asks
    `skip_relu_fusable` / `skip_dropout_fusable` first."""
    if (x.is_cuda or _tgemm_backend()) and x.dtype == weight.dtype and x.numel() // x.shape[-1] >= _MIN_TOKENS and torch.is_grad_enabled() and x.requires_grad \
            and not torch.is_autocast_enabled() and (pos is None or not pos.requires_grad) \
            and (not relu or skip_relu_fusable(bias, x, weight)) and (dropout_p <= 0.0 or skip_dropout_fusable(x, weight, bias)) \
            and (not wide_out or (not relu and x.is_cuda and x.dtype == torch.bfloat16 and _tgemm_ok(x.reshape(-1, x.shape[-1]), weight, bias))):
        # relu_token: x is a ReLU output with this call as its only consumer (`ReluToken`): the mask goes into the input gradient here
        premask = relu_token is not None and pos is None and x.dtype == torch.bfloat16 and _tgemm_ok(x.reshape(-1, x.shape[-1]), weight, bias)
        if premask:
            relu_token.premasked = True
        return _TokenLinearSkip.apply(x, weight, bias, pos, relu, dropout_p, premask, wide_out, out_token if relu else None)
    if dropout_p > 0.0:
        raise RuntimeError("token_linear_skip: ask skip_dropout_fusable before passing dropout_p")
    return token_linear(x if pos is None else x + pos, weight, bias, relu=relu, wide_out=wide_out), x


def skip_relu_fusable(bias, x=None, weight=None):
    if x is not None and weight is not None and _tgemm_ok(x.reshape(-1, x.shape[-1]), weight, bias):
        return True
    return _GEMM_RELU and bias is not None and bias.dim() == 1 and bias.is_contiguous()


def skip_dropout_fusable(x, weight, bias):
    return (x.is_cuda or _tgemm_backend()) and x.dtype == weight.dtype and x.numel() // x.shape[-1] >= _MIN_TOKENS \
        and not torch.is_autocast_enabled() and _tgemm_ok(x.reshape(-1, x.shape[-1]), weight, bias)


class _TokenLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, fused_relu=False, wide_out=False, dropout_p=0.0):
        """fused_relu: ReLU in the epilogue of the GEMM kernel or, failing that, of the library GEMM.
        dropout_p (csrc/tgemm.hip only, behind the ReLU): Dropout in the same epilogue.
        wide_out: bf16 operands, fp32 RESULT (the fp32 accumulator is written out unrounded) -- the decoder's deformable
        cross-attention differences neighbouring value rows for d/d(location), and 8 mantissa bits on the values made those
        gradients the least accurate of the bf16 model (cosine 0.86-0.98 against the fp32 model, round 2)."""
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.fused_relu = False
        ctx.wide_out = bool(wide_out)
        ctx.scale = 1.0 / (1.0 - dropout_p) if dropout_p > 0.0 else 1.0
        x2 = x.reshape(-1, x.shape[-1])
        if _tgemm_ok(x2, weight, bias):
            seed, seed_dev = _drop_seed(x, dropout_p)
            y = _fwd_product(x2, weight, bias, fused_relu, None, dropout_p, seed, seed_dev, torch.float32 if wide_out else torch.bfloat16)
            y = y.view(x.shape[:-1] + (weight.shape[0],))
            ctx.fused_relu = bool(fused_relu)
            if fused_relu:
                ctx.save_for_backward(x, weight, y)
            else:
                ctx.save_for_backward(x, weight)
            return y
        if dropout_p > 0.0:
            raise RuntimeError("token_linear: dropout is only fused into csrc/tgemm.hip's epilogue")
        if wide_out:
            y = torch.mm(x2, weight.t(), out_dtype=torch.float32)
            if bias is not None:
                y += bias.float()
            ctx.save_for_backward(x, weight)
            return y.view(x.shape[:-1] + (weight.shape[0],))
        if fused_relu:                                               # library GEMM, RELU_BIAS epilogue
            y = torch._addmm_activation(bias, x2, weight.t()).view(x.shape[:-1] + (weight.shape[0],))
            ctx.fused_relu = True
            ctx.save_for_backward(x, weight, y)
            return y
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        if ctx.fused_relu:
            x, weight, y = ctx.saved_tensors
            dy = _act_backward(dy.contiguous(), y, ctx.scale)        # ReLU (+ Dropout) of the epilogue: dy where y > 0, one launch
        else:
            x, weight = ctx.saved_tensors
        if ctx.wide_out:
            dy = dy.to(x.dtype)                                        # one rounding of the fp32 gradient, as the narrow form receives it
        dx = dw = db = None
        x2 = x.reshape(-1, x.shape[-1])
        dy2 = dy.reshape(-1, dy.shape[-1])
        if ctx.needs_input_grad[0]:
            dx = _input_gradient(dy2, weight).view_as(x)
        dw, db = _weight_bias_grads(x2, dy2, weight, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], ctx.bias_dtype)
        return dx, dw, db, None, None, None


class _TokenLinearResidualRelu(torch.autograd.Function):
    """out = relu(x W^T + b + res) from ONE kernel (csrc/tgemm.hip, residual epilogue): the tail of a bottleneck block -- the 1x1
    expansion with the frozen BN folded in, "+ identity" and the ReLU (torchvision Bottleneck.forward behind
    lib/models/monodetr/backbone.py:93-106) -- without the elementwise pass that read the product back (csrc/bias_act.hip: 0.47 ms
    per iteration forward).  Backward: the ReLU mask from the saved output; the masked gradient IS the identity's gradient."""

    @staticmethod
    def forward(ctx, x2, weight, bias, res2, out_token=None, in_premask=False):
        out = _fwd_product(x2, weight, bias, True, res2)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.out_token = out_token                    # set by the consumer of `out` if IT applies this ReLU's mask (`ReluToken`)
        ctx.in_premask = bool(in_premask)            # x2 is a ReLU output whose producer leaves its mask to this backward
        ctx.save_for_backward(x2, weight, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x2, weight, out = ctx.saved_tensors
        if ctx.out_token is not None and ctx.out_token.premasked:
            g = dout.contiguous()                    # already zero where out <= 0
        else:
            g = torch.ops.aten.threshold_backward(dout.contiguous(), out, 0.0)
        dx = _input_gradient(g, weight, None, x2 if ctx.in_premask else None) if ctx.needs_input_grad[0] else None
        dw, db = _weight_bias_grads(x2, g, weight, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], ctx.bias_dtype)
        return dx, dw, db, (g if ctx.needs_input_grad[3] else None), None, None


class _SplitRows(torch.autograd.Function):
    """Row blocks of a packed parameter (nn.MultiheadAttention's in_proj_weight / in_proj_bias: q | k | v) as views, with ONE
    gradient assembly: the framework's slice backward builds a zero tensor of the whole parameter per block, copies the block's
    gradient in and adds the results (3 fills + 3 copies + 2 adds per parameter and step); here the blocks' gradients are
    concatenated once."""

    @staticmethod
    def forward(ctx, packed, *sizes):
        ctx.sizes = sizes
        ctx.meta = (packed.shape[1:], packed.dtype, packed.device)
        return tuple(p.view_as(p) for p in packed.split(list(sizes), 0))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        tail, dtype, device = ctx.meta
        from .. import chunk_sums
        chunk_sums.flush()                           # the blocks' gradients are read by the concatenation: registered chunk sums first
        parts = [g if g is not None else torch.zeros((n,) + tuple(tail), dtype=dtype, device=device) for g, n in zip(grads, ctx.sizes)]
        return (torch.cat(parts, 0),) + (None,) * len(ctx.sizes)


def split_rows(packed, *sizes):
    """`packed.split(sizes, 0)` whose backward is a single concatenation (see `_SplitRows`)."""
    if not (torch.is_grad_enabled() and packed.requires_grad):
        return packed.split(list(sizes), 0)
    return _SplitRows.apply(packed, *sizes)


def _kernel_relu(x, weight, bias=None):
    """Can the ReLU ride in the GEMM's epilogue?  (csrc/tgemm.hip, or the library's RELU_BIAS epilogue)"""
    if _tgemm_ok(x.reshape(-1, x.shape[-1]), weight, bias):
        return True
    return _GEMM_RELU and bias is not None and bias.dim() == 1 and bias.is_contiguous()


def dropout_fusable(x, weight, bias):
    """Can `token_linear(..., relu=True, dropout_p=p)` run ReLU and Dropout inside the GEMM's epilogue?  (csrc/tgemm.hip)"""
    return (x.is_cuda or _tgemm_backend()) and x.dtype == weight.dtype and x.numel() // x.shape[-1] >= _MIN_TOKENS and torch.is_grad_enabled() \
        and not torch.is_autocast_enabled() and _tgemm_ok(x.reshape(-1, x.shape[-1]), weight, bias)


def _tgemm_backend():
    from .. import tgemm_ext
    return tgemm_ext._backend is not None


def token_linear(x, weight, bias=None, relu=False, wide_out=False, dropout_p=0.0):
    """F.linear (followed by ReLU if `relu`) with the split-K weight gradient for big token counts on the GPU;
    plain F.linear otherwise.  With a GEMM kernel of this repository enabled the ReLU runs in its epilogue.
    wide_out (bf16 operands on the GPU only): the result in fp32, see `_TokenLinear.forward`.
    dropout_p: Dropout behind the ReLU in the same epilogue (ask `dropout_fusable` first)."""
    if dropout_p > 0.0:
        if not (relu and dropout_fusable(x, weight, bias)):
            raise RuntimeError("token_linear: ask dropout_fusable before passing dropout_p (ReLU + Dropout ride in csrc/tgemm.hip's epilogue only)")
        return _TokenLinear.apply(x, weight, bias, True, False, dropout_p)
    if wide_out and not relu and x.is_cuda and x.dtype == weight.dtype == torch.bfloat16 and not torch.is_autocast_enabled():
        return _TokenLinear.apply(x, weight, bias, False, True)
    if (x.is_cuda or _tgemm_backend()) and x.dtype == weight.dtype and x.numel() // x.shape[-1] >= _MIN_TOKENS and torch.is_grad_enabled() \
            and not torch.is_autocast_enabled():
        if relu and _kernel_relu(x, weight, bias):
            return _TokenLinear.apply(x, weight, bias, True)
        y = _TokenLinear.apply(x, weight, bias, False)
    else:
        y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


def pointwise_eligible(x, kernel_size, stride, padding, groups):
    return ((x.is_cuda or _tgemm_backend()) and tuple(kernel_size) == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0)
            and groups == 1 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))


def pointwise_conv(x, weight, bias=None, relu=False):
    """1x1 stride-1 convolution of a channels_last activation == a linear layer over its B*H*W tokens:
    the [B,H,W,C] permutation is a view, the GEMMs go to hipBLASLt (which runs these memory-bound
    shapes near the HBM roofline) and the weight gradient takes `token_linear`'s split-K path; MIOpen's
    implicit-GEMM kernels plus their cast / zero-fill helpers took 2-3x as long on the same shapes.
    `relu`: followed by a ReLU (in the GEMM's epilogue where `token_linear` can put it there)."""
    B, C, H, W = x.shape
    y = token_linear(x.permute(0, 2, 3, 1).reshape(B * H * W, C), weight.reshape(weight.shape[0], C), bias, relu=relu)
    return y.view(B, H, W, -1).permute(0, 3, 1, 2)


def pointwise_conv_skip(x, weight, bias=None, relu=False, relu_token=None, hand_out_token=False):
    """-> (pointwise_conv(x, weight, bias, relu), x'): the 1x1 convolution that opens a residual block together with the tensor the
    identity connection continues from (x' == x) -- the gradient arriving through the identity path is folded into the
    convolution's input-gradient GEMM (beta = 1) instead of a separate 30-60 MB elementwise add per bottleneck
    (`_TokenLinearSkip`; reference torchvision Bottleneck.forward behind lib/models/monodetr/backbone.py:93-106)."""
    B, C, H, W = x.shape
    # hand_out_token: the (ReLU) result has exactly one consumer, which may apply this ReLU's backward mask itself (`ReluToken`)
    token = ReluToken() if (_PREMASK and hand_out_token and relu) else None
    y, xs = token_linear_skip(x.permute(0, 2, 3, 1).reshape(B * H * W, C), weight.reshape(weight.shape[0], C), bias, relu=relu, relu_token=relu_token,
                              out_token=token)
    y = y.view(B, H, W, -1).permute(0, 3, 1, 2)
    if token is not None:
        y._mdetr_relu_token = token                                 # (a producer on the library route ignores it and masks itself: masking twice is masking once)
    return y, xs.view(B, H, W, C).permute(0, 3, 1, 2)


def pointwise_residual_relu_eligible(x, weight, bias, identity):
    """Can `pointwise_conv_residual_relu` take relu(conv1x1(x) + identity)?  channels-last bf16 activations whose token views are
    views, a bf16 weight, csrc/tgemm.hip's shape rules."""
    if not (_TGEMM and x.dim() == 4 and identity.dim() == 4 and x.dtype == torch.bfloat16 and identity.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and not torch.is_autocast_enabled()
            and x.is_contiguous(memory_format=torch.channels_last) and identity.is_contiguous(memory_format=torch.channels_last)):
        return False
    B, C, H, W = x.shape
    N = weight.shape[0]
    if identity.shape != (B, N, H, W):
        return False
    return _tgemm_ok(x.permute(0, 2, 3, 1).reshape(B * H * W, C), weight.reshape(N, C), bias,
                     identity.permute(0, 2, 3, 1).reshape(B * H * W, N))


def pointwise_conv_residual_relu(x, weight, bias, identity, in_token=None, hand_out_token=False):
    """relu(conv1x1(x, weight, bias) + identity) for channels-last activations (ask `pointwise_residual_relu_eligible` first).
    in_token: x is a ReLU output whose producer agreed to leave its mask to this call's backward (`ReluToken`).
    hand_out_token: the result goes to exactly one consumer inside the caller's module: a token rides on it."""
    B, C, H, W = x.shape
    N = weight.shape[0]
    x2, r2 = x.permute(0, 2, 3, 1).reshape(B * H * W, C), identity.permute(0, 2, 3, 1).reshape(B * H * W, N)
    token = None
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or identity.requires_grad):
        token = ReluToken() if (_PREMASK and hand_out_token) else None
        in_premask = in_token is not None and x.requires_grad
        if in_premask:
            in_token.premasked = True
        out = _TokenLinearResidualRelu.apply(x2, weight.reshape(N, C), bias, r2, token, in_premask)
    else:
        out = _fwd_product(x2, weight.reshape(N, C), bias, True, r2)
    out = out.view(B, H, W, N).permute(0, 3, 1, 2)
    if token is not None:
        out._mdetr_relu_token = token
    return out


def pointwise_relu_fusable(x, weight, bias):
    """Would `pointwise_conv(..., relu=True)` run the ReLU inside the GEMM?  (callers that apply an in-place ReLU
    themselves otherwise)"""
    if not (_GEMM_RELU or _TGEMM):                    # the default path pays nothing for the question
        return False
    C = x.shape[1]
    return (x.numel() // C >= _MIN_TOKENS and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and _kernel_relu(x.permute(0, 2, 3, 1).reshape(-1, C), weight.reshape(weight.shape[0], C), bias))


def ffn_hidden(x, lin, dropout, activation=F.relu, tokenwise=True, skip=False):
    """``dropout(activation(lin(x)))`` -- the first half of an FFN (depthaware_transformer.py:334-337, :431-435;
    depth_predictor/transformer.py:57-65).  ``tokenwise``: the GEMM through `token_linear` (the encoder's 81 600 token
    rows) instead of the module call.  With csrc/tgemm.hip (MDETR_TGEMM) bias, ReLU and Dropout are the product's epilogue;
    otherwise, with MDETR_FUSED_EPILOGUE=1, ReLU and Dropout are one pass behind the GEMM (csrc/bias_act.hip), or the ReLU
    rides in the library GEMM's epilogue when a GEMM switch allows it."""
    from .. import bias_act_ext
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    if skip:
        # -> (hidden, x'): x' == x, to be used by the residual connection that follows (`token_linear_skip`)
        if activation is F.relu and tokenwise and torch.is_grad_enabled() and x.requires_grad and skip_dropout_fusable(x, lin.weight, lin.bias):
            return token_linear_skip(x, lin.weight, lin.bias, relu=True, dropout_p=p)
        h, x_out = token_linear_skip(x, lin.weight, lin.bias)
        if activation is F.relu and bias_act_ext.ENABLED and p > 0.0 and torch.is_grad_enabled() and bias_act_ext.supported(h):
            return bias_act_ext.bias_act(h, None, None, relu=True, dropout_p=p), x_out
        h = activation(h)
        return (dropout(h) if dropout is not None else h), x_out
    if activation is F.relu and tokenwise and p > 0.0 and dropout_fusable(x, lin.weight, lin.bias):
        return token_linear(x, lin.weight, lin.bias, relu=True, dropout_p=p)
    first = (lambda relu: token_linear(x, lin.weight, lin.bias, relu=relu)) if tokenwise else \
        (lambda relu: F.relu(lin(x)) if relu else lin(x))
    if activation is not F.relu:
        h = activation(first(False))
    elif bias_act_ext.ENABLED and p > 0.0 and torch.is_grad_enabled():
        h = first(False)
        if bias_act_ext.supported(h):
            return bias_act_ext.bias_act(h, None, None, relu=True, dropout_p=p)
        h = F.relu(h)
    else:
        h = first(True)
    return dropout(h) if dropout is not None else h


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose forward is `token_linear`: from 4 096 token rows on -- the
    decoder's B x 550 = 4 400 query rows, the depth encoder's B x 1 920, the encoder's 81 600 -- the kernels / split forms of this
    module's header; below that, or on the CPU, plain F.linear."""

    def forward(self, x):
        return token_linear(x, self.weight, self.bias)


class PointwiseConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose forward takes `pointwise_conv` when the
    input qualifies and nn.Conv2d's otherwise."""

    def forward(self, x):
        if pointwise_eligible(x, self.kernel_size, self.stride, self.padding, self.groups) \
                and x.dtype == self.weight.dtype and not torch.is_autocast_enabled():
            return pointwise_conv(x, self.weight, self.bias)
        from .. import conv_taps_ext
        if conv_taps_ext.ENABLED and x.dtype == self.weight.dtype and not torch.is_autocast_enabled() and self.padding_mode == "zeros" \
                and conv_taps_ext.supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            # 3x3 / stride 2 (the fourth pyramid level, monodetr.py:87-92; the depth predictor's downsample, depth_predictor.py:29-31)
            return conv_taps_ext.conv_strided(x, self.weight, self.bias, relu=False)
        return super().forward(x)
