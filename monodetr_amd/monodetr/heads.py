"""The five prediction heads that read one decoder level's output (reference lib/models/monodetr/monodetr.py:222-262 and the
box / size heads the decoder calls for its iterative refinement, depthaware_transformer.py:602-613), evaluated together:

    bbox_embed   MLP 256 -> 256 -> 256 -> 6        dim_embed_3d  MLP 256 -> 256 -> 3
    depth_embed  MLP 256 -> 256 -> 2               angle_embed   MLP 256 -> 256 -> 24          class_embed  Linear 256 -> 3

The heads stay in fp32 (parameters, arithmetic, results) behind a bf16 body.  As modules they are 18 products per level and
direction on a [B x queries, 256] tensor, each a library launch of 5 - 45 microseconds, with ~30 elementwise launches between
them (casts, ReLUs and their masks, the concatenation of the first layers' weights, bias sums): 1.2 ms of the round-5 iteration.
Here a level's forward is three grouped launches of csrc/sgemm.hip (exact fp32 on the f32-input matrix instruction; bias, ReLU
and the widening of a bf16 input inside) and its backward four (ReLU masks, the sum of the five input gradients, the bf16
rounding of dX and the gradient arriving from the next decoder layer inside; bias gradients ride on the weight-gradient group).
No parameter is copied or concatenated: a group addresses every weight where it lies.
"""
import os

import torch

from .. import sgemm_ext

# MDETR_HEADS=1 (kernel_families decides): the grouped fp32 kernels; off = the modules (library GEMMs + elementwise launches)
ENABLED = os.environ.get("MDETR_HEADS") == "1"


def _heads_ok(x, bbox, dim, dep, ang, cls):
    from .depthaware_transformer import MLP
    if not ENABLED:
        return False
    if not (isinstance(bbox, MLP) and bbox.num_layers == 3 and all(isinstance(m, MLP) and m.num_layers == 2 for m in (dim, dep, ang))
            and isinstance(cls, torch.nn.Linear)):
        return False
    ps = [p for m in (bbox, dim, dep, ang, cls) for p in m.parameters()]
    if not all(p.dtype == torch.float32 and p.is_contiguous() for p in ps):
        return False
    C = x.shape[-1]
    firsts = [bbox.layers[0], dim.layers[0], dep.layers[0], ang.layers[0]]
    hid = firsts[0].out_features
    if not (all(l.in_features == C and l.out_features == hid and l.bias is not None for l in firsts) and cls.in_features == C
            and cls.bias is not None and bbox.layers[1].in_features == hid and bbox.layers[1].out_features == hid
            and all(m.layers[-1].bias is not None for m in (bbox, dim, dep, ang)) and bbox.layers[1].bias is not None):
        return False
    return x.dim() == 3 and x.is_contiguous() and sgemm_ext.usable(x.view(-1, C), *ps) and not torch.is_autocast_enabled()


class _HeadsLevel(torch.autograd.Function):
    """(x, 18 parameters) -> (delta, size, depth, angle, logits, x'): x' == x, for whatever continues from x (the next decoder
    layer): its gradient arrives here and is summed inside the input-gradient launch (`res`), not by a separate pass."""

    @staticmethod
    def forward(ctx, x, w1b, b1b, w2b, b2b, w3b, b3b, w1d, b1d, wd, bd, w1p, b1p, wp, bp, w1a, b1a, wa, ba, wc, bc):
        ctx.set_materialize_grads(False)
        B, Q, C = x.shape
        T, hid = B * Q, w1b.shape[0]
        x2 = x.view(T, C)
        new = lambda n: torch.empty((T, n), dtype=torch.float32, device=x.device)         # noqa: E731
        h1, h2 = new(4 * hid), new(hid)
        delta, size, depth, angle, logits = new(w3b.shape[0]), new(wd.shape[0]), new(wp.shape[0]), new(wa.shape[0]), new(wc.shape[0])
        P, G = sgemm_ext.Problem, sgemm_ext.grouped
        sl = [slice(i * hid, (i + 1) * hid) for i in range(4)]                              # box | size | depth | angle
        G(sgemm_ext.NT, [P([(x2, w)], h1[:, s], bias=b, relu_cols=True) for w, b, s in zip((w1b, w1d, w1p, w1a), (b1b, b1d, b1p, b1a), sl)]
          + [P([(x2, wc)], logits, bias=bc)])
        G(sgemm_ext.NT, [P([(h1[:, sl[0]], w2b)], h2, bias=b2b, relu_cols=True), P([(h1[:, sl[1]], wd)], size, bias=bd),
                         P([(h1[:, sl[2]], wp)], depth, bias=bp), P([(h1[:, sl[3]], wa)], angle, bias=ba)])
        G(sgemm_ext.NT, [P([(h2, w3b)], delta, bias=b3b)])
        ctx.save_for_backward(x, h1, h2, w1b, w2b, w3b, w1d, wd, w1p, wp, w1a, wa, wc)
        shape = lambda t: t.view(B, Q, -1)                                                  # noqa: E731
        return shape(delta), shape(size), shape(depth), shape(angle), shape(logits), x.view_as(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_delta, g_size, g_depth, g_angle, g_logits, g_skip):
        x, h1, h2, w1b, w2b, w3b, w1d, wd, w1p, wp, w1a, wa, wc = ctx.saved_tensors
        B, Q, C = x.shape
        T, hid = B * Q, w1b.shape[0]
        x2 = x.view(T, C)

        def flat(g, n):                                                # an fp32 [T, n] gradient (an output nobody used: zeros)
            if g is None:
                return torch.zeros((T, n), dtype=torch.float32, device=x.device)
            return g.reshape(T, n).to(torch.float32).contiguous()
        g_delta, g_size, g_depth, g_angle, g_logits = (flat(g, w.shape[0]) for g, w in
                                                       ((g_delta, w3b), (g_size, wd), (g_depth, wp), (g_angle, wa), (g_logits, wc)))
        P, G = sgemm_ext.Problem, sgemm_ext.grouped
        sl = [slice(i * hid, (i + 1) * hid) for i in range(4)]
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)              # noqa: E731
        dh2, dh1 = new(T, hid), new(T, 4 * hid)
        G(sgemm_ext.NN, [P([(g_delta, w3b)], dh2, mask=h2)])
        G(sgemm_ext.NN, [P([(g, w)], dh1[:, s], mask=h1[:, s]) for g, w, s in zip((dh2, g_size, g_depth, g_angle), (w2b, wd, wp, wa), sl)])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            skip = g_skip.reshape(T, C) if g_skip is not None else None
            if skip is not None and not (skip.stride(1) == 1 and skip.dtype in (torch.float32, torch.bfloat16)):
                skip = skip.contiguous()
            G(sgemm_ext.NN, [P([(dh1[:, s], w) for s, w in zip(sl, (w1b, w1d, w1p, w1a))] + [(g_logits, wc)], dx.view(T, C), res=skip)])
        # weight and bias gradients: one group of ten products over the token axis
        outs = {}
        jobs = []
        for key, dy, act, w in (("3b", g_delta, h2, w3b), ("2b", dh2, h1[:, sl[0]], w2b), ("d", g_size, h1[:, sl[1]], wd),
                                ("p", g_depth, h1[:, sl[2]], wp), ("a", g_angle, h1[:, sl[3]], wa),
                                ("1b", dh1[:, sl[0]], x2, w1b), ("1d", dh1[:, sl[1]], x2, w1d), ("1p", dh1[:, sl[2]], x2, w1p),
                                ("1a", dh1[:, sl[3]], x2, w1a), ("c", g_logits, x2, wc)):
            dw, db = torch.empty_like(w), new(w.shape[0])
            outs[key] = (dw, db)
            jobs.append(P([(dy, act)], dw, colsum=db))
        G(sgemm_ext.TN, jobs)
        o = outs
        return (dx, o["1b"][0], o["1b"][1], o["2b"][0], o["2b"][1], o["3b"][0], o["3b"][1], o["1d"][0], o["1d"][1], o["d"][0], o["d"][1],
                o["1p"][0], o["1p"][1], o["p"][0], o["p"][1], o["1a"][0], o["1a"][1], o["a"][0], o["a"][1], o["c"][0], o["c"][1])


def heads_level(x, bbox, dim, dep, ang, cls):
    """-> (delta [B, Q, 6], size [B, Q, 3], depth [B, Q, 2], angle [B, Q, 24], logits [B, Q, classes], x') or None when the grouped
    kernels do not apply (other head structures, parameters not fp32, CPU tensors without the emulation backend)."""
    if not _heads_ok(x, bbox, dim, dep, ang, cls):
        return None
    lb, ld, lp, la = bbox.layers, dim.layers, dep.layers, ang.layers
    return _HeadsLevel.apply(x, lb[0].weight, lb[0].bias, lb[1].weight, lb[1].bias, lb[2].weight, lb[2].bias,
                             ld[0].weight, ld[0].bias, ld[1].weight, ld[1].bias, lp[0].weight, lp[0].bias, lp[1].weight, lp[1].bias,
                             la[0].weight, la[0].bias, la[1].weight, la[1].bias, cls.weight, cls.bias)
