"""Generate tests/golden/msda_*.npz from the REFERENCE's own code (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports ``ms_deform_attn_core_pytorch`` unmodified from
/root/reference/lib/models/monodetr/ops/functions/ms_deform_attn_func.py:41-61 (the extension
import at :18 is satisfied by a stub module) and records its outputs and autograd gradients.
/root/reference does not exist on the GPU box, so the vectors are committed; nothing at test time
reads the reference tree.

Cases
  ref_test_f64 / ref_test_f32   the reference's own test problem (ops/test.py:21-37): seed 3,
                                N=1 M=2 D=2 Lq=2 L=2 P=2, shapes (6,4),(3,2)
  grad_d{30,32,64,71}           ops/test.py:63-78,85-86 gradcheck problems (channels=D), fp64, with
                                the reference's autograd gradients for a fixed grad_output
  kitti_small                   M=8 D=32 L=4 P=4 (the default config's head geometry), B=2, levels
                                (12,40),(6,20),(3,10),(2,5), Lq=96, sampling locations in
                                [-0.2,1.2) so borders / out-of-window samples are exercised
  border                        hand-placed locations on and around every window edge
"""
import os
import sys
import types

import numpy as np
import torch

REF_OPS = "/root/reference/lib/models/monodetr/ops"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    sys.path.insert(0, REF_OPS)
    from functions.ms_deform_attn_func import ms_deform_attn_core_pytorch  # noqa
    return ms_deform_attn_core_pytorch


def level_start(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def run_ref(ref, value, shapes, loc, attn, grad_out=None):
    value = value.clone().requires_grad_(grad_out is not None)
    loc = loc.clone().requires_grad_(grad_out is not None)
    attn = attn.clone().requires_grad_(grad_out is not None)
    out = ref(value, shapes, loc, attn)
    res = {"out": out.detach()}
    if grad_out is not None:
        out.backward(grad_out)
        res.update(grad_value=value.grad, grad_loc=loc.grad, grad_attn=attn.grad)
    return res


def save(name, shapes, value, loc, attn, res, grad_out=None):
    d = dict(shapes=shapes.numpy(), level_start=level_start(shapes).numpy(), value=value.numpy(),
             loc=loc.numpy(), attn=attn.numpy())
    if grad_out is not None:
        d["grad_out"] = grad_out.numpy()
    d.update({k: v.numpy() for k, v in res.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, {k: (v.shape, str(v.dtype)) for k, v in d.items()})


def problem(N, M, D, Lq, shapes, P, dtype, lo=0.0, hi=1.0):
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, M, D) * 0.01).to(dtype)
    loc = (torch.rand(N, Lq, M, L, P, 2) * (hi - lo) + lo).to(dtype)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    return value, loc, attn


def main():
    ref = load_reference()
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)

    # the reference's own forward checks (ops/test.py:32-60)
    torch.manual_seed(3)
    v, l, a = problem(1, 2, 2, 2, shapes, 2, torch.float64)
    save("msda_ref_test_f64", shapes, v, l, a, run_ref(ref, v, shapes, l, a))
    v, l, a = problem(1, 2, 2, 2, shapes, 2, torch.float32)
    save("msda_ref_test_f32", shapes, v, l, a, run_ref(ref, v, shapes, l, a))

    # gradcheck problems (ops/test.py:63-78); 1025/2048/3096 omitted from the fixtures for size,
    # they are covered by oracle-vs-HIP and numerical gradcheck tests instead
    for D in (30, 32, 64, 71):
        v, l, a = problem(1, 2, D, 2, shapes, 2, torch.float64)
        g = torch.randn(1, 2, 2 * D, dtype=torch.float64)
        save("msda_grad_d%d" % D, shapes, v, l, a, run_ref(ref, v, shapes, l, a, g), g)

    # default-config head geometry at reduced spatial size
    kshapes = torch.as_tensor([(12, 40), (6, 20), (3, 10), (2, 5)], dtype=torch.long)
    v, l, a = problem(2, 8, 32, 96, kshapes, 4, torch.float32, lo=-0.2, hi=1.2)
    g = torch.randn(2, 96, 256, dtype=torch.float32)
    r64 = run_ref(ref, v.double(), kshapes, l.double(), a.double(), g.double())
    r32 = run_ref(ref, v, kshapes, l, a, g)
    res = {k + "_f64": x for k, x in r64.items()}
    res.update({k + "_f32": x for k, x in r32.items()})
    save("msda_kitti_small", kshapes, v, l, a, res, g)

    # window edges: pixel coordinate h_im = loc*H - 0.5 at -1, -1+eps, -0.5, 0, H-1, H-0.5, H-eps, H
    bshapes = torch.as_tensor([(4, 8), (2, 4)], dtype=torch.long)
    pix = torch.tensor([-1.0, -0.999, -0.5, -0.25, 0.0, 0.5, 2.75, 3.0, 3.25, 3.5, 3.999, 4.0, 7.0, 7.5, 7.99, 8.0])
    Lq, M, L, P = pix.numel(), 2, 2, 2
    loc = torch.empty(1, Lq, M, L, P, 2, dtype=torch.float64)
    for lvl, (H, W) in enumerate(bshapes.tolist()):
        for p in range(P):
            for m in range(M):
                # x sweeps the edge list, y sweeps it reversed (and swapped for the 2nd head)
                xs = (pix.double() + 0.5) / W
                ys = (pix.double().flip(0) + 0.5) / H
                loc[0, :, m, lvl, p, 0] = xs if m == 0 else ys * H / W
                loc[0, :, m, lvl, p, 1] = ys if m == 0 else xs * W / H
    torch.manual_seed(5)
    v, _, a = problem(1, M, 4, Lq, bshapes, P, torch.float64)
    g = torch.randn(1, Lq, M * 4, dtype=torch.float64)
    save("msda_border", bshapes, v, loc, a, run_ref(ref, v, bshapes, loc, a, g), g)


if __name__ == "__main__":
    main()
