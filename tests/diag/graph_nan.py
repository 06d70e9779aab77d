"""Diagnostic (not a test): where does a non-finite value first appear when the replayed iteration runs on changing batches?
    python tests/diag/graph_nan.py [--no-ref] [--seed0 50] [--steps 12] [--lr 0]"""
import argparse
import os
import sys

if os.environ.get("MDETR_DIAG_RUNTIME_DEFAULTS") != "1":                 # (=1: leave the runtime's own defaults alone)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import monodetr_amd._runtime_env  # noqa: F401

import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_trainer_gpu import build, collated_batch  # noqa: E402


def bad(t):
    return t is not None and torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())


def detail(it, twin):
    """Per non-finite gradient: how much of it is bad, and whether the finite part agrees with the eagerly computed twin's."""
    tw = dict(twin.raw_model.named_parameters()) if twin is not None else {}
    for n, p in it.raw_model.named_parameters():
        if bad(p.grad):
            g = p.grad.float()
            fin = torch.isfinite(g)
            msg = "   grad %s shape=%s non-finite %d of %d (nan %d, inf %d)" % (n, tuple(g.shape), int((~fin).sum()), g.numel(), int(torch.isnan(g).sum()), int(torch.isinf(g).sum()))
            if n in tw and tw[n].grad is not None:
                t = tw[n].grad.float()
                msg += " | finite part vs eager twin: max|diff| %.3g of scale %.3g" % (float(((g - t).abs() * fin).max()), float(t.abs().max()))
            idx = (~fin).flatten().nonzero().flatten()
            msg += " | first bad flat indices %s" % idx[:6].tolist()
            print(msg, flush=True)


def report(tag, it):
    out = []
    for k, v in (it.losses or {}).items():
        if bad(v):
            out.append("loss:" + k)
    n_p = [n for n, p in it.raw_model.named_parameters() if bad(p)]
    n_g = [n for n, p in it.raw_model.named_parameters() if bad(p.grad)]
    n_s = []
    for p, st in it.optimizer.state.items():
        for k, v in st.items():
            if bad(v):
                n_s.append(k)
    st_bad = [i for i, t in enumerate(_leaves(it.static)) if bad(t)] if it.static is not None else []
    print(tag, "bad losses", out[:6], "| params", len(n_p), n_p[:3], "| grads", len(n_g), n_g[:4], "| opt state", len(n_s), sorted(set(n_s)), "| static", st_bad, flush=True)
    return bool(out or n_p or n_g or n_s)


def _leaves(x, out=None):
    out = [] if out is None else out
    if torch.is_tensor(x):
        out.append(x)
    elif isinstance(x, dict):
        for v in x.values():
            _leaves(v, out)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _leaves(v, out)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--seed0", type=int, default=50)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--lr", type=float, default=0.0)
    ap.add_argument("--sync-load", action="store_true", help="synchronise after the static buffers are filled")
    ap.add_argument("--fresh-wgrad", action="store_true", help="small_wgrad returns freshly computed tensors instead of views of one buffer")
    ap.add_argument("--drop", default="", help="comma-separated kernel families to leave out")
    ap.add_argument("--sleep", type=float, default=0.0, help="host sleep between iterations (instead of the per-tensor checks)")
    ap.add_argument("--quiet", action="store_true", help="check the total loss only (no per-tensor reads between replays)")
    ap.add_argument("--settle", type=float, default=0.0, help="host sleep between the synchronisation and the per-tensor checks")
    ap.add_argument("--dump", default="", help="write the captured graph's DOT dump here")
    a = ap.parse_args()
    import bench
    from monodetr_amd.helpers.trainer_helper import TARGET_KEYS
    dev = torch.device("cuda", 0)
    switches = bench.committed_switches("bf16")[0] - {s_ for s_ in a.drop.split(",") if s_}
    if a.fresh_wgrad:
        from monodetr_amd import small_wgrad_ext
        real = small_wgrad_ext.small_wgrad

        def fresh(*args, **kw):
            dw, db = real(*args, **kw)
            return dw + 0, db + 0                                 # elementwise kernels: fresh, stealable tensors (no D2D memcpy by autograd)
        small_wgrad_ext.small_wgrad = fresh
    it, _ = build(dev, True, switches)
    if a.dump:
        real_capture = it.capture

        def capture(batch, in_place=False):
            import torch.cuda
            orig = torch.cuda.CUDAGraph

            def make(*x, **k):
                g_ = orig(*x, **k)
                g_.enable_debug_mode()
                capture.graphs.append(g_)
                return g_
            capture.graphs = []
            torch.cuda.CUDAGraph = make
            try:
                return real_capture(batch, in_place)
            finally:
                torch.cuda.CUDAGraph = orig
                for k_, g_ in enumerate(capture.graphs):
                    g_.debug_dump("%s.%d.dot" % (a.dump, k_))
        it.capture = capture
    for g_ in it.optimizer.param_groups:
        g_['lr'].fill_(a.lr)
    ref = None
    if not a.no_ref:
        ref, _ = build(dev, False, switches)
        ref.raw_model.load_state_dict(it.raw_model.state_dict())
        for g_ in ref.optimizer.param_groups:
            g_['lr'] = a.lr
    if a.sync_load:
        load = it.load

        def synced(batch):
            load(batch)
            torch.cuda.synchronize()
        it.load = synced
    for i in range(a.steps):
        images, calibs, t = collated_batch(2, seed=a.seed0 + i)
        images = images.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        batch = (images, calibs.to(dev), t['img_size'], {k: t[k] for k in TARGET_KEYS})
        before = it.replays
        x = float(it.run(batch))
        torch.cuda.synchronize()
        if a.settle:
            import time
            time.sleep(a.settle)
        if a.quiet:
            import math
            import time
            print("i=%d %s loss=%.4f" % (i, "replay" if it.replays > before else "eager ", x), flush=True)
            if a.sleep:
                time.sleep(a.sleep)
            if not math.isfinite(x):
                report("   ", it)
                detail(it, ref)
                break
            continue
        hit = report("i=%d %s loss=%.4f objects=%s" % (i, "replay" if it.replays > before else "eager ", x, t['mask_2d'].sum(1).tolist()), it)
        if ref is not None:
            y = float(ref.run(batch))
            print("      eager twin loss=%.4f" % y, flush=True)
        if hit:
            detail(it, ref)
            break


if __name__ == "__main__":
    main()
