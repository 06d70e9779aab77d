"""Diagnostic (not a test): where does a non-finite value first appear when the replayed iteration runs on changing batches?
    python tests/diag/graph_nan.py [--no-ref] [--seed0 50] [--steps 12] [--lr 0]"""
import argparse
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_trainer_gpu import build, collated_batch  # noqa: E402


def bad(t):
    return t is not None and torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())


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
    a = ap.parse_args()
    import bench
    from monodetr_amd.helpers.trainer_helper import TARGET_KEYS
    dev = torch.device("cuda", 0)
    switches = bench.committed_switches("bf16")[0]
    it, _ = build(dev, True, switches)
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
        hit = report("i=%d %s loss=%.4f objects=%s" % (i, "replay" if it.replays > before else "eager ", x, t['mask_2d'].sum(1).tolist()), it)
        if ref is not None:
            y = float(ref.run(batch))
            print("      eager twin loss=%.4f" % y, flush=True)
        if hit:
            break


if __name__ == "__main__":
    main()
