"""Developer tool: count the ATen operator calls of one training step per source line (forward) and
per operator (backward), on the CPU -- a free proxy for the kernel-launch count of the GPU step (one
launch per elementwise / reduction / copy operator).

    python tests/opcount.py [--precision bf16] [--top 40]

Lives under tests/ because, like bench.py's cpu_baseline leg, it lets the CPU oracle stand in for the
MSDA operator (the product has no CPU path).
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

VIEW_OPS = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "t.default",
            "detach", "alias", "split", "unbind", "as_strided", "_unsafe_view", "unflatten", "flatten", "narrow",
            "_reshape_alias", "chunk", "lift_fresh", "is_", "size", "stride", "sym_", "numel", "dim", "empty", "unfold",
            "movedim", "diagonal")


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_site = collections.Counter()
        self.by_op = collections.Counter()
        self.by_site_op = collections.Counter()
        self.phase = "forward"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW_OPS):
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "monodetr_amd" in fr.filename:
                    site = "%s:%d" % (os.path.relpath(fr.filename, ROOT), fr.lineno)
                    break
            self.by_site[(self.phase, site)] += 1
            self.by_site_op[(self.phase, site, name)] += 1
            self.by_op[(self.phase, name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--op", default=None, help="also list the source lines of these operators (comma separated), e.g. _to_copy,add.Tensor")
    ap.add_argument("--device", default="cpu", help="cuda: the real step (committed switch list, B = 8, 384 x 1280) on the GPU")
    a = ap.parse_args()
    if a.device == "cuda":
        step = bench.TrainStep(torch.device("cuda", 0), 8, a.precision, switches=bench.committed_switches(a.precision)[0])
        for _ in range(3):
            step._step()
    else:
        from oracle import msda_oracle
        from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
        msda_oracle.build()
        F_.MSDA = msda_oracle.OracleMSDA
        step = bench.TrainStep(torch.device("cpu"), 2, a.precision, size=(96, 320))
    step._step()                                   # caches
    images, calibs, img_sizes, targets = step.inputs
    c = Counter()
    with c:
        step.optimizer.zero_grad(set_to_none=True)
        out = step.model(images, calibs, targets, img_sizes, dn_args=None)
        c.phase = "criterion"
        losses = step.criterion(out, targets, None)
        total = step.criterion.weighted_total(losses)
        c.phase = "backward"
        total.backward()
        c.phase = "optimizer"
        step.optimizer.step()
    tot = collections.Counter()
    for (ph, _), n in c.by_site.items():
        tot[ph] += n
    print("non-view ATen calls per phase:", dict(tot), "total", sum(tot.values()))
    for ph in ("forward", "criterion"):
        print("\n== %s: top source lines" % ph)
        for (p, site), n in sorted(((k, v) for k, v in c.by_site.items() if k[0] == ph), key=lambda kv: -kv[1])[:a.top]:
            print("%5d  %s" % (n, site))
    for opname in (a.op.split(",") if a.op else []):
        print("\n== source lines of %s" % opname)
        for (p, site, op), n in sorted(((k, v) for k, v in c.by_site_op.items() if opname in k[2]), key=lambda kv: -kv[1])[:a.top]:
            print("%5d  %-10s %s" % (n, p, site))
    for ph in ("forward", "criterion", "backward", "optimizer"):
        print("\n== %s: top operators" % ph)
        for (p, op), n in sorted(((k, v) for k, v in c.by_op.items() if k[0] == ph), key=lambda kv: -kv[1])[:25]:
            print("%5d  %s" % (n, op))


if __name__ == "__main__":
    main()
