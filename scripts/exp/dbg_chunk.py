import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import monodetr_amd._runtime_env
import torch, bench
from monodetr_amd import chunk_sums
from model_init import disable_dropout_
dev = torch.device("cuda", 0)
names = tuple(sorted(set(bench.COMMITTED_SWITCHES["bf16"])))
res = {}
for mode in ("immediate", "deferred"):
    chunk_sums.IMMEDIATE, chunk_sums.POISON = mode == "immediate", mode == "deferred"
    step = bench.TrainStep(dev, 8, "bf16", size=(384, 1280), switches=names, graph=True)     # graph wanted: eager iterations run on the side stream
    disable_dropout_(step.raw_model)
    step.optimizer.step = lambda *a, **k: None                                                # gradients only
    step._eager(step.inputs)
    torch.cuda.synchronize()
    res[mode] = {n: p.grad.detach().float().clone() for n, p in step.raw_model.named_parameters() if p.grad is not None}
    del step
a, b = res["immediate"], res["deferred"]
bad = [(float((a[n] - b[n]).abs().max() / (a[n].abs().max() + 1e-20)) if torch.isfinite(b[n]).all() else float("inf"), n) for n in a]
bad = sorted([x for x in bad if not x[0] == 0.0], reverse=True)
print("differing:", len(bad), "of", len(a))
for x in bad[:60]:
    print("  %.3g %s" % x)
