"""dev: find the backward node of a captured training step that reads memory it does not own.
Dropout off -> replays are deterministic; replay, zero-fill every free block of the regular pool,
replay again, and list the parameters whose gradients changed."""
import os, sys, functools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from model_init import disable_dropout_
print = functools.partial(print, flush=True)

dev = torch.device("cuda", 0)
prec = sys.argv[1]
fill = float(os.environ.get("FILL", "0"))
step = bench.TrainStep(dev, 8, prec, graph=True)
disable_dropout_(step.raw_model)
images, calibs, img_sizes, targets = step.inputs
w = step.criterion.weight_dict


keep = {}


def part():
    step.optimizer.zero_grad(set_to_none=True)
    out = step.model(images, calibs, targets, img_sizes, dn_args=None)
    losses = step.criterion(out, targets, None)
    total = sum(losses[k] * w[k] for k in losses if k in w)
    total.backward()
    keep.clear()
    keep.update({"out/" + k: v for k, v in out.items() if torch.is_tensor(v)})
    keep.update({"loss/" + k: v for k, v in losses.items()})
    return total


side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for i in range(3):
        print("eager", i, float(part().detach()))
torch.cuda.synchronize()
eager = {n: p.grad.detach().float().cpu() for n, p in step.raw_model.named_parameters() if p.grad is not None}
g = torch.cuda.CUDAGraph()
step.optimizer.zero_grad(set_to_none=True)
with torch.cuda.graph(g, stream=side):
    res = part()
params = [(n, p) for n, p in step.raw_model.named_parameters() if p.grad is not None]


def snap():
    g.replay()
    torch.cuda.synchronize()
    d = {n: p.grad.detach().float().cpu() for n, p in params}
    d.update({k: v.detach().float().cpu() for k, v in keep.items()})
    return float(res.detach()), d


l0, g0 = snap()
l1, g1 = snap()
print("replay/replay loss", l0, l1, "grad diffs", sum(1 for n in g0 if not torch.equal(g0[n], g1[n])))
junk = [torch.full((64 << 20,), fill, device=dev) for _ in range(12)]
torch.cuda.synchronize()
del junk
l2, g2 = snap()
print("after poisoning the free pool: loss", l2)
rows = []
for n in g0:
    noise = float((g0[n] - g1[n]).abs().max())
    shift = float((g0[n] - g2[n]).abs().max())
    scale = float(g0[n].abs().max())
    rows.append((n, noise, shift, scale))
flag = [r for r in rows if r[2] > 20 * r[1] + 1e-6 * r[3] or r[2] != r[2]]
print(len(flag), "of", len(rows), "tensors moved by more than 20x the replay-to-replay noise")
for n, noise, shift, scale in flag:
    print("   %-70s noise %.3e shift %.3e scale %.3e" % (n, noise, shift, scale))
print("largest noise:", sorted(rows, key=lambda r: -r[1] / (r[3] + 1e-30))[:5])

for n, noise, shift, scale in flag:
    if n in eager:
        print(n, "\n  eager ", eager[n].flatten()[:6].tolist(), "\n  replay", g0[n].flatten()[:6].tolist(),
              "\n  poison", g2[n].flatten()[:6].tolist())
        print("  |replay-eager|", float((g0[n] - eager[n]).abs().max()), " |poison-eager|", float((g2[n] - eager[n]).abs().max()))
worst = sorted(((float((g0[n] - eager[n]).abs().max()) / (float(eager[n].abs().max()) + 1e-30), n) for n in eager), reverse=True)[:6]
print("replay vs eager, worst relative:", worst)
