"""Is the FIRST backward pass of a process different from the second?  (fp32 model, same batch, same parameters)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import monodetr_amd._runtime_env  # noqa
import torch
from model_init import disable_dropout_, load_cfg, name_seeded_init_, synthetic_batch
from monodetr_amd.monodetr import build_monodetr

torch.manual_seed(0)
model, criterion = build_monodetr(load_cfg(device="cuda"))
disable_dropout_(name_seeded_init_(model)).cuda().train()
criterion.train()
images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7, device="cuda")
grads = []
for it in range(3):
    model.zero_grad(set_to_none=True)
    out = model(images, calibs, targets, img_sizes)
    losses = criterion(out, targets)
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    total.backward()
    grads.append({n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None})
    print("iteration", it, "loss", float(total))
big = max(float(g.norm()) for g in grads[0].values())
for a, b, tag in ((0, 1, "first vs second"), (1, 2, "second vs third")):
    rel = sorted(((float((grads[a][n] - grads[b][n]).norm() / grads[b][n].norm().clamp_min(1e-30)), n) for n in grads[b]
                  if float(grads[b][n].norm()) > 1e-5 * big), reverse=True)
    print(tag, [("%.2e" % r, n.replace("depthaware_transformer.", "")) for r, n in rel[:8]])
    for n in ("depthaware_transformer.encoder.layers.1.self_attn.sampling_offsets.weight", "backbone.0.body.layer2.0.conv1.weight"):
        print("   ", n, "%.3e" % float((grads[a][n] - grads[b][n]).norm() / grads[b][n].norm()))
