"""How close do the sampling locations of the float64 model come to cell boundaries, where d(output)/d(location) of deformable attention is
discontinuous?  (CPU, test infrastructure: the float64 model of tests/test_model_gpu.py with the C oracle as its MSDA operator.)
    python tests/diag/msda_boundary_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden
from model_init import name_seeded_init_, disable_dropout_
from test_model_gpu import load_cfg, synthetic_batch
from oracle import msda_oracle
msda_oracle.build()
from monodetr_amd.monodetr import build_monodetr
from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_

rec = []
class Rec(msda_oracle.OracleMSDA):
    pass
orig = msda_oracle.OracleMSDA.ms_deform_attn_forward
def fwd(value, shapes, lstart, loc, attn, step):
    rec.append((shapes.clone(), loc.detach().clone(), attn.detach().clone()))
    return orig(value, shapes, lstart, loc, attn, step)
msda_oracle.OracleMSDA.ms_deform_attn_forward = staticmethod(fwd)
F_.MSDA = msda_oracle.OracleMSDA
torch.manual_seed(0)
m64, c64 = build_monodetr(load_cfg())
disable_dropout_(name_seeded_init_(m64)).double().train()
images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7)
t64 = [{k: (v.double() if v.is_floating_point() else v) for k, v in t.items()} for t in targets]
with torch.no_grad():
    m64(images.double(), calibs.double(), t64, img_sizes)
for i, (shapes, loc, attn) in enumerate(rec):
    L = shapes.shape[0]
    H = shapes[:, 0].view(1, 1, 1, L, 1).double(); W = shapes[:, 1].view(1, 1, 1, L, 1).double()
    px = loc[..., 0] * W - 0.5; py = loc[..., 1] * H - 0.5
    dx = (px - px.round()).abs(); dy = (py - py.round()).abs()
    d = torch.minimum(dx, dy)
    inside = (px > -1) & (px < W) & (py > -1) & (py < H)
    near = (d < 1e-5) & inside
    print("call %d Lq=%d: min distance to a cell boundary %.3e; samples within 1e-5: %d, within 1e-6: %d (attention weight of the closest: %.3e)" % (
        i, loc.shape[1], float(d[inside].min()), int(near.sum()), int(((d < 1e-6) & inside).sum()), float(attn.flatten()[d.flatten().argmin()])))
