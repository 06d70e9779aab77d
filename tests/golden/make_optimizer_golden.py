"""tests/golden/optimizer_adamw.npz: six steps of the REFERENCE's AdamW (lib/helpers/optimizer_helper.py:30-129,
via its build_optimizer :7-27) on a small seeded model, recorded for tests/test_optimizer.py.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_optimizer_golden.py"""
import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optimizer_problem import make_model, make_grads  # noqa: E402

warnings.filterwarnings("ignore")
from lib.helpers.optimizer_helper import build_optimizer  # noqa: E402

model = make_model()
opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4}, model)
rec = {}
for step in range(6):
    make_grads(model, step)
    opt.step()
    if step in (0, 5):
        for n, p in model.named_parameters():
            rec["step%d/%s" % (step, n)] = p.detach().numpy().copy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "optimizer_adamw.npz"), **rec)
print("saved", sorted(rec)[:4], len(rec))
