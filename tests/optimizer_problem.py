"""Seeded toy problem shared by tests/golden/make_optimizer_golden.py and tests/test_optimizer.py."""
import torch
from torch import nn


def make_model():
    torch.manual_seed(11)
    m = nn.Sequential(nn.Linear(7, 5), nn.LayerNorm(5), nn.Linear(5, 3, bias=False))
    m.add_module("late", nn.Linear(3, 2))       # receives gradients only from step 2 on
    return m.double()


def make_grads(model, step):
    g = torch.Generator().manual_seed(100 + step)
    for n, p in model.named_parameters():
        if n.startswith("late") and step < 2:
            p.grad = None
        else:
            p.grad = torch.randn(p.shape, generator=g, dtype=p.dtype) * (1 + step)
