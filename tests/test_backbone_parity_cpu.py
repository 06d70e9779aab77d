"""The product's ResNet-50 body (monodetr_amd/monodetr/backbone.py: folded frozen BN, 1x1 convolutions as token GEMMs,
optional HIP convolution tails) against the independent functional restatement of the torchvision definition in
oracle/resnet50_ref.py -- values and gradients.  (The full-model golden vectors use the product's own ResNet as the
torchvision stand-in, so they cannot pin the backbone; this does.)  The GPU twin is tests/test_backbone_parity_gpu.py."""
import pytest
import torch

from model_init import name_seeded_init_
from oracle.resnet50_ref import resnet50_features


def build(dtype):
    from monodetr_amd.monodetr.backbone import build_backbone
    from model_init import load_cfg
    torch.manual_seed(0)
    bb = build_backbone(load_cfg(device="cpu"))
    name_seeded_init_(bb)
    return bb.to(dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
@pytest.mark.parametrize("prefold", [False, True])
def test_backbone_features_and_gradients_match_the_torchvision_definition(dtype, tol, prefold):
    bb = build(dtype)
    body = bb[0].body
    body.prefold = prefold
    x = torch.randn(2, 3, 96, 160, dtype=dtype, generator=torch.Generator().manual_seed(1))
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in body.state_dict().items()}
    want = resnet50_features(x, sd)
    feats, _ = bb(x)
    got = [f.tensors for f in feats]
    assert [tuple(g.shape) for g in got] == [(2, 512, 12, 20), (2, 1024, 6, 10), (2, 2048, 3, 5)]
    for g, w in zip(got, want):
        assert (g - w).abs().max() <= tol * max(1.0, w.abs().max().item())
    # gradients of every trainable convolution (layer2-4; the stem and layer1 are frozen, backbone.py:71-73)
    proj = [torch.randn(w.shape, dtype=dtype, generator=torch.Generator().manual_seed(7 + i)) for i, w in enumerate(want)]
    sum((g * p).sum() for g, p in zip(got, proj)).backward()
    sum((w * p).sum() for w, p in zip(want, proj)).backward()
    n, worst = 0, 0.0
    for name, p in body.named_parameters():
        if p.grad is None:
            assert not p.requires_grad or name.startswith(("conv1", "layer1"))
            continue
        ref = sd[name].grad
        # relative L2 per tensor (in fp32 a handful of pre-activations within rounding of zero flip their ReLU between the two
        # evaluation orders; a max-norm comparison would measure those isolated elements)
        worst = max(worst, ((p.grad - ref).norm() / ref.norm().clamp_min(1e-30)).item())
        n += 1
    assert n >= 40
    # fp32: two different but equally valid evaluation orders (folded weights, GEMM forms) of a 50-layer backward pass
    assert worst <= (1e-9 if dtype == torch.float64 else 2e-2), worst          # structure is pinned in fp64; fp32 measures rounding
