"""oracle/resnet50_ref.py and the product's ResNet-50 body against a THIRD-PARTY implementation of the same network.

The reference takes its backbone from torchvision (lib/models/monodetr/backbone.py:100-102), which is neither vendored in the
reference tree nor installed here, so the restatement in oracle/resnet50_ref.py could only be checked against the published
definition by reading.  Hugging Face `transformers` (5.15.0 in this image; `transformers.models.resnet.modeling_resnet`, the
implementation its `microsoft/resnet-50` checkpoint -- converted from the torchvision / timm weights by
`convert_resnet_to_pytorch.py` -- runs on) is an independent public implementation of that network: v1.5 bottlenecks, the
stage's stride on the 3x3 convolution (`downsample_in_bottleneck=False`), a strided 1x1 shortcut on each stage's first block,
`nn.BatchNorm2d` in eval mode (= the reference's FrozenBatchNorm2d formula, eps 1e-5).  With one state_dict in the
reference's key names loaded into both: the three pyramid features and the gradients of the input and of every convolution
agree to 1e-10 in fp64 -- for the oracle, and directly for the product's body.  TEST INFRASTRUCTURE ONLY."""
import pytest
import torch

from model_init import name_seeded_init_
from oracle.resnet50_ref import resnet50_features

hf = pytest.importorskip("transformers.models.resnet.modeling_resnet")


def hf_name(key):
    """The reference's (torchvision's) state_dict key -> transformers' key."""
    parts = key.split(".")
    if parts[0] == "conv1":
        return "embedder.embedder.convolution." + parts[1]
    if parts[0] == "bn1":
        return "embedder.embedder.normalization." + parts[1]
    stage, block = int(parts[0][len("layer"):]) - 1, parts[1]
    where = "encoder.stages.%d.layers.%s." % (stage, block)
    if parts[2] == "downsample":
        return where + "shortcut." + ("convolution." if parts[3] == "0" else "normalization.") + parts[4]
    index = int(parts[2][-1]) - 1                                   # conv1..3 / bn1..3
    return where + "layer.%d." % index + ("convolution." if parts[2].startswith("conv") else "normalization.") + parts[3]


def third_party(sd, dtype):
    from transformers import ResNetConfig
    cfg = ResNetConfig()                                            # the defaults ARE ResNet-50
    assert (list(cfg.depths), list(cfg.hidden_sizes), cfg.embedding_size, cfg.layer_type) == ([3, 4, 6, 3], [256, 512, 1024, 2048], 64, "bottleneck")
    assert not cfg.downsample_in_first_stage and not cfg.downsample_in_bottleneck
    model = hf.ResNetModel(cfg).to(dtype).eval()
    mapped = {hf_name(k): v for k, v in sd.items()}
    own = model.state_dict()
    missing = [k for k in own if k not in mapped and not k.endswith("num_batches_tracked")]
    assert not missing and all(k in own for k in mapped), (missing[:5], [k for k in mapped if k not in own][:5])
    model.load_state_dict(mapped, strict=False)
    return model


def body_state(dtype):
    from monodetr_amd.monodetr.backbone import build_backbone
    from model_init import load_cfg
    torch.manual_seed(0)
    bb = build_backbone(load_cfg(device="cpu"))
    name_seeded_init_(bb)                                           # non-trivial running statistics and affine parameters
    return bb.to(dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
def test_the_restated_resnet50_equals_the_transformers_implementation(dtype, tol):
    bb = body_state(dtype)
    body = bb[0].body
    plain = {k: v.detach().clone() for k, v in body.state_dict().items()}
    model = third_party(plain, dtype)
    x = torch.randn(2, 3, 96, 160, dtype=dtype, generator=torch.Generator().manual_seed(1))
    x_hf, x_or, x_pr = (x.clone().requires_grad_(True) for _ in range(3))
    hidden = model(x_hf, output_hidden_states=True).hidden_states     # (stem, stage 1 .. 4)
    ref = hidden[2:]
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in plain.items()}
    oracle = resnet50_features(x_or, sd)
    body.prefold = False
    feats, _ = bb(x_pr)
    product = [f.tensors for f in feats]
    assert [tuple(t.shape) for t in ref] == [(2, 512, 12, 20), (2, 1024, 6, 10), (2, 2048, 3, 5)]
    for r, o, p in zip(ref, oracle, product):
        scale = max(1.0, r.abs().max().item())
        assert (o - r).abs().max() <= tol * scale and (p - r).abs().max() <= tol * scale
    # gradients: of the images and of every convolution weight
    proj = [torch.randn(r.shape, dtype=dtype, generator=torch.Generator().manual_seed(7 + i)) for i, r in enumerate(ref)]
    sum((r * q).sum() for r, q in zip(ref, proj)).backward()
    sum((o * q).sum() for o, q in zip(oracle, proj)).backward()
    sum((p * q).sum() for p, q in zip(product, proj)).backward()
    gtol = 1e-9 if dtype == torch.float64 else 2e-2                 # fp32: ReLU flips of near-zero pre-activations, see the twin test
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()      # noqa: E731
    assert rel(x_or.grad, x_hf.grad) <= gtol
    own = dict(model.named_parameters())
    checked = 0
    for name, p in body.named_parameters():
        if "conv" not in name and "downsample.0" not in name:
            continue
        want = own[hf_name(name)].grad
        assert rel(sd[name].grad, want) <= gtol, name                # the oracle has no frozen layers: all 53 convolutions
        if p.grad is not None:
            assert rel(p.grad, want) <= gtol, name                    # the product trains layer2-4 only (backbone.py:71-73)
            checked += 1
    assert checked >= 40
