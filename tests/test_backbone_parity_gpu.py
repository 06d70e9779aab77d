"""GPU twin of tests/test_backbone_parity_cpu.py: the ResNet-50 body as the benchmark runs it (channels_last, folded frozen
BN, 1x1 convolutions as token GEMMs, MIOpen or csrc/conv3x3.hip for the 3x3s, csrc/bias_act.hip tails) against the
independent restatement of the torchvision definition (oracle/resnet50_ref.py) evaluated with plain F.conv2d on the same
GPU in fp32 -- north_star bar 1e-3 -- and the bf16 body against it at bf16 accuracy."""
import pytest
import torch

from model_init import load_cfg, name_seeded_init_
from oracle.resnet50_ref import resnet50_features

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("switches", ["default", "committed"])
def test_backbone_fp32_matches_the_torchvision_definition_on_the_gpu(switches):
    import bench
    from monodetr_amd.monodetr.backbone import build_backbone
    names = set(bench.COMMITTED_SWITCHES["fp32"]) - {"MDETR_FUSED_ADAMW", "MDETR_FUSED_LOSSES"} if switches == "committed" else set()
    bench.apply_switches(names)
    try:
        torch.manual_seed(0)
        bb = name_seeded_init_(build_backbone(load_cfg(device="cuda"))).cuda().to(memory_format=torch.channels_last)
        body = bb[0].body
        x = torch.randn(2, 3, 384, 1280, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in body.state_dict().items()}
        want = resnet50_features(x, sd)
        feats, _ = bb(x.contiguous(memory_format=torch.channels_last))
        got = [f.tensors for f in feats]
        assert [tuple(g.shape) for g in got] == [(2, 512, 48, 160), (2, 1024, 24, 80), (2, 2048, 12, 40)]
        for g, w in zip(got, want):
            assert (g - w).abs().max() <= 1e-3 * max(1.0, w.abs().max().item())
        proj = [torch.randn(w.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7 + i)) for i, w in enumerate(want)]
        sum((g * p).sum() for g, p in zip(got, proj)).backward()
        sum((w * p).sum() for w, p in zip(want, proj)).backward()
        worst, n = 0.0, 0
        for name, p in body.named_parameters():
            if p.grad is None:
                continue
            ref = sd[name].grad
            worst = max(worst, ((p.grad - ref).norm() / ref.norm().clamp_min(1e-30)).item())
            n += 1
        assert n >= 40 and worst <= 2e-2, (n, worst)
    finally:
        bench.apply_switches(set())


def test_backbone_bf16_body_tracks_the_torchvision_definition():
    """The bf16 configuration of the benchmark (fp32 backbone parameters folded and cast once per step, bf16 activations,
    csrc/conv3x3.hip + csrc/bias_act.hip) against the fp32 restatement: cosine per feature map, relative L2."""
    import bench
    from monodetr_amd.monodetr.backbone import build_backbone
    bench.apply_switches(set(bench.COMMITTED_SWITCHES["bf16"]) - {"MDETR_FUSED_ADAMW", "MDETR_FUSED_LOSSES"})
    try:
        torch.manual_seed(0)
        bb = name_seeded_init_(build_backbone(load_cfg(device="cuda"))).cuda().to(memory_format=torch.channels_last)
        body = bb[0].body
        x = torch.randn(2, 3, 384, 1280, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        with torch.no_grad():
            want = resnet50_features(x, body.state_dict())
        feats, _ = bb(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
        for f, w in zip(feats, want):
            g = f.tensors.float()
            assert g.shape == w.shape
            cos = torch.nn.functional.cosine_similarity(g.flatten(), w.flatten(), dim=0).item()
            rel = ((g - w).norm() / w.norm()).item()
            assert cos >= 0.999 and rel <= 3e-2, (cos, rel)
    finally:
        bench.apply_switches(set())
