"""csrc/tgemm.hip on the GPU: every token-wise product shape of the training iteration (forward NT and input-gradient NN forms, the
tails each call site fuses) held ELEMENT BY ELEMENT to the fp64 product of the same bf16 operands (tests/gemm_bounds.py: output
rounding + fp32 accumulation; no absolute-of-max bound), each tile shape and pipeline depth on one awkward shape, determinism, and
the C ABI's argument checks."""
import pytest
import torch

from gemm_bounds import assert_product_close
from conftest import tune

pytestmark = pytest.mark.gpu

T1, T2, T3, T4, TE, TD = 245760, 61440, 15360, 3840, 81600, 4400
STEP_SHAPES = [
    # T, K, N, nn, tail
    (TE, 256, 256, False, "bias"), (TE, 256, 256, False, "relu_bias"), (TE, 256, 384, False, "bias"),
    (TE, 256, 256, True, "accum"), (TE, 384, 256, True, ""),
    (T1, 256, 64, False, "relu_bias"), (T1, 64, 256, False, "res_relu"),
    (T2, 512, 128, False, "relu_bias"), (T2, 128, 512, False, "res_relu"), (T2, 128, 512, True, "accum"), (T2, 512, 128, True, ""),
    (T3, 1024, 256, False, "relu_bias"), (T3, 256, 1024, False, "res_relu"), (T3, 256, 1024, True, "accum"), (T3, 1024, 256, True, ""),
    (T4, 2048, 512, False, "relu_bias"), (T4, 512, 2048, False, "res_relu"), (T4, 512, 2048, True, "accum"), (T4, 2048, 512, True, ""),
    (T4, 2048, 256, False, "bias"),
    (TD, 256, 1032, False, "bias"), (TD, 1032, 256, True, ""), (TD, 256, 256, False, "relu_bias"),
]


def _operands(T, K, N, nn, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = ((torch.randn(K, N, generator=g) if nn else torch.randn(N, K, generator=g)) * 0.1).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    r = torch.randn(T, N, generator=g).to(torch.bfloat16).to(dev)
    return a, w, b, r


def _reference(a, w, nn, bias, res, relu):
    wd = w.double() if nn else w.double().t()
    ref = a.double() @ wd
    mag = a.double().abs() @ wd.abs()
    if bias is not None:
        ref += bias.double()
        mag += bias.double().abs()
    if res is not None:
        ref += res.double()
        mag += res.double().abs()
    return (ref.clamp_(min=0) if relu else ref), mag


@pytest.mark.parametrize("T,K,N,nn,tail", STEP_SHAPES)
def test_tgemm_step_shapes_element_wise_against_fp64(T, K, N, nn, tail):
    from monodetr_amd import tgemm_ext
    dev = torch.device("cuda", 0)
    a, w, b, r = _operands(T, K, N, nn, T + 3 * K + N, dev)
    bias = b if "bias" in tail else None
    res = r.clone() if tail in ("res_relu", "accum") else None
    relu = "relu" in tail
    assert tgemm_ext.supported(a, w, nn=nn, res=res, bias=bias)
    out = res if tail == "accum" else None
    keep = res.clone() if res is not None else None
    y = tgemm_ext.tgemm(a, w, bias, res, relu=relu, nn=nn, out=out)
    torch.cuda.synchronize()
    ref, mag = _reference(a, w, nn, bias, keep, relu)
    assert_product_close(y, ref, mag, K, "T=%d K=%d N=%d nn=%s %s" % (T, K, N, nn, tail))
    y2 = tgemm_ext.tgemm(a, w, bias, keep, relu=relu, nn=nn)        # (out of place, also where y was accumulated into its residual)
    assert torch.equal(y, y2)                                        # deterministic


@pytest.mark.parametrize("tile", ["128x128", "128x64", "64x128", "64x64"])
@pytest.mark.parametrize("pf", ["1", "2"])
@pytest.mark.parametrize("nn", [False, True])
def test_tgemm_every_tile_shape_and_pipeline_depth(monkeypatch, tile, pf, nn):
    from monodetr_amd import tgemm_ext
    tune(monkeypatch, tgemm_tile=tile)
    tune(monkeypatch, tgemm_pf=pf)
    dev = torch.device("cuda", 0)
    for T, K, N in ((4133, 456, 264), (300, 64, 72), (9000, 1032, 136)):
        a, w, b, r = _operands(T, K, N, nn, T + K, dev)
        y = tgemm_ext.tgemm(a, w, b.float(), r, relu=True, nn=nn)
        ref, mag = _reference(a, w, nn, b, r, True)
        assert_product_close(y, ref, mag, K, "%s pf%s nn=%s T=%d" % (tile, pf, nn, T))
        y32 = tgemm_ext.tgemm(a, w, None, None, nn=nn, out_dtype=torch.float32)
        ref, mag = _reference(a, w, nn, None, None, False)
        assert_product_close(y32, ref, mag, K, "fp32 out")


def test_tgemm_dropout_tail_is_the_bias_act_decision():
    from monodetr_amd import bias_act_ext, tgemm_ext
    dev = torch.device("cuda", 0)
    T, K, N = 81600, 256, 256
    a, w, b, _ = _operands(T, K, N, False, 9, dev)
    y = tgemm_ext.tgemm(a, w, b, None, relu=True, dropout_p=0.1, seed=77)
    pre = tgemm_ext.tgemm(a, w, b, None, out_dtype=torch.float32)
    want = bias_act_ext.bias_act(pre, None, None, relu=True, dropout_p=0.1, seed=77).to(torch.bfloat16)
    assert torch.equal(y, want)
    frac = (y == 0).float().mean().item()
    assert 0.5 < frac < 0.6                                          # half negative + a tenth of the rest dropped


def test_training_step_with_the_token_gemm_kernel_matches_default():
    """The whole bf16 training iteration with MDETR_TGEMM on top of the other committed families against the same list without it:
    loss trajectories of three optimizer steps (dropout off: the two routes draw their masks at different sites)."""
    import bench
    from model_init import disable_dropout_
    dev = torch.device("cuda", 0)
    base = tuple(sorted(set(bench.COMMITTED_SWITCHES["bf16"]) - {"MDETR_TGEMM"}))
    traj = {}
    try:
        for names in (base, base + ("MDETR_TGEMM",)):
            step = bench.TrainStep(dev, 8, "bf16", size=(192, 640), switches=names)
            disable_dropout_(step.raw_model)
            traj[names] = [float(step()) for _ in range(3)]
    finally:
        bench.apply_switches(set())
    for a, b in zip(traj[base], traj[base + ("MDETR_TGEMM",)]):
        assert abs(a - b) <= 2e-2 * abs(a), traj


def test_bottleneck_with_fused_tails_on_the_gpu_is_as_close_to_fp32_as_the_default_route(monkeypatch):
    """ResNet bottlenecks at layer2's shape (B = 8, 48 x 160) with the conv1 / conv3 tails in csrc/tgemm.hip's epilogue: outputs and
    every gradient against the block evaluated in fp32 on the same bf16-valued parameters, no worse than the default bf16 route."""
    import copy
    from monodetr_amd.monodetr import backbone, linear
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)

    def rel(a, b):
        return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()

    for down in (False, True):
        inpl, planes = (512, 128) if not down else (256, 128)
        ds = torch.nn.Sequential(torch.nn.Conv2d(inpl, planes * 4, 1, 1, bias=False), backbone.FrozenBatchNorm2d(planes * 4)) if down else None
        blk = backbone.Bottleneck(inpl, planes, 1, ds).to(dev)
        for m in blk.modules():
            if isinstance(m, backbone.FrozenBatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
        x = (torch.randn(8, inpl, 48, 160, device=dev) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for p in blk.parameters():                                     # bf16-valued fp32 parameters (the backbone keeps fp32 masters)
            p.data = p.data.to(torch.bfloat16).float()
        res = {}
        for on in (False, True):
            monkeypatch.setattr(linear, "_TGEMM", on)
            for p in blk.parameters():
                p.grad = None
            x.grad = None
            pairs = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)] + ([(ds[0], ds[1])] if down else [])
            backbone.prefold(pairs, torch.bfloat16)
            y = blk(x)
            (y.float() * 0.01).sum().backward()
            res[on] = (y.detach().float(), x.grad.float().clone(), {n: p.grad.float().clone() for n, p in blk.named_parameters()})
        monkeypatch.setattr(linear, "_TGEMM", False)
        ref = copy.deepcopy(blk).double()
        x64 = x.detach().double().requires_grad_(True)
        y64 = ref(x64)
        (y64 * 0.01).sum().backward()
        assert rel(res[True][0], y64.detach()) <= max(1.2 * rel(res[False][0], y64.detach()), 6e-3)
        assert rel(res[True][1], x64.grad) <= max(1.3 * rel(res[False][1], x64.grad), 2e-2)
        for n, p in ref.named_parameters():
            assert rel(res[True][2][n], p.grad) <= max(1.3 * rel(res[False][2][n], p.grad), 2e-2), n


# ---- the ReLU backward inside the consumer's input gradient (mdetr_tgemm_masked, linear.ReluToken) ---------------------------------
@pytest.mark.parametrize("T,K,N,with_res", [(61440, 128, 512, True), (15360, 256, 1024, True), (3840, 512, 2048, True),
                                            (61440, 512, 128, False), (15360, 1024, 256, False), (4403, 72, 264, True)])
def test_masked_input_gradient_equals_the_product_followed_by_threshold_backward(T, K, N, with_res):
    from monodetr_amd import tgemm_ext
    g = torch.Generator(device="cuda").manual_seed(T + K + N)
    dy = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(K, N, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    x = torch.randn(T, N, device="cuda", generator=g).clamp(min=0).to(torch.bfloat16)         # a ReLU output
    r = torch.randn(T, N, device="cuda", generator=g).to(torch.bfloat16) if with_res else None
    assert tgemm_ext.masked_supported(dy, w, x, r)
    want = torch.ops.aten.threshold_backward(tgemm_ext.tgemm(dy, w, None, r, nn=True), x, 0.0)
    got = tgemm_ext.tgemm_masked(dy, w, x, r)
    assert torch.equal(got, want)                                        # one rounding of the same fp32 sum, then the same zeros
    ref = dy.double() @ w.double() + (r.double() if with_res else 0.0)
    mag = dy.double().abs() @ w.double().abs() + (r.double().abs() if with_res else 0.0)
    keep = x > 0
    from gemm_bounds import assert_product_close
    assert_product_close(got, torch.where(keep, ref, torch.zeros_like(ref)), mag, K)


def test_bottleneck_stage_with_premasked_relu_backward_is_bit_identical(monkeypatch):
    """A ResNet stage at layer3's shape on the GPU kernels (tgemm, conv3x3, twgrad, conv_wgrad) with and without MDETR_RELU_PREMASK:
    the masks moved into the consumers' input-gradient products change no bit of any output or gradient."""
    from monodetr_amd import conv3x3_ext, conv_wgrad_ext, tgemm_ext
    from monodetr_amd.monodetr import backbone, linear
    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    for mod, on in ((conv3x3_ext, True), (conv_wgrad_ext, True)):
        monkeypatch.setattr(mod, "ENABLED", on)
    monkeypatch.setattr(linear, "_TGEMM", True)
    down = torch.nn.Sequential(torch.nn.Conv2d(512, 1024, 1, 1, bias=False), backbone.FrozenBatchNorm2d(1024))
    blocks = [backbone.Bottleneck(512, 256, 1, down), backbone.Bottleneck(1024, 256), backbone.Bottleneck(1024, 256)]
    for blk in blocks[:-1]:
        blk.__dict__["feeds_next_block"] = True
    stage = torch.nn.Sequential(*blocks).to(dev).to(memory_format=torch.channels_last)
    for m in stage.modules():
        if isinstance(m, backbone.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    x = (torch.randn(8, 512, 24, 80, device=dev) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    calls = []
    real = tgemm_ext.tgemm_masked
    monkeypatch.setattr(tgemm_ext, "tgemm_masked", lambda a, w, m, r=None: (calls.append(r is not None), real(a, w, m, r))[1])
    proj = torch.linspace(-1, 1, 8 * 1024 * 24 * 80, device=dev).view(8, 24, 80, 1024).permute(0, 3, 1, 2)
    res = {}
    for on in (False, True):
        monkeypatch.setattr(linear, "_PREMASK", on)
        for p in stage.parameters():
            p.grad = None
        x.grad = None
        calls.clear()
        pairs = []
        for blk in blocks:
            pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)] + ([(blk.downsample[0], blk.downsample[1])] if blk.downsample is not None else [])
        backbone.prefold(pairs, torch.bfloat16)
        y = stage(x)
        (y.float() * proj).sum().backward()
        res[on] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in stage.named_parameters()}, sorted(calls))
    assert res[False][3] == [] and res[True][3] == [False, False, False, True, True], res[True][3]
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for n, gr in res[False][2].items():
        assert torch.equal(res[True][2][n], gr), n
