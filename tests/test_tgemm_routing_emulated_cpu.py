"""MDETR_TGEMM routing (monodetr/linear.py, monodetr/backbone.py) with csrc/tgemm.hip running on the CPU shim: every token-wise
product of a module goes through the kernel (forward, input gradient, fused tails) and the module's outputs and gradients stay
those of the default route on the same bf16 parameters."""
import pytest
import torch
import torch.nn.functional as F

import native_emul


@pytest.fixture
def tgemm_on(monkeypatch):
    from monodetr_amd import bias_act_ext, small_wgrad_ext, tgemm_ext
    from monodetr_amd.monodetr import linear
    L = native_emul.lib()
    monkeypatch.setattr(tgemm_ext, "_backend", L)
    monkeypatch.setattr(bias_act_ext, "_backend", L)
    monkeypatch.setattr(small_wgrad_ext, "ENABLED", False)            # (weight gradients through the plain products here)
    monkeypatch.setattr(linear, "_TGEMM", True)
    calls = []
    real = tgemm_ext.tgemm
    monkeypatch.setattr(tgemm_ext, "tgemm", lambda *a, **k: (calls.append((tuple(a[0].shape), k.get("nn", False), a[3] is not None if len(a) > 3 else False,
                                                                          k.get("relu", False), k.get("dropout_p", 0.0))), real(*a, **k))[1])
    return calls


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


def test_token_linear_forms_take_the_kernel_and_match_autograd(tgemm_on):
    from monodetr_amd.monodetr import linear
    g = torch.Generator().manual_seed(3)
    T, K, N = 4200, 64, 72
    x = (torch.randn(3, T // 3, K, generator=g) * 0.5).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(N, generator=g).to(torch.bfloat16).requires_grad_(True)
    proj = torch.randn(3, T // 3, N, generator=g).to(torch.bfloat16)

    def ref(relu, skip):
        x2, w2, b2 = (t.detach().float().requires_grad_(True) for t in (x, w, b))
        y = F.linear(x2, w2, b2)
        y = F.relu(y) if relu else y
        ((y * proj.float()).sum() + ((x2 * 2.0).sum() if skip else 0.0)).backward()
        return y, x2.grad, w2.grad, b2.grad

    for relu in (False, True):
        for skip in (False, True):
            for t in (x, w, b):
                t.grad = None
            tgemm_on.clear()
            if skip:
                y, xs = linear.token_linear_skip(x, w, b, relu=relu)
                ((y * proj).float().sum() + (xs.float() * 2.0).sum()).backward()
            else:
                y = linear.token_linear(x, w, b, relu=relu)
                (y * proj).float().sum().backward()
            want = ref(relu, skip)
            assert _rel(y, want[0]) <= 4e-3
            assert _rel(x.grad, want[1]) <= 6e-3 and _rel(w.grad, want[2]) <= 1e-2 and _rel(b.grad, want[3]) <= 1e-2
            kinds = [(c[1], c[2]) for c in tgemm_on]
            assert (False, False) in kinds                             # forward through the kernel ...
            assert (True, skip) in kinds                               # ... and the input gradient, the residual gradient summed inside


def test_ffn_first_half_runs_relu_and_dropout_in_the_epilogue(tgemm_on):
    from monodetr_amd.monodetr import linear
    torch.manual_seed(0)
    lin = linear.Linear(64, 72).to(torch.bfloat16)
    drop = torch.nn.Dropout(0.25)
    x = (torch.randn(2, 2100, 64) * 0.5).to(torch.bfloat16).requires_grad_(True)
    for skip in (False, True):
        tgemm_on.clear()
        out = linear.ffn_hidden(x, lin, drop, skip=skip)
        h = out[0] if skip else out
        assert [c for c in tgemm_on if c[3] and c[4] == 0.25], tgemm_on          # one launch: bias + ReLU + Dropout
        pre = F.linear(x.float(), lin.weight.float(), lin.bias.float())
        kept = h != 0
        # kept elements are relu(pre) / 0.75; dropped or negative ones are zero
        assert _rel(h[kept], (pre.clamp(min=0) / 0.75)[kept]) <= 5e-3
        frac = kept.float().mean().item()
        assert 0.3 < frac < 0.45                                       # ~ half positive x three quarters kept
        gsum = torch.autograd.grad(h.float().sum(), x, retain_graph=False)[0]
        want = (kept.float() / 0.75) @ lin.weight.float()
        assert _rel(gsum, want) <= 1e-2


def test_bottleneck_with_fused_tails_matches_the_default_block(tgemm_on, monkeypatch):
    """conv1 (shift + ReLU in the epilogue, the identity's gradient summed inside its input-gradient product) and conv3 (shift +
    identity + ReLU in the epilogue) of a bottleneck against the default route, with and without a projection shortcut."""
    from monodetr_amd.monodetr import backbone, linear
    torch.manual_seed(1)
    for down in (False, True):
        inpl, planes = (64, 16) if not down else (32, 16)
        ds = torch.nn.Sequential(torch.nn.Conv2d(inpl, planes * 4, 1, 1, bias=False), backbone.FrozenBatchNorm2d(planes * 4)) if down else None
        blk = backbone.Bottleneck(inpl, planes, 1, ds)
        for m in blk.modules():
            if isinstance(m, backbone.FrozenBatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
        x = (torch.randn(2, inpl, 48, 48) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        res = {}
        for on in (False, True):
            monkeypatch.setattr(linear, "_TGEMM", on)
            for p in blk.parameters():
                p.grad = None
            x.grad = None
            tgemm_on.clear()
            pairs = [(c, b) for c, b in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)) + (((ds[0], ds[1]),) if down else ())]
            backbone.prefold(pairs, torch.bfloat16)
            y = blk(x)
            (y.float() * 0.01).sum().backward()
            res[on] = (y.detach().float(), x.grad.float().clone(), {n: p.grad.float().clone() for n, p in blk.named_parameters()})
            if on:
                assert [c for c in tgemm_on if c[2] and c[3] and not c[1]], tgemm_on      # conv3: residual + ReLU epilogue
        # both routes against the block in fp32 on the same (bf16-valued) parameters: the fused route must be as close as the default
        import copy
        ref = copy.deepcopy(blk).float()
        x32 = x.detach().float().requires_grad_(True)
        monkeypatch.setattr(linear, "_TGEMM", False)
        y32 = ref(x32)
        (y32 * 0.01).sum().backward()
        want = (y32.detach(), x32.grad, {n: p.grad for n, p in ref.named_parameters()})
        for on in (False, True):
            assert _rel(res[on][0], want[0]) <= 8e-3
        e_def, e_fused = _rel(res[False][1], want[1]), _rel(res[True][1], want[1])
        assert e_fused <= max(1.3 * e_def, 3e-2), (e_def, e_fused)
        for n, gd in want[2].items():
            e_def, e_fused = _rel(res[False][2][n], gd), _rel(res[True][2][n], gd)
            assert e_fused <= max(1.3 * e_def, 3e-2), (n, e_def, e_fused)


def test_stage_with_premasked_relu_backward_is_bit_identical(tgemm_on, monkeypatch):
    """linear.ReluToken: with MDETR_RELU_PREMASK the ReLU masks between conv2 -> conv3 and between consecutive blocks of a stage are
    applied inside the consumers' input-gradient products (mdetr_tgemm_masked) and the producers skip their passes.  Same bits:
    one rounding of the same fp32 sum, then the same zeros."""
    from monodetr_amd import conv3x3_ext, conv_wgrad_ext, tgemm_ext
    from monodetr_amd.monodetr import backbone, linear
    L = native_emul.lib()
    monkeypatch.setattr(conv3x3_ext, "_backend", L)
    monkeypatch.setattr(conv3x3_ext, "ENABLED", True)
    monkeypatch.setattr(conv_wgrad_ext, "ENABLED", False)
    torch.manual_seed(2)
    down = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 1, 1, bias=False), backbone.FrozenBatchNorm2d(256))
    blocks = [backbone.Bottleneck(128, 64, 1, down), backbone.Bottleneck(256, 64), backbone.Bottleneck(256, 64)]
    for blk in blocks[:-1]:
        blk.__dict__["feeds_next_block"] = True
    stage = torch.nn.Sequential(*blocks).to(memory_format=torch.channels_last)
    for m in stage.modules():
        if isinstance(m, backbone.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    x = (torch.randn(2, 128, 48, 48) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    masked_calls = []
    real = tgemm_ext.tgemm_masked
    monkeypatch.setattr(tgemm_ext, "tgemm_masked", lambda a, w, m, r=None: (masked_calls.append((tuple(a.shape), r is not None)), real(a, w, m, r))[1])
    res = {}
    for on in (False, True):
        monkeypatch.setattr(linear, "_PREMASK", on)
        for p in stage.parameters():
            p.grad = None
        x.grad = None
        masked_calls.clear()
        pairs = []
        for blk in blocks:
            pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)] + ([(blk.downsample[0], blk.downsample[1])] if blk.downsample is not None else [])
        backbone.prefold(pairs, torch.bfloat16)
        y = stage(x)
        assert (getattr(y, "_mdetr_relu_token", None) is None)          # the stage's output leaves the module: no token on it
        (y.float() * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
        res[on] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in stage.named_parameters()}, list(masked_calls))
    assert res[False][3] == []
    # three conv3 input gradients masked by conv2's output (no residual), two conv1 input gradients of the identity blocks masked by
    # the previous block's output with the identity's gradient summed inside
    assert sorted(r for _, r in res[True][3]) == [False, False, False, True, True], res[True][3]
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for n, g in res[False][2].items():
        assert torch.equal(res[True][2][n], g), n


def test_no_relu_token_is_handed_out_when_a_hook_can_see_the_tensor(tgemm_on, monkeypatch):
    """linear.ReluToken: a forward hook on a block is a second party that may consume its output -- the block then keeps its own mask
    pass (no token rides on its result) and the next block does not premask."""
    from monodetr_amd import tgemm_ext
    from monodetr_amd.monodetr import backbone, linear
    monkeypatch.setattr(linear, "_PREMASK", True)
    torch.manual_seed(5)
    blocks = [backbone.Bottleneck(64, 16), backbone.Bottleneck(64, 16)]
    blocks[0].__dict__["feeds_next_block"] = True
    seen = []
    blocks[0].register_forward_hook(lambda m, i, o: seen.append(getattr(o, "_mdetr_relu_token", None)))
    stage = torch.nn.Sequential(*blocks).to(memory_format=torch.channels_last)
    masked = []
    real = tgemm_ext.tgemm_masked
    monkeypatch.setattr(tgemm_ext, "tgemm_masked", lambda a, w, m, r=None: (masked.append(1), real(a, w, m, r))[1])
    x = (torch.randn(2, 64, 48, 48) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    pairs = []
    for blk in blocks:
        pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)]
    backbone.prefold(pairs, torch.bfloat16)
    y = stage(x)
    y.float().sum().backward()
    assert seen == [None] and masked == []                            # block 0: hooked -> no token; block 1's conv2 is the library's here -> nothing premasked
    assert x.grad is not None and torch.isfinite(x.grad.float()).all()
