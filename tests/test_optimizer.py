"""AdamW mirror vs six recorded steps of the reference's AdamW (tests/golden/optimizer_adamw.npz)."""
import pytest
import torch

from conftest import load_golden
from optimizer_problem import make_grads, make_model


def test_adamw_matches_reference_steps():
    from monodetr_amd.helpers.optimizer_helper import AdamW, build_optimizer
    g = load_golden("optimizer_adamw")
    model = make_model()
    opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4}, model)
    assert isinstance(opt, AdamW)
    # biases: no decay; weights: decay (reference :8-16)
    assert opt.param_groups[0]['weight_decay'] == 0 and opt.param_groups[1]['weight_decay'] == 1e-4
    assert len(opt.param_groups[0]['params']) == 3 and len(opt.param_groups[1]['params']) == 4
    for step in range(6):
        make_grads(model, step)
        opt.step()
        if step in (0, 5):
            for n, p in model.named_parameters():
                ref = g["step%d/%s" % (step, n)]
                assert (p.detach() - ref).abs().max() < 1e-14, (step, n)


def test_state_dict_roundtrip():
    from monodetr_amd.helpers.optimizer_helper import build_optimizer
    model = make_model()
    opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4}, model)
    make_grads(model, 3)
    opt.step()
    sd = opt.state_dict()
    opt2 = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4}, make_model())
    opt2.load_state_dict(sd)
    assert opt2.state_dict()['state'][0]['step'] == 1


def test_bf16_parameters_use_fp32_master_weights():
    """A bf16 parameter follows the fp32 trajectory (rounded), instead of stalling on bf16 rounding."""
    from monodetr_amd.helpers.optimizer_helper import AdamW
    torch.manual_seed(0)
    w32 = torch.nn.Parameter(torch.randn(64, 64))
    w16 = torch.nn.Parameter(w32.detach().to(torch.bfloat16))
    o32, o16 = AdamW([w32], lr=1e-4, weight_decay=1e-4), AdamW([w16], lr=1e-4, weight_decay=1e-4)
    start = w32.detach().clone()
    start16 = w16.detach().float().clone()
    for step in range(50):
        g = torch.randn(64, 64, generator=torch.Generator().manual_seed(step)) + 0.5
        w32.grad, w16.grad = g.clone(), g.to(torch.bfloat16)
        o32.step(); o16.step()
    master = o16.state[w16]['master']
    assert master.dtype == torch.float32 and w16.dtype == torch.bfloat16
    assert torch.equal(w16.detach(), master.to(torch.bfloat16))
    # the master tracks the fp32 run closely (gradients differ only by their bf16 rounding)
    assert ((master - start16) - (w32.detach() - start)).abs().max() < 2e-2 * (w32.detach() - start).abs().max()
    assert (w32.detach() - start).abs().max() > 1e-3


def test_capturable_mode_matches_host_step_size():
    """capturable=True (device-resident step count and step size) follows the same trajectory."""
    from monodetr_amd.helpers.optimizer_helper import build_optimizer
    g = load_golden("optimizer_adamw")
    model = make_model()
    opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4, 'capturable': True}, model)
    for step in range(6):
        make_grads(model, step)
        opt.step()
    for n, p in model.named_parameters():
        ref = g["step5/%s" % n]
        # the step size is rounded to fp32 on the device: relative 6e-8 of an update of order lr
        assert (p.detach() - ref).abs().max() < 1e-9, n
    assert max(float(t) for t in opt.param_groups[0]['step_dev'].values()) == 6.0


def _fused(params_or_model, backend="host", **kw):
    """FusedAdamW wired to a CPU stand-in for the library (tests/backends.py: the host build of the kernel
    arithmetic, or the real adamw.hip on the HIP-on-CPU shim) so that the flat layout logic and the update
    run on CPU tensors."""
    import backends
    from monodetr_amd.helpers.optimizer_helper import FusedAdamW, build_optimizer
    if isinstance(params_or_model, torch.nn.Module):
        opt = build_optimizer({'type': 'adamw', 'lr': 2e-4, 'weight_decay': 1e-4, 'fused': True, **kw}, params_or_model)
    else:
        opt = FusedAdamW(params_or_model, **kw)
    assert isinstance(opt, FusedAdamW)
    opt._lib, opt._allow_cpu = backends.get(backend), True
    return opt


@pytest.mark.parametrize("backend", ["host", "emul"])
def test_fused_adamw_matches_reference_steps_fp32(backend):
    """One launch per (group, dtype) over flat buffers == six recorded steps of the reference's AdamW
    (fp32 arithmetic against the fp64 recording: 1e-6 relative)."""
    g = load_golden("optimizer_adamw")
    shadow, model = make_model(), make_model().float()          # the recorded gradients are float64 draws
    opt = _fused(model, backend)
    for step in range(6):
        make_grads(shadow, step)
        for ps, p in zip(shadow.parameters(), model.parameters()):
            p.grad = None if ps.grad is None else ps.grad.float()
        opt.step()
    for n, p in model.named_parameters():
        ref = torch.as_tensor(g["step5/%s" % n]).float()
        assert (p.detach() - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item()), n
    # the parameters are now views of one flat buffer per (group, dtype, cohort): 2 groups x {from step 0, 'late' from step 2}
    flat = opt._flat[1]
    assert len(flat) == 4 and all(b["param"].data_ptr() <= p.data_ptr() < b['param'].data_ptr() + 4 * b['n']
                                  for b in flat for p in b['params'])
    sd = opt.state_dict()
    assert sd['state'][0]['exp_avg'].shape == opt.param_groups[0]['params'][0].shape


@pytest.mark.parametrize("backend", ["host", "emul"])
def test_fused_adamw_equals_foreach_adamw_on_mixed_layouts_and_dtypes(backend):
    """Same trajectory as the multi-tensor AdamW for fp32 + bf16 parameters, a channels_last 4-d weight,
    odd sizes (padding) and a parameter that never gets a gradient; then a state_dict round trip."""
    from monodetr_amd.helpers.optimizer_helper import AdamW
    torch.manual_seed(5)

    def make():
        torch.manual_seed(5)
        w4 = torch.nn.Parameter(torch.randn(6, 5, 3, 3).contiguous(memory_format=torch.channels_last))
        b1 = torch.nn.Parameter(torch.randn(7))
        wl = torch.nn.Parameter(torch.randn(33, 17).to(torch.bfloat16))
        bl = torch.nn.Parameter(torch.randn(33).to(torch.bfloat16))
        unused = torch.nn.Parameter(torch.randn(4))
        groups = [{'params': [b1, bl, unused], 'weight_decay': 0}, {'params': [w4, wl], 'weight_decay': 1e-2}]
        return [w4, b1, wl, bl, unused], groups

    pa, ga = make()
    pb, gb = make()
    oa, ob = AdamW(ga, lr=1e-3), _fused(gb, backend, lr=1e-3)
    for step in range(5):
        gen = torch.Generator().manual_seed(100 + step)
        for x, y in zip(pa[:4], pb[:4]):
            g = torch.randn(x.shape, generator=gen)
            if x.dim() == 4:
                g = g.contiguous(memory_format=torch.channels_last)
            x.grad, y.grad = g.to(x.dtype), g.to(y.dtype).clone()
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert x.stride() == y.stride()
        tol = 1e-6 if x.dtype == torch.float32 else 1e-2
        assert (x.detach().float() - y.detach().float()).abs().max() <= tol * max(1.0, x.detach().float().abs().max().item())
    # fp32 master copies of the bf16 parameters agree much more closely than the rounded model copies
    assert (oa.state[pa[2]]['master'] - ob.state[pb[2]]['master']).abs().max() < 1e-5
    assert ob.state[pb[2]]['master'].dtype == torch.float32 and torch.equal(pb[2].detach(), ob.state[pb[2]]['master'].to(torch.bfloat16))
    assert pb[4].grad is None and 'exp_avg' not in ob.state[pb[4]]
    # state_dict round trip into a fresh fused optimizer continues the same trajectory
    pc, gc = make()
    oc = _fused(gc, backend, lr=1e-3)
    with torch.no_grad():
        for y, z in zip(pb, pc):
            z.copy_(y)
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))          # as read back from a checkpoint file
    gen = torch.Generator().manual_seed(999)
    for y, z in zip(pb[:4], pc[:4]):
        g = torch.randn(y.shape, generator=gen)
        if y.dim() == 4:
            g = g.contiguous(memory_format=torch.channels_last)
        y.grad, z.grad = g.to(y.dtype), g.to(z.dtype).clone()
    ob.step(); oc.step()
    for y, z in zip(pb[:4], pc[:4]):
        assert torch.equal(y.detach(), z.detach())


def test_replay_bookkeeping_and_checkpoint_sanitisation():
    """helpers/step_helper.TrainIteration replays a captured ``step()``: the device counters advance by themselves, the host
    counts (what a checkpoint saves) are brought up to date lazily; a device-resident learning rate is saved as a float and
    the device counters stay out of the checkpoint."""
    from monodetr_amd.helpers.optimizer_helper import AdamW
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(5, 3))
    b = torch.nn.Parameter(torch.randn(3))
    opt = AdamW([{'params': [b], 'weight_decay': 0}, {'params': [w], 'weight_decay': 1e-4}], lr=1e-3, capturable=True)
    for g in opt.param_groups:
        g['lr'] = torch.tensor(1e-3, dtype=torch.float64)
    ref_w, ref_b = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = AdamW([{'params': [ref_b], 'weight_decay': 0}, {'params': [ref_w], 'weight_decay': 1e-4}], lr=1e-3)
    for i in range(6):
        gw, gb = torch.randn(5, 3), torch.randn(3)
        w.grad, b.grad, ref_w.grad, ref_b.grad = gw.clone(), gb.clone(), gw.clone(), gb.clone()
        opt.step()
        ref.step()
        if i == 2:                         # a capture: the host bookkeeping of a step whose kernels do not run is taken back ...
            opt.uncount_step()
            opt.note_replay()              # ... and the replay that follows is counted lazily
    assert torch.allclose(w, ref_w, atol=1e-7) and torch.allclose(b, ref_b, atol=1e-7)
    opt.note_replay(); opt.note_replay()
    sd = opt.state_dict()
    assert {int(s['step']) for s in sd['state'].values()} == {8}
    assert all(isinstance(g['lr'], float) and 'step_dev' not in g for g in sd['param_groups'])
    assert all(torch.is_tensor(g['lr']) and 'step_dev' in g for g in opt.param_groups)      # the live groups keep theirs


def test_gathered_update_reads_gradient_slices_at_any_offset():
    """The gathered AdamW (csrc/adamw.hip on the CPU shim) with gradients that are slices of ONE exchange buffer at odd element
    offsets -- what `dist_helper`'s flat all-reduce leaves in `.grad` -- against the multi-tensor AdamW."""
    from monodetr_amd.helpers.optimizer_helper import AdamW

    def make():
        torch.manual_seed(11)
        ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 64, 3, 130, 7)] + [torch.nn.Parameter(torch.randn(9, 6).to(torch.bfloat16)),
                                                                                 torch.nn.Parameter(torch.randn(33).to(torch.bfloat16))]
        return ps, [{'params': ps, 'weight_decay': 1e-2}]

    pa, ga = make()
    pb, gb = make()
    oa, ob = AdamW(ga, lr=1e-3), _fused(gb, "emul", lr=1e-3)
    assert ob._gather
    for step in range(3):
        gen = torch.Generator().manual_seed(50 + step)
        flat32 = torch.randn(1 + sum(p.numel() for p in pb[:5]), generator=gen)
        flat16 = torch.randn(1 + sum(p.numel() for p in pb[5:]), generator=gen).to(torch.bfloat16)
        o32, o16 = 1, 1                                   # the first slice starts at element 1: no 16-byte alignment anywhere
        for x, y in zip(pa, pb):
            if y.dtype == torch.float32:
                y.grad = flat32[o32:o32 + y.numel()].view_as(y); o32 += y.numel()
            else:
                y.grad = flat16[o16:o16 + y.numel()].view_as(y); o16 += y.numel()
            x.grad = y.grad.clone()
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        tol = 1e-6 if x.dtype == torch.float32 else 1e-2
        assert (x.detach().float() - y.detach().float()).abs().max() <= tol * max(1.0, x.detach().float().abs().max().item())
