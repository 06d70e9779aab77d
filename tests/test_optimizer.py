"""AdamW mirror vs six recorded steps of the reference's AdamW (tests/golden/optimizer_adamw.npz)."""
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
