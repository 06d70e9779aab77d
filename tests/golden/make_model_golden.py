"""Generate tests/golden/model_*.npz by running the REFERENCE model code (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_model_golden.py

The reference package cannot be imported as-is here: torchvision is not
installed, its torch-version gates mis-fire on torch 2.x, the native extension is CUDA-only and the
criterion hard-codes .cuda().  This script supplies, IN ITS OWN PROCESS ONLY, the missing pieces:
  * sys.modules['torchvision'] stub: ``models.resnet50`` builds this repo's torchvision-free ResNet
    (so the backbone itself is NOT independently validated by these vectors -- everything after it
    is: input_proj, depth predictor, encoder, decoder, heads, criterion, matcher);
    ``IntermediateLayerGetter`` and ``box_area`` are restated in a few lines;
  * sys.modules['MultiScaleDeformableAttention'] = the CPU oracle (pinned to the reference's own
    pure-PyTorch op by tests/test_oracle_golden.py);
  * attribute shims for ``torch.nn.modules.linear._LinearWithBias`` / ``torch._overrides`` (needed by
    ops/modules/ms_deform_attn.py:34-37,55-58), ``torch.cuda.current_device`` (ddn_loss.py:32),
    ``Tensor.cuda`` (monodetr.py:439) and ``torch.tensor(device='cuda')`` (monodetr.py:452).
Then the unmodified reference classes are built from configs/monodetr.yaml (dropout 0 for
determinism), their parameters are filled by the name-seeded initialiser shared with the tests
(tests/model_init.py), and outputs / losses / matcher indices are recorded for one KITTI-sized
batch (2 x 3 x 384 x 1280) in train mode (550 queries) and eval mode (50 queries).
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"
sys.dont_write_bytecode = True


def install_shims():
    from collections import OrderedDict
    from torch import nn
    from monodetr_amd.monodetr.backbone import ResNetBody
    from oracle import msda_oracle

    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.9.0"
    models = types.ModuleType("torchvision.models")
    _utils = types.ModuleType("torchvision.models._utils")
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")
    misc = types.ModuleType("torchvision.ops.misc")

    def resnet50(replace_stride_with_dilation=None, pretrained=False, norm_layer=None):
        body = ResNetBody("resnet50", {}, dilation=bool(replace_stride_with_dilation and replace_stride_with_dilation[2]),
                          norm_layer=norm_layer)
        body.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        body.fc = nn.Linear(2048, 1000)
        return body

    class IntermediateLayerGetter(nn.ModuleDict):
        def __init__(self, model, return_layers):
            layers = OrderedDict()
            remaining = dict(return_layers)
            for name, module in model.named_children():
                layers[name] = module
                remaining.pop(name, None)
                if not remaining:
                    break
            super().__init__(layers)
            self.return_layers = dict(return_layers)

        def forward(self, x):
            out = OrderedDict()
            for name, module in self.items():
                x = module(x)
                if name in self.return_layers:
                    out[self.return_layers[name]] = x
            return out

    models.resnet50 = resnet50
    _utils.IntermediateLayerGetter = IntermediateLayerGetter
    boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    tv.models, models._utils, tv.ops, ops.boxes, ops.misc = models, _utils, ops, boxes, misc
    for name, mod in (("torchvision", tv), ("torchvision.models", models), ("torchvision.models._utils", _utils),
                      ("torchvision.ops", ops), ("torchvision.ops.boxes", boxes), ("torchvision.ops.misc", misc)):
        sys.modules[name] = mod

    msda_oracle.build()
    ext = types.ModuleType("MultiScaleDeformableAttention")
    ext.ms_deform_attn_forward = msda_oracle.OracleMSDA.ms_deform_attn_forward
    ext.ms_deform_attn_backward = msda_oracle.OracleMSDA.ms_deform_attn_backward
    sys.modules["MultiScaleDeformableAttention"] = ext

    import torch.nn.modules.linear as lin
    lin._LinearWithBias = lin.NonDynamicallyQuantizableLinear
    ov = types.ModuleType("torch._overrides")
    ov.has_torch_function, ov.handle_torch_function = torch.overrides.has_torch_function, torch.overrides.handle_torch_function
    sys.modules["torch._overrides"] = ov
    torch.cuda.current_device = lambda: 0
    torch.Tensor.cuda = lambda self, *a, **k: self
    _tensor = torch.tensor
    torch.tensor = lambda *a, **k: _tensor(*a, **{kk: vv for kk, vv in k.items() if not (kk == "device" and str(vv).startswith("cuda"))})


def main():
    install_shims()
    sys.path.insert(0, REF)
    from lib.models.monodetr import build_monodetr as ref_build          # the reference's builder
    from model_init import disable_dropout_, load_cfg, name_seeded_init_, synthetic_batch

    cfg = load_cfg(os.path.join(REF, "configs", "monodetr.yaml"))
    torch.manual_seed(0)
    model, criterion = ref_build(cfg)
    name_seeded_init_(model)
    disable_dropout_(model)
    # the reference body carries avgpool/fc (dropped by IntermediateLayerGetter) - nothing to do

    keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    images, calibs, img_sizes, targets = synthetic_batch(2, 384, 1280, seed=7)
    rec = {}
    for mode in ("train", "eval"):
        model.train(mode == "train")
        criterion.train(mode == "train")
        with torch.no_grad():
            out = model(images, calibs, targets, img_sizes)
            losses = criterion(out, targets)
            group_num = 11 if mode == "train" else 1
            idx = criterion.matcher({k: v for k, v in out.items() if k != "aux_outputs"}, targets, group_num=group_num)
        for k, v in out.items():
            if k != "aux_outputs":
                rec[f"{mode}/{k}"] = v.numpy()
        for i, aux in enumerate(out["aux_outputs"]):
            for k, v in aux.items():
                rec[f"{mode}/aux{i}/{k}"] = v.numpy()
        for k, v in losses.items():
            rec[f"{mode}/loss/{k}"] = np.asarray(float(v))
        for b, (i, j) in enumerate(idx):
            rec[f"{mode}/match/{b}/src"], rec[f"{mode}/match/{b}/tgt"] = i.numpy(), j.numpy()
        with torch.no_grad():
            for a, aux in enumerate(out["aux_outputs"]):
                for b, (i, j) in enumerate(criterion.matcher(aux, targets, group_num=group_num)):
                    rec[f"{mode}/match_aux{a}/{b}/src"], rec[f"{mode}/match_aux{a}/{b}/tgt"] = i.numpy(), j.numpy()
        print(mode, {k: round(float(v), 5) for k, v in losses.items() if not k[-1].isdigit()})

    # gradients of the weighted loss w.r.t. a few parameters (train mode), for the backward path
    model.train(); criterion.train()
    out = model(images, calibs, targets, img_sizes)
    losses = criterion(out, targets)
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    total.backward()
    rec["train/total_loss"] = np.asarray(float(total))
    for name in ("depthaware_transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
                 "depthaware_transformer.decoder.layers.2.cross_attn.value_proj.weight",
                 "depthaware_transformer.decoder.layers.0.cross_attn_depth.in_proj_weight",
                 "depth_predictor.depth_classifier.weight", "input_proj.0.0.weight", "query_embed.weight",
                 "backbone.0.body.layer4.2.conv3.weight", "class_embed.2.bias"):
        g = dict(model.named_parameters())[name].grad
        rec["grad/" + name] = g.numpy()
    # float64 pass: structural parity without fp32 noise.  Every parameter's gradient is recorded
    # through two scalars (its norm and its projection on a name-seeded random direction).
    from model_init import grad_fingerprint
    model.double(); model.zero_grad(set_to_none=True)
    out = model(images.double(), calibs.double(), targets64(targets), img_sizes)
    losses = criterion(out, targets64(targets))
    total = sum(losses[k] * criterion.weight_dict[k] for k in losses if k in criterion.weight_dict)
    total.backward()
    rec["f64/total_loss"] = np.asarray(float(total.detach()))
    for k in ("pred_logits", "pred_boxes", "pred_depth", "pred_3d_dim", "pred_angle"):
        rec["f64/" + k] = out[k].detach().numpy()
    for k, v in losses.items():
        rec["f64/loss/" + k] = np.asarray(float(v))
    with torch.no_grad():
        layers = [{k: v for k, v in out.items() if k != "aux_outputs"}] + list(out["aux_outputs"])
        for li, layer in enumerate(layers):
            for b, (i, j) in enumerate(criterion.matcher(layer, targets64(targets), group_num=11)):
                rec[f"f64/match{li}/{b}/src"], rec[f"f64/match{li}/{b}/tgt"] = i.numpy(), j.numpy()
    fp = grad_fingerprint(model)
    rec["f64/grad_names"] = np.array(sorted(fp))
    rec["f64/grad_fp"] = np.array([fp[k] for k in sorted(fp)])
    np.savez_compressed(os.path.join(HERE, "model_kitti_b2.npz"), **rec)
    with open(os.path.join(HERE, "model_state_dict_keys.txt"), "w") as f:
        for k in sorted(keys):
            f.write("%s %s\n" % (k, "x".join(map(str, keys[k]))))
    print("saved", len(rec), "arrays;", len(keys), "state_dict entries; f64 total", float(total))


def targets64(targets):
    return [{k: (v.double() if v.is_floating_point() else v) for k, v in t.items()} for t in targets]


if __name__ == "__main__":
    main()
