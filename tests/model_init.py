"""Shared by the golden generator and the model tests: config loading, a construction-order
independent parameter initialiser, and the synthetic KITTI-shaped batch."""
import hashlib
import math
import os

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))

# configs/monodetr.yaml `model:` section of the reference (configs/monodetr.yaml:29-90), restated so
# tests do not read /root/reference; dropout 0 makes train mode deterministic.
MODEL_CFG = dict(
    num_classes=3, return_intermediate_dec=True, device='cpu', backbone='resnet50', train_backbone=True,
    num_feature_levels=4, dilation=False, position_embedding='sine', masks=False, mode='LID',
    num_depth_bins=80, depth_min=1e-3, depth_max=60.0, with_box_refine=True, two_stage=False,
    use_dab=False, use_dn=False, two_stage_dino=False, init_box=False, enc_layers=3, dec_layers=3,
    hidden_dim=256, dim_feedforward=256, dropout=0.0, nheads=8, num_queries=50, enc_n_points=4,
    dec_n_points=4, scalar=5, label_noise_scale=0.2, box_noise_scale=0.4, num_patterns=0, aux_loss=True,
    cls_loss_coef=2, focal_alpha=0.25, bbox_loss_coef=5, giou_loss_coef=2, dim_loss_coef=1,
    angle_loss_coef=1, depth_loss_coef=1, depth_map_loss_coef=1, set_cost_class=2, set_cost_bbox=5,
    set_cost_giou=2, set_cost_3dcenter=10)
MODEL_CFG['3dcenter_loss_coef'] = 10


def load_cfg(path=None, dropout=0.0, device='cpu'):
    if path is None:
        cfg = dict(MODEL_CFG)
    else:
        cfg = dict(yaml.load(open(path), Loader=yaml.Loader)['model'])
    cfg['dropout'] = dropout
    cfg['device'] = device
    return cfg


def _seed(name):
    return int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")


@torch.no_grad()
def name_seeded_init_(model):
    """Fill every parameter / buffer from a generator seeded by its NAME, so two implementations with
    the same state_dict surface get identical weights whatever their construction order."""
    for name, t in sorted(model.state_dict().items()):
        if not t.is_floating_point():
            continue
        g = torch.Generator().manual_seed(_seed(name))
        if name.endswith("running_var"):
            t.copy_(1 + 0.2 * torch.rand(t.shape, generator=g))
        elif name.endswith("running_mean"):
            t.copy_(0.1 * torch.randn(t.shape, generator=g))
        elif ".bn" in name or "downsample.1" in name:                      # frozen BN weight / bias
            t.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * torch.randn(t.shape, generator=g))
        elif name.endswith("depth_bin_values"):
            continue                                                        # derived constant
        elif "norm" in name or ".1.weight" in name and t.dim() == 1 or ".1.bias" in name and t.dim() == 1:
            t.copy_((1.0 if name.endswith("weight") else 0.0) + 0.05 * torch.randn(t.shape, generator=g))
        elif t.dim() > 1:
            fan_in = t[0].numel()
            t.copy_(torch.randn(t.shape, generator=g) / math.sqrt(fan_in))
        elif "sampling_offsets.bias" in name:
            t.copy_(torch.randn(t.shape, generator=g) * 1.5)
        else:
            t.copy_(0.02 * torch.randn(t.shape, generator=g))
    return model


def synthetic_batch(B, H=384, W=1280, seed=0, device="cpu", max_objs=8):
    """KITTI-shaped synthetic inputs and targets: ImageNet-normalised-like images, P2
    calibration, 1..max_objs cars per image with consistent 2D / 3D-centre boxes."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g)
    P2 = torch.tensor([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]])
    calibs = P2[None].repeat(B, 1, 1)
    img_sizes = torch.tensor([[1242, 375]] * B)
    targets = []
    for _ in range(B):
        K = int(torch.randint(1, max_objs + 1, (1,), generator=g))
        c = torch.rand(K, 2, generator=g) * 0.6 + 0.2
        lr = torch.rand(K, 2, generator=g) * 0.08 + 0.02
        tb = torch.rand(K, 2, generator=g) * 0.06 + 0.02
        boxes_3d = torch.cat([c, lr, tb], 1)                                     # cx, cy, l, r, t, b
        x0, x1 = c[:, 0] - lr[:, 0], c[:, 0] + lr[:, 1]
        y0, y1 = c[:, 1] - tb[:, 0], c[:, 1] + tb[:, 1]
        boxes = torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], 1)  # cxcywh
        targets.append(dict(
            labels=torch.ones(K, dtype=torch.int8), boxes=boxes, boxes_3d=boxes_3d,
            calibs=P2[None].repeat(K, 1, 1), depth=torch.rand(K, 1, generator=g) * 55 + 5,
            size_3d=torch.rand(K, 3, generator=g) * 3 + 1,
            heading_bin=torch.randint(0, 12, (K, 1), generator=g),
            heading_res=(torch.rand(K, 1, generator=g) - 0.5) * (math.pi / 6)))
    dev = torch.device(device)
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    return images.to(dev), calibs.to(dev), img_sizes.to(dev), targets


def disable_dropout_(model):
    """Zero every dropout probability (nn.Dropout modules and attention-internal dropout), including
    the depth encoder's hard-coded 0.1 (reference depth_predictor.py:48), so train mode is deterministic."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    return model


def grad_fingerprint(model):
    """{param name: (||grad||, <grad, r_name>)} with r_name a name-seeded Gaussian direction: two
    scalars that pin every parameter's gradient."""
    fp = {}
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = torch.Generator().manual_seed(_seed("dir:" + name))
        r = torch.randn(p.shape, generator=g, dtype=torch.float64)
        gr = p.grad.detach().double().cpu()
        fp[name] = (float(gr.norm()), float((gr * r).sum()))
    return fp
