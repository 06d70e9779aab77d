"""MonoDETR model and criterion -- mirror of lib/models/monodetr/monodetr.py (``MonoDETR`` :28-293,
``SetCriterion`` :296-532, ``MLP`` :535-547, ``build`` :550-614).

Same constructor arguments, parameter names (state_dict surface), forward
signature ``model(images, calibs, targets, img_sizes, dn_args=None) -> dict`` with the same keys,
and ``criterion(outputs, targets, mask_dict=None) -> dict`` with the same loss names and values.

Differences underneath (results unchanged):
  * the backbone is the torchvision-free ResNet of backbone.py; masks known to be all-False are
    tagged so downstream code skips mask work without device syncs;
  * the criterion matches all decoder layers with one device->host copy (matcher.match_layers),
    concatenates the ground truth once per call instead of once per loss, builds the
    (image, query, target) index triple once per layer, and evaluates GIoU on matched pairs only
    (the reference builds an N x N matrix and takes its diagonal, :381-383);
  * no hard-coded ``.cuda()`` / ``device='cuda'`` (reference :439, :452) and the depth-map scale
    comes from the logits' shape instead of the literal [80, 24, 80, 24] (:452), so the criterion
    runs on any device and at any resolution;
  * ``num_boxes`` stays on the host when no process group exists (the reference round-trips it
    through the device and ``.item()``, :503-508).
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from ..losses.focal_loss import sigmoid_focal_loss
from ..utils import box_ops
from ..utils.misc import (NestedTensor, get_world_size, inverse_sigmoid,
                          is_dist_avail_and_initialized, mark_no_padding)
from . import _cut
from .backbone import build_backbone
from .depth_predictor import DepthPredictor
from .depth_predictor.ddn_loss import DDNLoss
from .depthaware_transformer import MLP, build_depthaware_transformer
from ..group_norm_ext import GroupNorm
from .linear import PointwiseConv2d
from .matcher import build_matcher


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class MonoDETR(nn.Module):
    """Monocular 3D object detector: ResNet features -> depth predictor -> depth-aware transformer
    -> per-layer heads (class, 3D-centre + 2D box sides, 3D size, heading, depth)."""

    def __init__(self, backbone, depthaware_transformer, depth_predictor, num_classes, num_queries,
                 num_feature_levels, aux_loss=True, with_box_refine=False, two_stage=False, init_box=False,
                 use_dab=False, group_num=11, two_stage_dino=False):
        super().__init__()
        if two_stage or use_dab or two_stage_dino:
            raise NotImplementedError("two_stage / use_dab / two_stage_dino are off in configs/monodetr.yaml and not built")
        self.num_queries = num_queries
        self.group_num = group_num
        self.depthaware_transformer = depthaware_transformer
        self.depth_predictor = depth_predictor
        hidden_dim = depthaware_transformer.d_model
        self.hidden_dim = hidden_dim
        self.num_feature_levels = num_feature_levels
        self.two_stage_dino, self.use_dab, self.two_stage = two_stage_dino, use_dab, two_stage
        self.label_enc = nn.Embedding(num_classes + 1, hidden_dim - 1)      # only the DN branch reads it

        class_embed = nn.Linear(hidden_dim, num_classes)
        class_embed.bias.data = torch.ones(num_classes) * -math.log((1 - 0.01) / 0.01)   # prior prob 0.01
        bbox_embed = MLP(hidden_dim, hidden_dim, 6, 3)
        dim_embed_3d = MLP(hidden_dim, hidden_dim, 3, 2)
        angle_embed = MLP(hidden_dim, hidden_dim, 24, 2)
        depth_embed = MLP(hidden_dim, hidden_dim, 2, 2)                     # depth and log-variance
        if init_box:
            nn.init.constant_(bbox_embed.layers[-1].weight.data, 0)
            nn.init.constant_(bbox_embed.layers[-1].bias.data, 0)

        # one (positional | content) embedding per query, group_num groups of num_queries (train)
        self.query_embed = nn.Embedding(num_queries * group_num, hidden_dim * 2)

        def proj(cin, k, stride):
            return nn.Sequential(PointwiseConv2d(cin, hidden_dim, kernel_size=k, stride=stride, padding=k // 2),
                                 GroupNorm(32, hidden_dim))
        if num_feature_levels > 1:
            projs = [proj(c, 1, 1) for c in backbone.num_channels]
            cin = backbone.num_channels[-1]
            for _ in range(num_feature_levels - len(backbone.strides)):
                projs.append(proj(cin, 3, 2))
                cin = hidden_dim
            self.input_proj = nn.ModuleList(projs)
        else:
            self.input_proj = nn.ModuleList([proj(backbone.num_channels[0], 1, 1)])
        for p in self.input_proj:
            nn.init.xavier_uniform_(p[0].weight, gain=1)
            nn.init.constant_(p[0].bias, 0)

        self.backbone = backbone
        self.aux_loss = aux_loss
        self.with_box_refine = with_box_refine
        self.num_classes = num_classes

        num_pred = depthaware_transformer.decoder.num_layers
        if with_box_refine:
            self.class_embed = _get_clones(class_embed, num_pred)
            self.bbox_embed = _get_clones(bbox_embed, num_pred)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            self.depthaware_transformer.decoder.bbox_embed = self.bbox_embed      # shared with the decoder
            self.dim_embed_3d = _get_clones(dim_embed_3d, num_pred)
            self.depthaware_transformer.decoder.dim_embed = self.dim_embed_3d
            self.angle_embed = _get_clones(angle_embed, num_pred)
            self.depth_embed = _get_clones(depth_embed, num_pred)
            self.fuse_heads = True          # the decoder evaluates all five heads of a level together (see forward)
        else:
            nn.init.constant_(bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([class_embed for _ in range(num_pred)])
            self.bbox_embed = nn.ModuleList([bbox_embed for _ in range(num_pred)])
            self.dim_embed_3d = nn.ModuleList([dim_embed_3d for _ in range(num_pred)])
            self.angle_embed = nn.ModuleList([angle_embed for _ in range(num_pred)])
            self.depth_embed = nn.ModuleList([depth_embed for _ in range(num_pred)])
            self.depthaware_transformer.decoder.bbox_embed = None
            self.fuse_heads = False

    def pyramid(self, images):
        """Backbone + input projections (monodetr.py:156-178 of the reference): (srcs, masks, pos), one entry per
        feature level."""
        features, pos = self.backbone(images)
        boundary = self.__dict__.get("_grad_boundary")
        if boundary is not None and torch.is_grad_enabled():
            # a two-part backward pass (helpers/step_helper.TrainIteration, N > 1 ranks): the pyramid levels are cut out of the
            # autograd graph here -- everything above runs on detached copies, whose .grad the second part (the backbone's
            # backward) starts from, while the gradients of the part above are already being exchanged
            cut = []
            for f in features:
                t = f.tensors
                if t.requires_grad:
                    td = t.detach().requires_grad_(True)
                    boundary.append((t, td))
                    f = NestedTensor(td, f.mask)
                cut.append(f)
            features = cut
        srcs, masks = [], []
        for l, feat in enumerate(features):
            src, mask = feat.decompose()
            assert mask is not None
            srcs.append(self.input_proj[l](src))
            masks.append(mark_no_padding(mask))
        for l in range(len(srcs), self.num_feature_levels):        # extra stride-2 levels (:166-178)
            src = self.input_proj[l](features[-1].tensors if l == len(features) else srcs[-1])
            mask = mark_no_padding(torch.zeros((src.shape[0],) + tuple(src.shape[-2:]), dtype=torch.bool, device=src.device))
            pos.append(self.backbone[1](NestedTensor(src, mask)).to(src.dtype))
            srcs.append(src)
            masks.append(mask)
        return srcs, masks, pos

    def forward(self, images, calibs, targets, img_sizes, dn_args=None):
        """images [B,3,H,W]; calibs [B,3,4] (only the focal length calibs[:,0,0] is read); img_sizes
        [B,2] (w,h); targets / dn_args unused on the default path.  Returns the prediction dict."""
        srcs, masks, pos = self.pyramid(images)

        query_embeds = self.query_embed.weight if self.training else self.query_embed.weight[:self.num_queries]

        dp_srcs = [_cut.at("msda", s) for s in srcs] if _cut.active("msda") else srcs
        depth_logits, depth_pos_embed, weighted_depth, depth_pos_embed_ip = self.depth_predictor(dp_srcs, masks[1], pos[1])
        # the decoder evaluates the five heads that read a level's output together (first layers as one GEMM).  The head
        # lists are handed over per forward, through the decoder's __dict__: not registered sub-modules (the state_dict
        # keeps the reference's key names), and THIS replica's modules when a DataParallel wrapper has replicated the model
        self.depthaware_transformer.decoder.__dict__["fused_heads"] = \
            (self.class_embed, self.depth_embed, self.angle_embed) if self.fuse_heads else None
        hs, init_reference, inter_references, inter_references_dim, _, _ = self.depthaware_transformer(
            srcs, masks, pos, query_embeds, depth_pos_embed, depth_pos_embed_ip)

        # Prediction heads (:222-262).  The per-level MLPs have their own weights; everything after them
        # is evaluated once on level-stacked [L, B, Q, .] tensors instead of once per decoder level.
        head_dtype = self.class_embed[0].weight.dtype          # heads stay fp32 even behind a bf16 body
        L = hs.shape[0]
        weighted_depth = weighted_depth.to(head_dtype)
        calibs, img_sizes = calibs.to(head_dtype), img_sizes.to(head_dtype)
        focal = calibs[:, 0, 0].view(1, -1, 1)
        img_h = img_sizes[:, 1].view(1, -1, 1)
        # level 0 refines the initial reference, level l > 0 the reference left by level l-1 (:226-236).  A
        # 2-component reference only shifts (cx, cy): padded with zeros it adds nothing to (l, r, t, b)
        head_out = self.depthaware_transformer.decoder.__dict__.get("head_outputs") or []
        fused = len(head_out) == L                             # the decoder already evaluated the heads (fused first layers)
        from .. import head_tail_ext
        if fused and inter_references.shape[-1] == 6 and head_dtype == torch.float32 and \
                head_tail_ext.usable(head_out[0][0], weighted_depth, calibs, img_sizes, inter_references_dim):
            # everything between the heads' raw outputs and the predictions in one launch each way (csrc/head_tail.hip)
            coord, depth_ave = head_tail_ext.head_tail(
                torch.stack([head_out[lvl][0] for lvl in range(L)]), init_reference.to(head_dtype), inter_references[:L - 1],
                inter_references_dim[:L], torch.stack([head_out[lvl][1] for lvl in range(L)]), weighted_depth, img_sizes[:, 1], calibs[:, 0, 0])
            size3d = inter_references_dim[:L].to(head_dtype)
            classes = torch.stack([head_out[lvl][3] for lvl in range(L)])
            angles = torch.stack([head_out[lvl][2] for lvl in range(L)])
            self.depthaware_transformer.decoder.__dict__["head_outputs"] = []
            return self._outputs(classes, coord, size3d, angles, depth_ave, depth_logits)
        hs = hs.to(head_dtype)
        first = inverse_sigmoid(init_reference.to(head_dtype))
        if first.shape[-1] == 2:
            first = F.pad(first, (0, 4))
        later = inverse_sigmoid(inter_references[:L - 1].to(head_dtype))
        if later.shape[-1] == 2:
            later = F.pad(later, (0, 4))
        reference = torch.cat((first[None], later), 0)
        box = torch.stack([head_out[lvl][0] if fused else self.bbox_embed[lvl](hs[lvl]) for lvl in range(L)]) + reference
        coord = box.sigmoid()                                      # (cx, cy, l, r, t, b) of the 3D centre / 2D box
        size3d = inter_references_dim[:L].to(head_dtype)
        # depth from geometry: f * H3d / h2d  (:240-242)
        h2d = torch.clamp((coord[..., 4] + coord[..., 5]) * img_h, min=1.0)
        depth_geo = size3d[..., 0] / h2d * focal
        depth_reg = torch.stack([head_out[lvl][1] if fused else self.depth_embed[lvl](hs[lvl]) for lvl in range(L)])
        # depth read from the predicted depth map at the (detached) 3D centre (:248-253): one bilinear
        # lookup for the queries of all levels
        B, Q = coord.shape[1], coord.shape[2]
        centre = ((coord[..., :2] - 0.5) * 2).detach().permute(1, 0, 2, 3).reshape(B, L * Q, 1, 2)
        depth_map = F.grid_sample(weighted_depth.unsqueeze(1), centre, mode='bilinear', align_corners=True)
        depth_map = depth_map.view(B, L, Q).permute(1, 0, 2)
        depth_ave = torch.cat([((1. / (depth_reg[..., 0:1].sigmoid() + 1e-6) - 1.) + depth_geo.unsqueeze(-1)
                                + depth_map.unsqueeze(-1)) / 3, depth_reg[..., 1:2]], -1)
        classes = torch.stack([head_out[lvl][3] if fused else self.class_embed[lvl](hs[lvl]) for lvl in range(L)])
        angles = torch.stack([head_out[lvl][2] if fused else self.angle_embed[lvl](hs[lvl]) for lvl in range(L)])
        self.depthaware_transformer.decoder.__dict__["head_outputs"] = []     # do not keep the graph alive past this forward

        return self._outputs(classes, coord, size3d, angles, depth_ave, depth_logits)

    def _outputs(self, classes, coord, size3d, angles, depth_ave, depth_logits):
        out = {'pred_logits': classes[-1], 'pred_boxes': coord[-1], 'pred_3d_dim': size3d[-1],
               'pred_depth': depth_ave[-1], 'pred_angle': angles[-1], 'pred_depth_map_logits': depth_logits}
        if self.aux_loss:
            out['aux_outputs'] = self._set_aux_loss(classes, coord, size3d, angles, depth_ave)
            # level-stacked views of the same predictions for the criterion (levels 0 .. L-1; not part of
            # the reference's dict)
            out['_levels'] = {'pred_logits': classes, 'pred_boxes': coord, 'pred_3d_dim': size3d,
                              'pred_depth': depth_ave, 'pred_angle': angles}
        return out

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord, outputs_3d_dim, outputs_angle, outputs_depth):
        return [{'pred_logits': a, 'pred_boxes': b, 'pred_3d_dim': c, 'pred_angle': d, 'pred_depth': e}
                for a, b, c, d, e in zip(outputs_class[:-1], outputs_coord[:-1], outputs_3d_dim[:-1],
                                         outputs_angle[:-1], outputs_depth[:-1])]


# ---- ground truth in static shape ----------------------------------------------------------------
_DUMMY_BOX3D = (0.5, 0.5, 0.1, 0.1, 0.1, 0.1)       # benign stand-in for padded slots (keeps every formula finite)


def pad_targets(targets, kmax=None):
    """list (one per image) of target dicts -> dict of [B, K, ...] tensors padded to K = kmax (default:
    the largest image), plus `valid` [B, K] bool and `num` [B] int32.  Padded slots hold harmless
    values and are masked out of every loss.  This is the only place where the ragged list is
    touched; everything downstream has shapes that do not depend on the number of objects, which is
    what lets the criterion run without host synchronisation (and inside a hipGraph)."""
    B = len(targets)
    sizes = [int(t["labels"].shape[0]) for t in targets]
    K = max(max(sizes) if sizes else 0, 1) if kmax is None else int(kmax)
    assert all(s <= K for s in sizes), "an image has more objects than kmax"
    dev = targets[0]["labels"].device
    fdt = targets[0]["boxes_3d"].dtype

    def padded(key, shape, dtype, fill):
        out = torch.empty((B, K) + shape, dtype=dtype, device=dev)
        out[...] = torch.as_tensor(fill, dtype=dtype, device=dev)
        for b, t in enumerate(targets):
            if sizes[b] and key in t:
                out[b, :sizes[b]] = t[key].reshape((sizes[b],) + shape).to(dtype)
        return out

    gt = {
        "labels": padded("labels", (), torch.int64, 0),
        "boxes": padded("boxes", (4,), fdt, (0.0, 0.0, 0.0, 0.0)),
        "boxes_3d": padded("boxes_3d", (6,), fdt, _DUMMY_BOX3D),
        "depth": padded("depth", (), fdt, 1.0),
        "size_3d": padded("size_3d", (3,), fdt, (1.0, 1.0, 1.0)),
        "heading_bin": padded("heading_bin", (), torch.int64, 0),
        "heading_res": padded("heading_res", (), fdt, 0.0),
    }
    num = torch.tensor(sizes, dtype=torch.int32)
    gt["valid"] = (torch.arange(K)[None, :] < num[:, None]).to(dev)
    gt["num"] = num.to(dev)
    gt["num_host"] = sizes
    return gt


_CONSTS = {}


def _device_const(values, dtype, device):
    key = (values, dtype, str(device))
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor(values, dtype=dtype, device=device)
    return _CONSTS[key]


def pad_targets_from_batch(targets):
    """The loader's collated targets (``[B, 50, ...]`` tensors + ``mask_2d``, kitti_dataset.py:299-312) -> the padded
    dict of ``pad_targets`` with K = 50, WITHOUT leaving the device: where the reference's trainer builds ragged
    per-image lists with boolean indexing (``val[bz][mask[bz]]``, trainer_helper.py:175-186: one host
    synchronisation per image and key), the kept objects are moved to the front of each row by a stable sort of
    the mask, in the same order, and the rest of the row becomes padding."""
    mask = targets["mask_2d"].bool()
    B, K = mask.shape
    order = torch.argsort((~mask).to(torch.int8), dim=1, stable=True)            # kept slots first, original order
    num = mask.sum(1).to(torch.int32)
    valid = torch.arange(K, device=mask.device)[None, :] < num[:, None]
    fdt = targets["boxes_3d"].dtype

    def take(key, dtype, fill):
        t = targets[key]
        t = t.reshape(B, K, -1)
        t = torch.gather(t, 1, order[..., None].expand(-1, -1, t.shape[-1])).to(dtype)
        if isinstance(fill, tuple) and len(set(fill)) == 1:
            fill = fill[0]
        if isinstance(fill, tuple):                                        # a per-column constant: cached on the device, so
            pad = _device_const(fill, dtype, t.device).expand_as(t)        # that a captured graph holds no host-to-device copy
            return torch.where(valid[..., None], t, pad)
        return t.masked_fill(~valid[..., None], fill)

    return {
        "labels": take("labels", torch.int64, 0)[..., 0],
        "boxes": take("boxes", fdt, (0.0, 0.0, 0.0, 0.0)),
        "boxes_3d": take("boxes_3d", fdt, _DUMMY_BOX3D),
        "depth": take("depth", fdt, 1.0)[..., 0],
        "size_3d": take("size_3d", fdt, (1.0, 1.0, 1.0)),
        "heading_bin": take("heading_bin", torch.int64, 0)[..., 0],
        "heading_res": take("heading_res", fdt, 0.0)[..., 0],
        "valid": valid, "num": num, "num_host": None,
    }


def assignment_from_indices(indices, gt, Q, group_num):
    """Reference-style matcher output (per image (query idx, target idx)) -> [B, G, K] int64, -1 = none."""
    B, K = gt["valid"].shape
    n = Q // group_num
    a = torch.full((B, group_num, K), -1, dtype=torch.int64)
    for b, (src, tgt) in enumerate(indices):
        src, tgt = src.to("cpu", torch.int64), tgt.to("cpu", torch.int64)
        if src.numel():
            a[b, torch.div(src, n, rounding_mode="floor"), tgt] = src
    return a.to(gt["valid"].device)


class _Pairs:
    """All (decoder layer, image, group, target-slot) pairs in static shape [L, B, G, K]: the matched
    query index (0 for unmatched / padded slots), the validity mask, and helpers to gather.  The L
    decoder layers are evaluated together: every loss below is one pass over [L, ...] tensors that
    ends in a per-layer vector [L], instead of L passes of the same ~230 tiny kernels."""

    def __init__(self, assign, gt):
        self.gt = gt
        self.ok = (assign >= 0) & gt["valid"][None, :, None, :]
        self.q = assign.clamp(min=0)

    def pred(self, x):                        # x [L, B, Q, D] -> [L, B, G, K, D]
        # gather, not x[l, b, q]: the backward of advanced indexing is a sort-based index_put (164 us per
        # call on MI355X); gather's backward is a scatter_add
        L, B, G, K = self.q.shape
        D = x.shape[-1]
        return x.gather(2, self.q.reshape(L, B, G * K, 1).expand(L, B, G * K, D)).view(L, B, G, K, D)

    def target(self, key):                    # gt[key] [B, K, ...] -> broadcast over layers and groups [L, B, G, K, ...]
        t = self.gt[key]
        L, B, G, K = self.q.shape
        return t[None, :, None].expand((L, B, G) + tuple(t.shape[1:]))

    def msum(self, x):                        # x [L, B, G, K] -> [L]: sum over the valid pairs of each layer
        return torch.where(self.ok, x, torch.zeros((), dtype=x.dtype, device=x.device)).flatten(1).sum(1)


class SetCriterion(nn.Module):
    """Hungarian matching of predictions to ground truth, then the eight MonoDETR losses on the
    matched pairs, for the last decoder layer and (auxiliary) every earlier one.

    Everything after `pad_targets` is shape-static: pairs are [L, B, G, K] with a validity mask, the
    assignment comes from the device solver on the GPU (scipy on CPU tensors), `num_boxes` may stay a
    device scalar.  Accepts the reference's list-of-dicts targets or an already padded dict."""

    def __init__(self, num_classes, matcher, weight_dict, focal_alpha, losses, group_num=11):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.focal_alpha = focal_alpha
        self.ddn_loss = DDNLoss()
        self.group_num = group_num
        # MDETR_FUSED_LOSSES=1: all matched-pair losses of all levels in one HIP launch each way
        # (csrc/pair_losses.hip).  Off until the kernel has had its first GPU validation
        # (tests/test_fused_gpu.py); its arithmetic is validated on the CPU (tests/test_fused_losses_cpu.py).
        import os
        self.fused_pair_losses = os.environ.get("MDETR_FUSED_LOSSES") == "1"

    # ---- individual losses (reference :320-458); `outputs` values are layer-stacked [L, B, Q, D], every
    # ---- returned entry is a per-layer vector [L] ------------------------------------------------------
    def _labels(self, outputs, pr, num_boxes, log=True):
        logits = outputs['pred_logits']
        L, B, Q, C = logits.shape
        labels = pr.target("labels")
        # scatter the matched labels; unmatched / padded pairs go to a dummy (Q-th) query that is dropped
        classes = torch.full((L, B, Q + 1), self.num_classes, dtype=torch.int64, device=logits.device)
        classes.scatter_(2, torch.where(pr.ok, pr.q, torch.full_like(pr.q, Q)).reshape(L, B, -1), labels.reshape(L, B, -1))
        onehot = F.one_hot(classes[..., :Q], self.num_classes + 1)[..., :-1].to(logits.dtype)
        losses = {'loss_ce': sigmoid_focal_loss(logits, onehot, num_boxes, alpha=self.focal_alpha, gamma=2,
                                                per_layer=True) * Q}
        if log:
            with torch.no_grad():
                hit = (pr.pred(logits).argmax(-1) == labels) & pr.ok
                nmatch = pr.ok.flatten(1).sum(1)
                acc = torch.where(nmatch > 0, hit.flatten(1).sum(1) * 100.0 / nmatch.clamp(min=1),
                                  torch.zeros((), device=logits.device))
            losses['class_error'] = 100 - acc
        return losses

    @torch.no_grad()
    def _cardinality(self, outputs, pr, num_boxes):
        logits = outputs['pred_logits']
        card_pred = (logits.argmax(-1) != logits.shape[-1] - 1).sum(2)              # [L, B]
        return {'cardinality_error': (card_pred.float() - pr.gt["num"].float()[None]).abs().mean(1)}

    def _center(self, outputs, pr, num_boxes):
        d = (pr.pred(outputs['pred_boxes'])[..., 0:2] - pr.target('boxes_3d')[..., 0:2]).abs().sum(-1)
        return {'loss_center': pr.msum(d) / num_boxes}

    def _boxes(self, outputs, pr, num_boxes):
        src, tgt = pr.pred(outputs['pred_boxes']), pr.target('boxes_3d')
        src = torch.where(pr.ok[..., None], src, tgt)                # padded pairs: identical boxes, finite GIoU
        l1 = (src[..., 2:6] - tgt[..., 2:6]).abs().sum(-1)
        giou = box_ops.elementwise_giou(box_ops.box_cxcylrtb_to_xyxy(src).reshape(-1, 4),
                                        box_ops.box_cxcylrtb_to_xyxy(tgt).reshape(-1, 4)).view_as(l1)
        return {'loss_bbox': pr.msum(l1) / num_boxes, 'loss_giou': pr.msum(1 - giou) / num_boxes}

    def _depths(self, outputs, pr, num_boxes):
        src, tgt = pr.pred(outputs['pred_depth']), pr.target('depth')
        mu, log_var = src[..., 0], src[..., 1]                      # Laplacian aleatoric uncertainty (:398-399)
        log_var = torch.where(pr.ok, log_var, torch.zeros_like(log_var))
        loss = 1.4142 * torch.exp(-log_var) * torch.abs(mu - tgt) + log_var
        return {'loss_depth': pr.msum(loss) / num_boxes}

    def _dims(self, outputs, pr, num_boxes):
        src, tgt = pr.pred(outputs['pred_3d_dim']), pr.target('size_3d')
        diff = torch.abs(src - tgt)
        rel = diff / tgt.detach()                                   # dimension-aware L1 (:410-416)
        ok3 = pr.ok[..., None].expand_as(diff)
        zero = torch.zeros((), dtype=diff.dtype, device=diff.device)
        rel_ok = torch.where(ok3, rel, zero)
        with torch.no_grad():                                       # mean |d| / mean relative |d| over the matched pairs of a layer
            comp = torch.where(ok3, diff, zero).flatten(1).sum(1) / rel_ok.flatten(1).sum(1).clamp(min=1e-12)
        return {'loss_dim': rel_ok.flatten(1).sum(1) * comp / num_boxes}

    def _angles(self, outputs, pr, num_boxes):
        pred = pr.pred(outputs['pred_angle'])                        # [L, B, G, K, 24]
        bins, res = pr.target('heading_bin'), pr.target('heading_res')
        cls_loss = F.cross_entropy(pred[..., 0:12].reshape(-1, 12), bins.reshape(-1), reduction='none').view_as(res)
        pred_res = pred[..., 12:24].gather(-1, bins[..., None]).squeeze(-1)       # residual of the true bin
        return {'loss_angle': pr.msum(cls_loss + (pred_res - res).abs()) / num_boxes}

    def _depth_map(self, outputs, pr, num_boxes):
        """Final layer only (:521-523): `outputs['pred_depth_map_logits']` is not layer-stacked; returns [1]."""
        logits = outputs['pred_depth_map_logits']
        gt = pr.gt
        if self.fused_pair_losses:                                   # one launch each way (ddn_loss_ext)
            from ..ddn_loss_ext import fused_ddn_loss
            bal = self.ddn_loss.balancer
            return {"loss_depth_map": fused_ddn_loss(logits, gt["boxes"], gt["depth"], gt["valid"], self.ddn_loss.alpha,
                                                     bal.fg_weight, bal.bg_weight).reshape(1)}
        H, W = logits.shape[-2:]
        b = gt["boxes"]                                              # x (80, 24, 80, 24) at 384x1280, no host->device copy
        boxes = box_ops.box_cxcywh_to_xyxy(torch.stack((b[..., 0] * W, b[..., 1] * H, b[..., 2] * W, b[..., 3] * H), -1))
        boxes = torch.where(gt["valid"][..., None], boxes, torch.zeros_like(boxes))              # padded slots cover nothing
        loss = self.ddn_loss(logits, boxes.reshape(-1, 4), gt["valid"].shape[1], gt["depth"].reshape(-1),
                             valid=gt["valid"].reshape(-1))
        return {"loss_depth_map": loss.reshape(1)}

    _LOSSES = {'labels': '_labels', 'cardinality': '_cardinality', 'boxes': '_boxes', 'depths': '_depths',
               'dims': '_dims', 'angles': '_angles', 'center': '_center', 'depth_map': '_depth_map'}
    _STACKED = ('pred_logits', 'pred_boxes', 'pred_3d_dim', 'pred_depth', 'pred_angle')

    def _get(self, loss, outputs, pr, num_boxes, **kwargs):
        assert loss in self._LOSSES, f'do you really want to compute {loss} loss?'
        return getattr(self, self._LOSSES[loss])(outputs, pr, num_boxes, **kwargs)

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        """Reference signature (:466-481): one layer's `outputs`, `indices` the matcher's list of
        (query idx, target idx); returns {name: 0-d tensor}."""
        gt = targets if isinstance(targets, dict) else pad_targets(targets)
        Q = outputs['pred_logits'].shape[1]
        G = self.group_num if Q % self.group_num == 0 and Q // self.group_num * self.group_num == Q and self.training else 1
        assign = assignment_from_indices(indices, gt, Q, G)
        one = {k: (v[None] if k in self._STACKED else v) for k, v in _widen(outputs).items() if torch.is_tensor(v)}
        return {k: v[0] for k, v in self._get(loss, one, _Pairs(assign[None], gt), num_boxes, **kwargs).items()}

    def weighted_total(self, losses):
        """sum_k weight_dict[k] * losses[k] over the weighted entries -- what the reference's trainer
        computes in a Python loop (lib/helpers/trainer_helper.py:141-143: ~24 multiplies and adds on 0-d
        tensors, each a kernel launch, and as many again in the backward) -- as one stack and one dot
        product."""
        keys = [k for k in losses if k in self.weight_dict]
        vec = torch.stack([losses[k] for k in keys])
        cache = self.__dict__.get("_weight_vec")
        if cache is None or cache[0] != (tuple(keys), vec.dtype, vec.device):
            w = torch.tensor([float(self.weight_dict[k]) for k in keys], dtype=vec.dtype, device=vec.device)
            cache = self.__dict__["_weight_vec"] = ((tuple(keys), vec.dtype, vec.device), w)
        return torch.dot(vec, cache[1])

    def forward(self, outputs, targets, mask_dict=None):
        """outputs: the model's dict; targets: list (one per image) of dicts with 'labels', 'boxes',
        'boxes_3d', 'depth', 'size_3d', 'heading_bin', 'heading_res' -- or the dict `pad_targets`
        makes of it.  Returns {name: 0-d tensor}, auxiliary layers suffixed _0, _1, ... (:511-530)."""
        final = _widen({k: v for k, v in outputs.items() if k not in ('aux_outputs', '_levels')})
        levels = outputs.get('_levels')
        if levels is not None:
            # `_levels` is level-first [L, B, Q, D]; a caller that gathers replica outputs along dim 0
            # (nn.DataParallel, what the reference's tools/train_val.py:55 wraps the model in) turns it into
            # [n*L, B/n, ...] while pred_* become [B, ...]: use it only when it still describes this batch
            ref = final['pred_logits']
            n_aux = len(outputs.get('aux_outputs', []))
            if not all(v.dim() == final[k].dim() + 1 and v.shape[0] == n_aux + 1 and v.shape[1:] == final[k].shape
                       for k, v in levels.items() if k in final) or 'pred_logits' not in levels or ref.dim() != 3:
                levels = None
        if levels is not None:
            # the model's own level-stacked predictions [L, B, Q, D] (levels 0 .. L-1, the final layer last)
            stacked = _widen(levels)
            L = stacked['pred_logits'].shape[0]
            names = [str(i) for i in range(L - 1)] + [None]                  # suffix of every level, None = final
        else:
            layers = [final] + [_widen(a) for a in outputs.get('aux_outputs', [])]
            L = len(layers)
            stacked = {k: torch.stack([lay[k] for lay in layers]) for k in self._STACKED}     # final layer first
            names = [None] + [str(i) for i in range(L - 1)]
        last = names.index(None)
        if 'pred_depth_map_logits' in final:
            stacked['pred_depth_map_logits'] = final['pred_depth_map_logits']
        group_num = self.group_num if self.training else 1
        gt = targets if isinstance(targets, dict) else pad_targets(targets)
        assign = self.matcher.assign_stacked(stacked['pred_logits'], stacked['pred_boxes'], gt, group_num)   # [L, B, G, K]

        if gt.get("num_global") is not None:
            # the caller already averaged the object count over the ranks (helpers/step_helper.TrainIteration: a collective
            # cannot sit inside a captured graph while RCCL's watchdog polls its events): a device scalar
            num_boxes = torch.clamp(gt["num_global"].to(torch.float32) * group_num, min=1)
        elif gt.get("num_host") is not None and not is_dist_avail_and_initialized():
            num_boxes = max(float(sum(gt["num_host"]) * group_num), 1.0)
        else:                                                               # stays on the device: no sync
            nb = gt["num"].sum().to(torch.float32) * group_num
            if is_dist_avail_and_initialized():
                torch.distributed.all_reduce(nb)
            num_boxes = torch.clamp(nb / get_world_size(), min=1)

        pr = _Pairs(assign, gt)
        losses, aux = {}, {}
        per_loss = []
        if self.fused_pair_losses and all(k in self._LOSSES for k in self.losses):
            # every per-pair loss of every level in one launch (pair_losses_ext); the depth-map loss stays below
            from ..pair_losses_ext import fused_pair_losses
            rows = fused_pair_losses(stacked, assign, gt, num_boxes, self.focal_alpha)
            wanted = {'labels': ('loss_ce', 'class_error'), 'cardinality': ('cardinality_error',),
                      'center': ('loss_center',), 'boxes': ('loss_bbox', 'loss_giou'), 'depths': ('loss_depth',),
                      'dims': ('loss_dim',), 'angles': ('loss_angle',)}
            for loss in self.losses:
                if loss == 'depth_map':
                    per_loss.append(self._get(loss, stacked, pr, num_boxes))
                else:
                    per_loss.append({name: rows[name] for name in wanted[loss]})
        else:
            per_loss = [self._get(loss, stacked, pr, num_boxes) for loss in self.losses]
        for ld in per_loss:
            for name, vec in ld.items():
                parts = vec.unbind(0)
                # auxiliary layers: no depth-map loss (:521-523) and no class_error (log=False, :524-526)
                if name in ('loss_depth_map', 'class_error'):
                    losses[name] = parts[0] if name == 'loss_depth_map' else parts[last]
                    continue
                losses[name] = parts[last]
                aux.update({f'{name}_{names[i]}': parts[i] for i in range(L) if i != last})
        losses.update(aux)
        return losses


def _widen(d):
    return {k: (v.float() if torch.is_tensor(v) and v.dtype in (torch.bfloat16, torch.float16) else v) for k, v in d.items()}


def build(cfg):
    backbone = build_backbone(cfg)
    depthaware_transformer = build_depthaware_transformer(cfg)
    depth_predictor = DepthPredictor(cfg)
    model = MonoDETR(backbone, depthaware_transformer, depth_predictor, num_classes=cfg['num_classes'],
                     num_queries=cfg['num_queries'], aux_loss=cfg['aux_loss'],
                     num_feature_levels=cfg['num_feature_levels'], with_box_refine=cfg['with_box_refine'],
                     two_stage=cfg['two_stage'], init_box=cfg['init_box'], use_dab=cfg['use_dab'],
                     two_stage_dino=cfg['two_stage_dino'])
    matcher = build_matcher(cfg)

    weight_dict = {'loss_ce': cfg['cls_loss_coef'], 'loss_bbox': cfg['bbox_loss_coef'],
                   'loss_giou': cfg['giou_loss_coef'], 'loss_dim': cfg['dim_loss_coef'],
                   'loss_angle': cfg['angle_loss_coef'], 'loss_depth': cfg['depth_loss_coef'],
                   'loss_center': cfg['3dcenter_loss_coef'], 'loss_depth_map': cfg['depth_map_loss_coef']}
    if cfg['use_dn']:
        for k, src in (('tgt_loss_ce', 'cls_loss_coef'), ('tgt_loss_bbox', 'bbox_loss_coef'),
                       ('tgt_loss_giou', 'giou_loss_coef'), ('tgt_loss_angle', 'angle_loss_coef'),
                       ('tgt_loss_center', '3dcenter_loss_coef')):
            weight_dict[k] = cfg[src]
    if cfg['aux_loss']:
        aux = {}
        for i in range(cfg['dec_layers'] - 1):
            aux.update({k + f'_{i}': v for k, v in weight_dict.items()})
        aux.update({k + '_enc': v for k, v in weight_dict.items()})
        weight_dict.update(aux)

    losses = ['labels', 'boxes', 'cardinality', 'depths', 'dims', 'angles', 'center', 'depth_map']
    criterion = SetCriterion(cfg['num_classes'], matcher=matcher, weight_dict=weight_dict,
                             focal_alpha=cfg['focal_alpha'], losses=losses)
    criterion.to(torch.device(cfg['device']) if (cfg['device'] != 'cuda' or torch.cuda.is_available()) else 'cpu')
    return model, criterion
