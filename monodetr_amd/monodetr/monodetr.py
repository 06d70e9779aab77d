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
from ..utils.misc import (NestedTensor, accuracy, get_world_size, inverse_sigmoid,
                          is_dist_avail_and_initialized, mark_no_padding)
from .backbone import build_backbone
from .depth_predictor import DepthPredictor
from .depth_predictor.ddn_loss import DDNLoss
from .depthaware_transformer import MLP, build_depthaware_transformer
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
            return nn.Sequential(nn.Conv2d(cin, hidden_dim, kernel_size=k, stride=stride, padding=k // 2),
                                 nn.GroupNorm(32, hidden_dim))
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
        else:
            nn.init.constant_(bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([class_embed for _ in range(num_pred)])
            self.bbox_embed = nn.ModuleList([bbox_embed for _ in range(num_pred)])
            self.dim_embed_3d = nn.ModuleList([dim_embed_3d for _ in range(num_pred)])
            self.angle_embed = nn.ModuleList([angle_embed for _ in range(num_pred)])
            self.depth_embed = nn.ModuleList([depth_embed for _ in range(num_pred)])
            self.depthaware_transformer.decoder.bbox_embed = None

    def forward(self, images, calibs, targets, img_sizes, dn_args=None):
        """images [B,3,H,W]; calibs [B,3,4] (only the focal length calibs[:,0,0] is read); img_sizes
        [B,2] (w,h); targets / dn_args unused on the default path.  Returns the prediction dict."""
        features, pos = self.backbone(images)
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

        query_embeds = self.query_embed.weight if self.training else self.query_embed.weight[:self.num_queries]

        depth_logits, depth_pos_embed, weighted_depth, depth_pos_embed_ip = self.depth_predictor(srcs, masks[1], pos[1])
        hs, init_reference, inter_references, inter_references_dim, _, _ = self.depthaware_transformer(
            srcs, masks, pos, query_embeds, depth_pos_embed, depth_pos_embed_ip)

        coords, classes, dims3d, depths, angles = [], [], [], [], []
        head_dtype = self.class_embed[0].weight.dtype          # heads stay fp32 even behind a bf16 body
        hs = hs.to(head_dtype)
        weighted_depth = weighted_depth.to(head_dtype)
        calibs, img_sizes = calibs.to(head_dtype), img_sizes.to(head_dtype)
        focal = calibs[:, 0, 0].unsqueeze(1)
        img_h = img_sizes[:, 1:2]
        for lvl in range(hs.shape[0]):
            reference = inverse_sigmoid(init_reference if lvl == 0 else inter_references[lvl - 1]).to(head_dtype)
            box = self.bbox_embed[lvl](hs[lvl])
            if reference.shape[-1] == 6:
                box = box + reference
            else:
                assert reference.shape[-1] == 2
                box = torch.cat((box[..., :2] + reference, box[..., 2:]), -1)
            coord = box.sigmoid()                                  # (cx, cy, l, r, t, b) of the 3D centre / 2D box
            size3d = inter_references_dim[lvl].to(head_dtype)
            # depth from geometry: f * H3d / h2d  (:240-242)
            h2d = torch.clamp((coord[:, :, 4] + coord[:, :, 5]) * img_h, min=1.0)
            depth_geo = size3d[:, :, 0] / h2d * focal
            depth_reg = self.depth_embed[lvl](hs[lvl])
            # depth read from the predicted depth map at the (detached) 3D centre (:248-253)
            centre = ((coord[..., :2] - 0.5) * 2).unsqueeze(2).detach()
            depth_map = F.grid_sample(weighted_depth.unsqueeze(1), centre, mode='bilinear', align_corners=True).squeeze(1)
            depth_ave = torch.cat([((1. / (depth_reg[:, :, 0:1].sigmoid() + 1e-6) - 1.) + depth_geo.unsqueeze(-1) + depth_map) / 3,
                                   depth_reg[:, :, 1:2]], -1)
            coords.append(coord)
            classes.append(self.class_embed[lvl](hs[lvl]))
            dims3d.append(size3d)
            depths.append(depth_ave)
            angles.append(self.angle_embed[lvl](hs[lvl]))

        out = {'pred_logits': classes[-1], 'pred_boxes': coords[-1], 'pred_3d_dim': dims3d[-1],
               'pred_depth': depths[-1], 'pred_angle': angles[-1], 'pred_depth_map_logits': depth_logits}
        if self.aux_loss:
            out['aux_outputs'] = self._set_aux_loss(classes, coords, dims3d, angles, depths)
        return out

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord, outputs_3d_dim, outputs_angle, outputs_depth):
        return [{'pred_logits': a, 'pred_boxes': b, 'pred_3d_dim': c, 'pred_angle': d, 'pred_depth': e}
                for a, b, c, d, e in zip(outputs_class[:-1], outputs_coord[:-1], outputs_3d_dim[:-1],
                                         outputs_angle[:-1], outputs_depth[:-1])]


class _Matched:
    """Index triple of one decoder layer's assignment: image, query and (global) target index of
    every matched pair, plus the ground truth gathered in that order."""

    def __init__(self, indices, gt, device):
        offs = gt["offsets"]
        img = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        qry = torch.cat([src for src, _ in indices])
        tgt = torch.cat([j + offs[i] for i, (_, j) in enumerate(indices)])
        packed = torch.stack((img, qry, tgt)).to(device, non_blocking=True)       # one H2D copy
        self.img, self.qry, self.tgt = packed[0], packed[1], packed[2]
        self.idx = (self.img, self.qry)
        self.gt = gt

    def target(self, key):
        return self.gt[key][self.tgt]


class SetCriterion(nn.Module):
    """Hungarian matching of predictions to ground truth, then the eight MonoDETR losses on the
    matched pairs, for the last decoder layer and (auxiliary) every earlier one."""

    def __init__(self, num_classes, matcher, weight_dict, focal_alpha, losses, group_num=11):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.focal_alpha = focal_alpha
        self.ddn_loss = DDNLoss()
        self.group_num = group_num

    # ---- ground truth, concatenated once per call ------------------------------------------------
    @staticmethod
    def _gather_targets(targets):
        sizes = [len(t["labels"]) for t in targets]
        gt = {k: torch.cat([t[k] for t in targets], 0)
              for k in ("labels", "boxes", "boxes_3d", "depth", "size_3d", "heading_bin", "heading_res") if k in targets[0]}
        gt["sizes"] = sizes
        gt["offsets"] = [0] + list(torch.tensor(sizes).cumsum(0).tolist())[:-1]
        return gt

    def _matched(self, indices, targets, outputs, gt=None):
        gt = gt if gt is not None else self._gather_targets(targets)
        return _Matched(indices, gt, outputs["pred_logits"].device)

    # ---- individual losses (reference :320-458); `m` is a _Matched -----------------------------
    def loss_labels(self, outputs, targets, indices, num_boxes, log=True, m=None):
        m = m or self._matched(indices, targets, outputs)
        logits = outputs['pred_logits']
        labels_o = m.target("labels").reshape(-1).long()
        classes = torch.full(logits.shape[:2], self.num_classes, dtype=torch.int64, device=logits.device)
        classes[m.idx] = labels_o
        onehot = F.one_hot(classes, self.num_classes + 1)[..., :-1].to(logits.dtype)
        losses = {'loss_ce': sigmoid_focal_loss(logits, onehot, num_boxes, alpha=self.focal_alpha, gamma=2) * logits.shape[1]}
        if log:
            losses['class_error'] = 100 - accuracy(logits[m.idx], labels_o)[0]
        return losses

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, num_boxes, m=None):
        logits = outputs['pred_logits']
        sizes = m.gt["sizes"] if m is not None else [len(v["labels"]) for v in targets]
        tgt_lengths = torch.as_tensor(sizes, device=logits.device)
        card_pred = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1)
        return {'cardinality_error': F.l1_loss(card_pred.float(), tgt_lengths.float())}

    def loss_3dcenter(self, outputs, targets, indices, num_boxes, m=None):
        m = m or self._matched(indices, targets, outputs)
        src = outputs['pred_boxes'][:, :, 0:2][m.idx]
        return {'loss_center': F.l1_loss(src, m.target('boxes_3d')[:, 0:2], reduction='none').sum() / num_boxes}

    def loss_boxes(self, outputs, targets, indices, num_boxes, m=None):
        m = m or self._matched(indices, targets, outputs)
        src, tgt = outputs['pred_boxes'][m.idx], m.target('boxes_3d')
        giou = box_ops.elementwise_giou(box_ops.box_cxcylrtb_to_xyxy(src), box_ops.box_cxcylrtb_to_xyxy(tgt))
        return {'loss_bbox': F.l1_loss(src[:, 2:6], tgt[:, 2:6], reduction='none').sum() / num_boxes,
                'loss_giou': (1 - giou).sum() / num_boxes}

    def loss_depths(self, outputs, targets, indices, num_boxes, m=None):
        m = m or self._matched(indices, targets, outputs)
        src = outputs['pred_depth'][m.idx]
        tgt = m.target('depth').reshape(-1)
        mu, log_var = src[:, 0], src[:, 1]                     # Laplacian aleatoric uncertainty (:398-399)
        loss = 1.4142 * torch.exp(-log_var) * torch.abs(mu - tgt) + log_var
        return {'loss_depth': loss.sum() / num_boxes}

    def loss_dims(self, outputs, targets, indices, num_boxes, m=None):
        m = m or self._matched(indices, targets, outputs)
        src, tgt = outputs['pred_3d_dim'][m.idx], m.target('size_3d')
        rel = torch.abs(src - tgt) / tgt.detach()              # dimension-aware L1 (:410-416)
        with torch.no_grad():
            comp = F.l1_loss(src, tgt) / rel.mean()
        return {'loss_dim': (rel * comp).sum() / num_boxes}

    def loss_angles(self, outputs, targets, indices, num_boxes, m=None):
        m = m or self._matched(indices, targets, outputs)
        pred = outputs['pred_angle'][m.idx].view(-1, 24)
        bins = m.target('heading_bin').view(-1).long()
        res = m.target('heading_res').view(-1)
        cls_loss = F.cross_entropy(pred[:, 0:12], bins, reduction='none')
        pred_res = pred[:, 12:24].gather(1, bins.view(-1, 1)).squeeze(1)       # residual of the true bin
        return {'loss_angle': (cls_loss + F.l1_loss(pred_res, res, reduction='none')).sum() / num_boxes}

    def loss_depth_map(self, outputs, targets, indices, num_boxes, m=None):
        logits = outputs['pred_depth_map_logits']
        gt = m.gt if m is not None else self._gather_targets(targets)
        H, W = logits.shape[-2:]
        scale = torch.tensor([W, H, W, H], dtype=gt["boxes"].dtype, device=gt["boxes"].device)   # 80,24,80,24 at 384x1280
        boxes = box_ops.box_cxcywh_to_xyxy(gt["boxes"] * scale)
        return {"loss_depth_map": self.ddn_loss(logits, boxes, gt["sizes"], gt["depth"].squeeze(dim=1))}

    def _get_src_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for (src, _) in indices])

    def _get_tgt_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        return batch_idx, torch.cat([tgt for (_, tgt) in indices])

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        loss_map = {'labels': self.loss_labels, 'cardinality': self.loss_cardinality, 'boxes': self.loss_boxes,
                    'depths': self.loss_depths, 'dims': self.loss_dims, 'angles': self.loss_angles,
                    'center': self.loss_3dcenter, 'depth_map': self.loss_depth_map}
        assert loss in loss_map, f'do you really want to compute {loss} loss?'
        return loss_map[loss](outputs, targets, indices, num_boxes, **kwargs)

    def forward(self, outputs, targets, mask_dict=None):
        """outputs: the model's dict; targets: list (one per image) of dicts with 'labels', 'boxes',
        'boxes_3d', 'depth', 'size_3d', 'heading_bin', 'heading_res'.  Returns {name: 0-d tensor}."""
        def widen(d):
            return {k: (v.float() if torch.is_tensor(v) and v.dtype in (torch.bfloat16, torch.float16) else v) for k, v in d.items()}
        final = widen({k: v for k, v in outputs.items() if k != 'aux_outputs'})
        layers = [final] + [widen(a) for a in outputs.get('aux_outputs', [])]
        group_num = self.group_num if self.training else 1
        all_indices = self.matcher.match_layers(layers, targets, group_num=group_num)

        num_boxes = sum(len(t["labels"]) for t in targets) * group_num
        if is_dist_avail_and_initialized():
            nb = torch.as_tensor([num_boxes], dtype=torch.float, device=final["pred_logits"].device)
            torch.distributed.all_reduce(nb)
            num_boxes = torch.clamp(nb / get_world_size(), min=1)[0]          # stays on device: no sync
        else:
            num_boxes = max(float(num_boxes), 1.0)

        gt = self._gather_targets(targets)
        losses = {}
        for li, (layer_out, indices) in enumerate(zip(layers, all_indices)):
            m = _Matched(indices, gt, final["pred_logits"].device)
            for loss in self.losses:
                if li > 0 and loss == 'depth_map':       # depth-map loss only on the final layer (:521-523)
                    continue
                kwargs = {'log': False} if (li > 0 and loss == 'labels') else {}
                ld = self.get_loss(loss, layer_out, targets, indices, num_boxes, m=m, **kwargs)
                losses.update(ld if li == 0 else {k + f'_{li - 1}': v for k, v in ld.items()})
        return losses


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
