"""ResNet backbone with frozen BatchNorm -- mirror of lib/models/monodetr/backbone.py
(``FrozenBatchNorm2d`` :27-64, ``BackboneBase`` :67-90, ``Backbone`` :93-106, ``Joiner`` :109-126,
``build_backbone`` :129-135).

The reference takes the network from torchvision (``torchvision.models.resnet50`` wrapped in
``IntermediateLayerGetter``, backbone.py:17-19,82,100-102), which is not a dependency here.  The
bottleneck ResNet (v1.5: stride on the 3x3 conv) is defined below with the same module names, so
``backbone.0.body.*`` state_dict keys of published checkpoints load unchanged.
ImageNet weights are not downloaded at construction (reference :102; there is no network): pass
``cfg['backbone_weights']`` (a state_dict path) or load a full checkpoint afterwards.

MI355X notes: every BatchNorm is frozen, i.e. an affine map per channel.  ``forward`` folds it into
the preceding convolution (conv(x, W*s) + t  ==  conv(x, W)*s + t), which removes one full
read+write pass over every activation of the backbone; gradients still reach W through W*s.
"""
import os
from typing import Dict, List

import torch

from .. import _tune
import torch.nn.functional as F
from torch import nn

from ..utils.misc import NestedTensor, mark_no_padding
from .. import bias_act_ext, conv3x3_ext, conv_stem_ext, conv_taps_ext, decimate_ext
from .linear import (pointwise_conv, pointwise_conv_residual_relu, pointwise_conv_skip, pointwise_eligible, pointwise_relu_fusable,
                     pointwise_residual_relu_eligible, skip_relu_fusable)
from .position_encoding import build_position_encoding


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (all buffers), eps inside the rsqrt."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        state_dict.pop(prefix + 'num_batches_tracked', None)       # reference :44-52
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    def affine(self):
        """(scale, shift) with y = x * scale + shift.  The four buffers never change during training,
        so the pair is computed once and cached (5 tiny kernels per BN per forward otherwise: 265
        launches for ResNet-50); any buffer update (load_state_dict, .to(), manual edit) bumps the
        tensors' version counters / identities and invalidates the cache."""
        key = tuple((t.data_ptr(), t._version, t.dtype, t.device) for t in
                    (self.weight, self.bias, self.running_mean, self.running_var))
        if getattr(self, "_affine_key", None) != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + self.eps).rsqrt()
                self._affine = (scale, self.bias - self.running_mean * scale)
            self._affine_key = key
        return self._affine

    def forward(self, x):
        scale, shift = self.affine()
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


class _FoldAll(torch.autograd.Function):
    """folded_i = (W_i * scale_i).to(dt) for every trainable convolution of the backbone in two
    multi-tensor launches (and two more in the backward: dW_i = dfolded_i.to(W dtype) * scale_i)
    instead of two small kernels per convolution each way.  `scales` are the frozen-BN scales expanded
    to the weights' shapes and strides, which is what keeps torch._foreach_* on its fused path."""

    @staticmethod
    def forward(ctx, scales, dt, *weights):
        ctx.scales, ctx.wdtype = scales, weights[0].dtype
        ctx.like = weights
        out = torch._foreach_mul(list(weights), scales)
        if dt != ctx.wdtype:
            low = [torch.empty_like(w, dtype=dt) for w in weights]
            torch._foreach_copy_(low, out)
            out = low
        return tuple(out)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        from .. import chunk_sums
        chunk_sums.flush()                           # (see _FoldKernel.backward)
        grads = [g if g is not None else torch.zeros_like(w, dtype=ctx.wdtype) for g, w in zip(grads, ctx.like)]
        if grads[0].dtype != ctx.wdtype or any(g.dtype != ctx.wdtype for g in grads):
            wide = [torch.empty_like(w) for w in ctx.like]
            torch._foreach_copy_(wide, grads)
            grads = wide
            torch._foreach_mul_(grads, ctx.scales)
        else:
            grads = torch._foreach_mul(grads, ctx.scales)
        return (None, None) + tuple(grads)


class _FoldKernel(torch.autograd.Function):
    """`_FoldAll` as ONE launch each way (csrc/wfold.hip): folded_i = bf16(W_i * scale_i) straight from the fp32 channels-last
    parameters and the per-channel scales, the 3x3 weights' [C][tap][O] copies for the input-gradient kernels from the same
    launch (handed out as non-differentiable outputs), dW_i = float(dfolded_i) * scale_i in the backward.  Same bits as `_FoldAll`."""

    @staticmethod
    def forward(ctx, svecs, want_t, *weights):
        from .. import wfold_ext
        folded, folded_t = wfold_ext.fold_weights(weights, svecs, want_t)
        ctx.svecs, ctx.like = svecs, weights
        extra = [t for t in folded_t if t is not None]
        ctx.mark_non_differentiable(*extra)
        ctx.set_materialize_grads(False)             # (no zero tensors for the copies' absent gradients: 13 fill launches)
        return tuple(folded) + tuple(extra)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        from .. import chunk_sums, wfold_ext
        chunk_sums.flush()                           # the folded weights' gradients may be registered chunk sums: they are READ here
        like = ctx.like
        grads = [g if g is not None else torch.zeros_like(w, dtype=torch.bfloat16) for g, w in zip(grads[:len(like)], like)]
        if wfold_ext.grads_supported(grads, like):
            out = wfold_ext.unfold_grads(grads, ctx.svecs, like)
        else:                                        # a gradient that is not a dense bf16 channels-last tensor (a library fall-back upstream)
            out = [g.to(w.dtype) * s.view(-1, 1, 1, 1) for g, w, s in zip(grads, like, ctx.svecs)]
        return (None, None) + tuple(out)


def prefold(pairs, dt):
    """Fold the frozen BN of every trainable (conv, bn) pair into its weight for THIS forward pass and
    park the result on the conv module (`conv_bn` consumes it)."""
    if not pairs:
        return
    key = tuple(bn.affine()[0].data_ptr() for _, bn in pairs) + tuple(c.weight.data_ptr() for c, _ in pairs) + (dt,)
    cache = pairs[0][0].__dict__.get("_prefold_cache")
    if cache is None or cache[0] != key:
        scales, shifts, svecs = [], [], []
        with torch.no_grad():
            for conv, bn in pairs:
                scale, shift = bn.affine()
                full = torch.empty_like(conv.weight)
                full.copy_(scale.to(conv.weight.dtype).view(-1, 1, 1, 1).expand_as(conv.weight))
                scales.append(full)
                shifts.append(shift.to(dt))
                svecs.append(scale.to(conv.weight.dtype).contiguous())
        cache = pairs[0][0].__dict__["_prefold_cache"] = (key, scales, shifts, svecs)
    from .. import wfold_ext
    weights = [conv.weight for conv, _ in pairs]
    if wfold_ext.ENABLED and wfold_ext.supported(weights, cache[3], dt):
        # one launch (csrc/wfold.hip); a 3x3 weight's [C][tap][O] copy rides on the folded tensor for the input-gradient kernels
        # (conv3x3_ext / conv_taps_ext look for it: `_mdetr_ihwo`)
        want_t = [tuple(conv.kernel_size) == (3, 3) for conv, _ in pairs]
        out = _FoldKernel.apply(cache[3], want_t, *weights)
        folded, extra = out[:len(pairs)], iter(out[len(pairs):])
        for w, t in zip(folded, want_t):
            if t:
                w._mdetr_ihwo = next(extra)
    else:
        folded = _FoldAll.apply(cache[1], dt, *weights)
    for (conv, _), w, b in zip(pairs, folded, cache[2]):
        conv.__dict__["_prefolded"] = (w, b)


def frozen_fold(conv, bn, dt):
    """(W * scale, shift) of a FROZEN convolution + frozen BN pair in dtype `dt`, cached on the module until a buffer or the weight
    changes."""
    scale, shift = bn.affine()
    key = (conv.weight.data_ptr(), conv.weight._version, bn._affine_key, dt)
    if getattr(conv, "_folded_key", None) != key:
        with torch.no_grad():
            conv._folded = ((conv.weight * scale.view(-1, 1, 1, 1)).to(dt).contiguous(memory_format=torch.channels_last),
                            shift.to(dt))
        conv._folded_key = key
    return conv._folded


def conv_bn(x, conv, bn, relu, skip_out=False, relu_token=None, hand_out_token=False):
    """conv -> frozen BN (-> ReLU), with the BN folded into the convolution's weight and bias.
    For frozen convolutions (stem, layer1) the folded weight itself is cached.
    skip_out: -> (result, x') with x' == x for the identity connection that follows (linear.pointwise_conv_skip: its gradient is
    folded into this convolution's input-gradient GEMM); (result, x) where that form does not apply.
    relu_token (with skip_out): x is a ReLU output that only this call consumes -- its backward mask moves into this convolution's
    input gradient where the kernel takes it (linear.ReluToken).  hand_out_token: the result (with relu) goes to exactly one consumer
    inside the caller's module, which may do the same for THIS ReLU."""
    if skip_out:
        if isinstance(bn, FrozenBatchNorm2d) and conv.bias is None and conv.weight.requires_grad and x.requires_grad \
                and tuple(conv.stride) == (1, 1) and torch.is_grad_enabled() \
                and pointwise_eligible(x, conv.kernel_size, (1, 1), conv.padding, conv.groups):
            pre = conv.__dict__.get("_prefolded", None)
            dt = x.dtype if x.dtype in (torch.bfloat16, torch.float16) and not torch.is_autocast_enabled() else conv.weight.dtype
            if pre is not None and pre[0].dtype == dt == x.dtype and (not relu or skip_relu_fusable(
                    pre[1], x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]), pre[0].reshape(pre[0].shape[0], -1))):
                conv.__dict__.pop("_prefolded", None)
                return pointwise_conv_skip(x, pre[0], pre[1], relu=relu, relu_token=relu_token, hand_out_token=hand_out_token)
        return conv_bn(x, conv, bn, relu), x
    if isinstance(bn, FrozenBatchNorm2d) and conv.bias is None:
        scale, shift = bn.affine()
        dt = x.dtype if x.dtype in (torch.bfloat16, torch.float16) and not torch.is_autocast_enabled() else conv.weight.dtype
        pre = conv.__dict__.pop("_prefolded", None)
        if pre is not None and pre[0].dtype == dt:
            w, b = pre
        elif not conv.weight.requires_grad:
            w, b = frozen_fold(conv, bn, dt)
        else:
            # trainable convolution: the fold is part of the autograd graph (dW = dW_folded * scale); the
            # casts of the frozen (scale, shift) pair are cached per dtype
            cast = getattr(bn, "_affine_cast", None)
            if cast is None or cast[0] != (bn._affine_key, conv.weight.dtype, dt):
                cast = bn._affine_cast = ((bn._affine_key, conv.weight.dtype, dt),
                                          scale.to(conv.weight.dtype).view(-1, 1, 1, 1), shift.to(dt))
            w = (conv.weight * cast[1]).to(dt)
            b = cast[2]
        # (decimate only when the decimated tensor WILL take the stride-1 token path: the fall-backs below still apply the
        # convolution's own stride and would stride the gathered pixels a second time)
        if decimate_ext.ENABLED and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (0, 0) \
                and conv.groups == 1 and x.dtype == w.dtype and decimate_ext.supported(x) and not torch.is_autocast_enabled() \
                and pointwise_eligible(x, conv.kernel_size, (1, 1), conv.padding, conv.groups):
            # the projection shortcut of a stage's first block: a 1x1 / stride-2 convolution reads only the pixels it keeps -- gather
            # them (csrc/decimate.hip) and run the stride-1 token GEMM path below
            x = decimate_ext.decimate2(x)
            stride1 = True
        else:
            stride1 = tuple(conv.stride) == (1, 1)
        if stride1 and pointwise_eligible(x, conv.kernel_size, (1, 1), conv.padding, conv.groups) and x.dtype == w.dtype:
            if relu and pointwise_relu_fusable(x, w, b):
                return pointwise_conv(x, w, b, relu=True)            # ReLU in the GEMM's epilogue
            x = pointwise_conv(x, w, b)
        elif relu and conv_stem_ext.ENABLED and conv_stem_ext.supported(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
            return conv_stem_ext.conv_stem(x, w, shift)              # the frozen 7x7 / stride-2 stem (csrc/conv_stem.hip), shift + ReLU inside
        elif conv_taps_ext.ENABLED and conv_taps_ext.supported(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
            # the stride-2 3x3 of a stage's first block and its 1x1 / stride-2 projection shortcut (csrc/conv_taps.hip), forward,
            # input gradient (four pixel-parity classes) and weight gradient (csrc/conv_wgrad.hip) by hand
            return conv_taps_ext.conv_strided(x, w, shift, relu=relu, hand_out_token=hand_out_token and relu)
        elif conv3x3_ext.ENABLED and conv3x3_ext.supported(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
            # stride-1 3x3: implicit GEMM with LDS im2col, shift and ReLU in its epilogue (csrc/conv3x3.hip); forward and
            # input gradient on the kernel, weight gradient with the library
            return conv3x3_ext.conv3x3(x, w, shift, relu=relu, hand_out_token=hand_out_token and relu, in_token=relu_token)
        elif relu and bias_act_ext.ENABLED and (x.is_cuda or bias_act_ext._backend is not None):
            # the shift and the ReLU in one pass behind the library convolution (csrc/bias_act.hip) instead of the
            # library's own bias kernel plus a clamp
            x = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
            if bias_act_ext.supported(x, b):
                return bias_act_ext.bias_act(x, b, None, relu=True)
            x = x + b.view(1, -1, 1, 1)
        else:
            x = F.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups)
    else:
        x = bn(conv(x))
    return F.relu(x, inplace=True) if relu else x


# (the identity-gradient fusion in Bottleneck.forward; False = the separate elementwise add, for tests)
_SKIP_FUSE = True
# conv2's input gradient applies the mask of the ReLU behind conv1 (mdetr_conv3x3_masked); MDETR_TUNE=conv2_mask=0: A-B runs
_CONV2_TAKES_MASK = _tune.get("conv2_mask", "1") != "0"


def _global_hooks():
    """Is a process-wide forward (pre-)hook registered?  (torch.nn.modules.module.register_module_forward_hook: it sees every block's tensors)"""
    m = torch.nn.modules.module
    return bool(getattr(m, "_global_forward_hooks", None)) or bool(getattr(m, "_global_forward_pre_hooks", None))


class Bottleneck(nn.Module):
    """1x1 reduce -> 3x3 (carries the stride) -> 1x1 expand (x4), residual add, ReLU."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        from . import linear
        # ReLU masks of the backward pass applied by the consuming kernel (linear.ReluToken): only where this module sees every
        # consumer of the tensor -- no hook may have been handed it
        premask = linear._PREMASK and torch.is_grad_enabled() and not (self._forward_hooks or self._forward_pre_hooks or _global_hooks())
        if self.downsample is None and _SKIP_FUSE:
            # (the identity's gradient meets conv1's inside its dgrad GEMM; so does, with a token from the previous block, its ReLU mask)
            y, skip = conv_bn(x, self.conv1, self.bn1, True, skip_out=True, relu_token=linear.relu_token_of(x) if premask else None,
                              hand_out_token=premask)             # (conv2 below is the one consumer of conv1's ReLU output)
        elif self.downsample is not None and _SKIP_FUSE and x.requires_grad and torch.is_grad_enabled():
            # a stage's first block: x feeds conv1 and the projection shortcut.  Read in a chain -- the shortcut continues from the x'
            # conv1 hands back -- the shortcut's input gradient is added inside conv1's input-gradient GEMM (one 30 - 60 MB sum less)
            y, x2 = conv_bn(x, self.conv1, self.bn1, True, skip_out=True)
            skip = conv_bn(x2, self.downsample[0], self.downsample[1], False)
        else:
            skip = x if self.downsample is None else conv_bn(x, self.downsample[0], self.downsample[1], False)
            y = conv_bn(x, self.conv1, self.bn1, True)
        y = conv_bn(y, self.conv2, self.bn2, True, hand_out_token=premask, relu_token=linear.relu_token_of(y) if (premask and _CONV2_TAKES_MASK) else None)
        pre = self.conv3.__dict__.get("_prefolded")
        if pre is None and not self.conv3.weight.requires_grad and isinstance(self.bn3, FrozenBatchNorm2d) and self.conv3.bias is None \
                and y.dtype in (torch.bfloat16, torch.float16) and not torch.is_autocast_enabled():
            pre = frozen_fold(self.conv3, self.bn3, y.dtype)     # (layer1: frozen, the folded weight is cached)
        if pre is not None and pre[0].dtype == y.dtype and pointwise_residual_relu_eligible(y, pre[0], pre[1], skip):
            # expansion + folded BN + "+ identity" + ReLU from ONE kernel (csrc/tgemm.hip's residual epilogue): the product is not
            # read back by an elementwise pass
            self.conv3.__dict__.pop("_prefolded", None)
            return pointwise_conv_residual_relu(y, pre[0], pre[1], skip, in_token=linear.relu_token_of(y) if premask else None,
                                                hand_out_token=premask and self.__dict__.get("feeds_next_block", False))
        y = conv_bn(y, self.conv3, self.bn3, False)
        if bias_act_ext.ENABLED and bias_act_ext.supported(y, None, skip):
            return bias_act_ext.bias_act(y, None, skip, relu=True)   # "+ identity" and the ReLU in one pass
        return F.relu(y + skip, inplace=True)


class ResNetBody(nn.Module):
    """Stem + layer1..4 of a bottleneck ResNet; returns the requested stages as an ordered dict
    {out_name: tensor} (what IntermediateLayerGetter does for the reference, backbone.py:82)."""
    BLOCKS = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3), 'resnet152': (3, 8, 36, 3)}

    def __init__(self, name, return_layers: Dict[str, str], dilation=False, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        if name not in self.BLOCKS:
            raise ValueError("backbone %r: number of channels are hard coded for bottleneck ResNets "
                             "(reference backbone.py:103)" % name)
        self.return_layers = dict(return_layers)
        self.prefold = None               # None = on CUDA tensors only; True / False to force (tests)
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        n = self.BLOCKS[name]
        self.layer1 = self._stage(64, n[0], 1, False, norm_layer)
        self.layer2 = self._stage(128, n[1], 2, False, norm_layer)
        self.layer3 = self._stage(256, n[2], 2, False, norm_layer)
        self.layer4 = self._stage(512, n[3], 2, dilation, norm_layer)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _stage(self, planes, blocks, stride, dilate, norm_layer):
        prev_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), norm_layer(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down, prev_dilation, norm_layer)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes, dilation=self.dilation, norm_layer=norm_layer) for _ in range(1, blocks)]
        for blk in layers[:-1]:
            blk.__dict__["feeds_next_block"] = True      # its output has ONE consumer: the next block of this Sequential (linear.ReluToken)
        return nn.Sequential(*layers)

    def _trainable_pairs(self):
        pairs = []
        for m in self.modules():
            if isinstance(m, Bottleneck):
                cands = [(m.conv1, m.bn1), (m.conv2, m.bn2), (m.conv3, m.bn3)]
                if m.downsample is not None:
                    cands.append((m.downsample[0], m.downsample[1]))
                pairs += [(c, b) for c, b in cands if c.weight.requires_grad and c.bias is None
                          and isinstance(b, FrozenBatchNorm2d)]
        return pairs

    def _stash_holders(self):
        hit = self.__dict__.get("_stash_holder_list")
        if hit is None:
            hit = self.__dict__["_stash_holder_list"] = [m for m in self.modules() if isinstance(m, nn.Conv2d)]
        return hit

    def forward(self, x):
        use = self.prefold if self.prefold is not None else x.is_cuda
        # a forward pass that was interrupted between `prefold` and `conv_bn` (an exception that the caller caught) leaves
        # per-pass folded weights parked on the convolutions; they must not survive into this pass -- least of all into a
        # no-grad / eval pass that folds nothing itself and would then run with weights from before the last optimizer step
        for m in self._stash_holders():
            m.__dict__.pop("_prefolded", None)
        if use and torch.is_grad_enabled() and not torch.is_autocast_enabled():
            dt = x.dtype if x.dtype in (torch.bfloat16, torch.float16) else None
            pairs = self._trainable_pairs()
            if pairs:
                prefold(pairs, dt if dt is not None else pairs[0][0].weight.dtype)
        x = conv_bn(x, self.conv1, self.bn1, True)
        if decimate_ext.ENABLED and decimate_ext.maxpool_supported(x):
            x = decimate_ext.maxpool3x3s2(x)            # frozen stem: no gradient flows here (csrc/decimate.hip)
        else:
            x = self.maxpool(x)
        out = {}
        for name in ('layer1', 'layer2', 'layer3', 'layer4'):
            x = getattr(self, name)(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class BackboneBase(nn.Module):
    def __init__(self, body: nn.Module, train_backbone: bool, return_interm_layers: bool):
        super().__init__()
        for name, p in body.named_parameters():             # stem + layer1 always frozen (:71-73)
            if not train_backbone or not any(s in name for s in ('layer2', 'layer3', 'layer4')):
                p.requires_grad_(False)
        if return_interm_layers:
            self.strides, self.num_channels = [8, 16, 32], [512, 1024, 2048]
        else:
            self.strides, self.num_channels = [32], [2048]
        self.body = body
        self._masks = {}

    def forward(self, images):
        out = {}
        for name, x in self.body(images).items():
            # all-False padding mask (:88), tagged so that consumers need not inspect it; one tensor per shape
            key = (x.shape[0], x.shape[2], x.shape[3], x.device)
            mask = self._masks.get(key)
            if mask is None:
                mask = self._masks[key] = mark_no_padding(torch.zeros(key[:3], dtype=torch.bool, device=x.device))
            out[name] = NestedTensor(x, mask)
        return out


class Backbone(BackboneBase):
    """ResNet backbone with frozen BatchNorm."""

    def __init__(self, name: str, train_backbone: bool, return_interm_layers: bool, dilation: bool, weights=None):
        layers = {"layer2": "0", "layer3": "1", "layer4": "2"} if return_interm_layers else {"layer4": "0"}
        body = ResNetBody(name, layers, dilation)
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            missing, unexpected = body.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc.")}, strict=False)
            if missing:
                raise RuntimeError("backbone weights are missing keys: %s" % missing[:5])
        super().__init__(body, train_backbone, return_interm_layers)
        if dilation:
            self.strides[-1] = self.strides[-1] // 2


class Joiner(nn.Sequential):
    """(backbone, position_embedding) -> (list of NestedTensor features, list of position encodings)."""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def _as(self, x, p, dt):
        """p in dtype dt.  A constant (sine) encoding of an unpadded map comes out of its per-shape cache as the same tensor every
        iteration: its cast is kept beside it, per shape and for good (three casts of up to 31 MB per iteration otherwise).  Never
        evicted: a captured hipGraph that has read an entry reads that address in every replay (an eager iteration on another
        batch shape in between must not free it -- tests/test_trainer_gpu.py poisons freed memory to show exactly this)."""
        if p.dtype == dt:
            return p
        from ..utils.misc import no_padding
        from .position_encoding import PositionEmbeddingSine
        if p.requires_grad or p.grad_fn is not None or not no_padding(x.mask) or not isinstance(self[1], PositionEmbeddingSine):
            # (a learned embedding returns a fresh tensor per call -- under no_grad nothing above tells it from a constant, and
            #  a chain of superseded casts would grow by one map per level and evaluation iteration)
            return p.to(dt)
        casts = self.__dict__.setdefault("_pos_cast", {})
        key = (tuple(p.shape), p.device, p.dtype, dt)
        hit = casts.get(key)
        if hit is None or hit[0] is not p or hit[1] != p._version:
            hit = casts[key] = (p, p._version, p.to(dt), hit)          # (a superseded entry stays referenced: see above)
        return hit[2]

    def forward(self, images):
        feats = self[0](images)
        out: List[NestedTensor] = [feats[k] for k in sorted(feats)]
        pos = [self._as(x, self[1](x), x.tensors.dtype) for x in out]
        return out, pos


def build_backbone(cfg):
    position_embedding = build_position_encoding(cfg)
    return_interm_layers = cfg['masks'] or cfg['num_feature_levels'] > 1
    backbone = Backbone(cfg['backbone'], cfg['train_backbone'], return_interm_layers, cfg['dilation'],
                        weights=cfg.get('backbone_weights'))
    return Joiner(backbone, position_embedding)
