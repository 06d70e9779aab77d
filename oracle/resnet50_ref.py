"""Independent restatement of the ResNet-50 body the reference takes from torchvision -- TEST INFRASTRUCTURE ONLY.

The reference builds its backbone with ``torchvision.models.resnet50(replace_stride_with_dilation=[False, False, False],
norm_layer=FrozenBatchNorm2d)`` wrapped in ``IntermediateLayerGetter`` (lib/models/monodetr/backbone.py:82, :100-102) and
torchvision is not vendored in the reference tree ("parity unpinned" at that boundary).  This file restates
the published torchvision definition -- the v1.5 bottleneck network: 7x7/2 stem, 3x3/2 max-pool, stages of (3, 4, 6, 3)
bottlenecks with 64/128/256/512 planes and expansion 4, the stage's stride on the 3x3 convolution, a 1x1 strided
projection on the first block of each stage -- as a pure FUNCTION of a state_dict with the reference's key names, using
nothing but ``F.conv2d``, the frozen-BN affine of backbone.py:54-64 (``w * rsqrt(running_var + 1e-5)``), ``F.max_pool2d``
and ``F.relu``.  It shares no code with monodetr_amd/monodetr/backbone.py (no folding, no GEMM forms, no module classes),
so agreement between the two pins the product's backbone values to something other than itself.

Pinning of THIS file: torchvision is installed nowhere in the build container, so its code cannot be run.  The restatement is
held instead to an independent public implementation of the same network that IS present -- Hugging Face transformers 5.15.0,
``transformers.models.resnet.modeling_resnet`` (what the ``microsoft/resnet-50`` checkpoint, converted from the torchvision /
timm weights, runs on): tests/test_backbone_parity_hf_cpu.py loads one state_dict into both and compares the three feature maps
and the gradients of the input and of all 53 convolutions (1e-10 / 1e-9 in fp64).  torchvision's own code path stays unpinned.
"""
import torch
import torch.nn.functional as F

STAGES = ((1, 64, 3, 1), (2, 128, 4, 2), (3, 256, 6, 2), (4, 512, 3, 2))      # (index, planes, blocks, stride)


def frozen_bn(x, sd, prefix, eps=1e-5):
    """FrozenBatchNorm2d.forward of the reference (backbone.py:54-64): per-channel affine from the running statistics."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w * (rv + eps).rsqrt()
    shift = b - rm * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def bottleneck(x, sd, prefix, stride, project):
    out = F.relu(frozen_bn(F.conv2d(x, sd[prefix + ".conv1.weight"]), sd, prefix + ".bn1"))
    out = F.relu(frozen_bn(F.conv2d(out, sd[prefix + ".conv2.weight"], stride=stride, padding=1), sd, prefix + ".bn2"))
    out = frozen_bn(F.conv2d(out, sd[prefix + ".conv3.weight"]), sd, prefix + ".bn3")
    if project:
        x = frozen_bn(F.conv2d(x, sd[prefix + ".downsample.0.weight"], stride=stride), sd, prefix + ".downsample.1")
    return F.relu(out + x)


def resnet50_features(images, sd, prefix=""):
    """images [B, 3, H, W] -> (layer2, layer3, layer4) feature maps (strides 8, 16, 32; 512 / 1024 / 2048 channels).
    ``sd`` maps ``<prefix>conv1.weight``, ``<prefix>bn1.*``, ``<prefix>layerN.K.*`` to tensors (the reference's
    ``backbone.0.body.`` state_dict entries)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = F.relu(frozen_bn(F.conv2d(images, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = []
    for index, planes, blocks, stride in STAGES:
        for k in range(blocks):
            x = bottleneck(x, sd, "layer%d.%d" % (index, k), stride if k == 0 else 1, k == 0)
        if index >= 2:
            feats.append(x)
    return tuple(feats)
