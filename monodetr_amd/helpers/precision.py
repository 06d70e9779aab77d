"""bf16 execution of the model body with fp32 heads and fp32 master weights (BASELINE configs[2]).

``to_bf16_body(model)`` stores the parameters of the token-heavy part of the network (input
projections, depth predictor, encoder, decoder layers) in bf16, so every activation there is bf16 and
no per-op autocast casts are launched (under ``torch.autocast`` one training step spends ~860 kernel
launches on fp32<->bf16 copies).  What stays fp32:
  * the prediction heads, reference points and all box/depth arithmetic after them (coordinates in
    bf16 would be quantised to ~1/256 of the image),
  * the backbone's parameters (its frozen-BN fold multiplies them by fp32 scales; the folded weight
    is cast once per step) and all FrozenBatchNorm buffers,
  * ``depth_bin_values``, the criterion, and the optimizer state: ``AdamW`` keeps an fp32 master copy
    of every bf16 parameter and writes the rounded result back after each step.
The MSDA operator accumulates in fp32 regardless (inputs are widened inside the op).
"""
import torch

BODY_PREFIXES = ("input_proj.", "depth_predictor.", "depthaware_transformer.encoder.",
                 "depthaware_transformer.decoder.layers.", "depthaware_transformer.level_embed")
KEEP_FP32 = ("depth_predictor.depth_bin_values",)


def to_bf16_body(model):
    n = 0
    for name, p in model.named_parameters():
        if name.startswith(BODY_PREFIXES) and name not in KEEP_FP32 and p.dtype == torch.float32:
            p.data = p.data.to(torch.bfloat16)
            n += 1
    model._mdetr_bf16_body = True
    return n
