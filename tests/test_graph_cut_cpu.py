"""monodetr/_cut.py on the CPU: the backward pass of a training iteration in two calls, cut at the encoder's last MSDA launch
(site "msda": the operator's output, the residual stream next to it, the pyramid levels the depth predictor reads).  Every
parameter ends up with bit-for-bit the gradient of the uncut pass, and the first call stays above the cut.  (The GPU twin with the
fused self-attention path, and the two-graph replay built on it, are in tests/test_trainer_gpu.py.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_two_call_backward_pass_cut_at_the_last_encoder_layer_equals_the_uncut_one():
    import bench
    from model_init import disable_dropout_
    from monodetr_amd.monodetr import _cut
    from monodetr_amd.monodetr.ops.functions import ms_deform_attn_func as F_
    from oracle import msda_oracle
    before = F_.MSDA
    F_.MSDA = msda_oracle.OracleMSDA                                 # test-only CPU backend of the operator
    try:
        step = bench.TrainStep(torch.device("cpu"), 1, "fp32", size=(64, 224), switches=())
        disable_dropout_(step.raw_model)
        step._forward_backward(step.inputs)
        want = {n: p.grad.clone() for n, p in step.raw_model.named_parameters() if p.grad is not None}
        step._forward_backward(step.inputs, cut="msda")
        assert not _cut.active("msda")                               # the sites are live during the forward pass only
        cuts = [tuple(td.shape) for _, td in step._boundary]
        tokens = sum(h * w for h, w in ((8, 28), (4, 14), (2, 7), (1, 4)))
        assert cuts.count((1, tokens, 256)) == 2 and sum(1 for c in cuts if len(c) == 4) == 4, cuts
        first = {n for n, p in step.raw_model.named_parameters() if p.grad is not None}
        above = ("backbone.", "input_proj.", "depthaware_transformer.encoder.layers.0.", "depthaware_transformer.encoder.layers.1.",
                 "depthaware_transformer.level_embed")
        assert not [n for n in first if n.startswith(above)] and any(n.startswith("depthaware_transformer.decoder.") for n in first)
        # the last layer's own value / offset projections sit BELOW the operator: second call
        assert not [n for n in first if ".encoder.layers.2.self_attn.value_proj" in n or ".encoder.layers.2.self_attn.sampling_offsets" in n]
        step._backward_backbone()
        got = {n: p.grad for n, p in step.raw_model.named_parameters() if p.grad is not None}
        assert set(got) == set(want)
        for n in want:
            assert torch.equal(got[n], want[n]), n
    finally:
        F_.MSDA = before
