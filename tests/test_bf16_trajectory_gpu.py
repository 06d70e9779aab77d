"""The headline precision against the reference's arithmetic over a TRAINING RUN, not a single step.

tests/test_model_gpu.py::test_bf16_body_outputs_and_gradients_vs_fp32 compares one forward / backward pass of the bf16 model body with
the fp32 model: whole-model gradient cosine >= 0.995, but the decoder's `cross_attn.sampling_offsets` gradients sit at 0.85-0.98 --
differences of neighbouring bf16 value rows.  What that costs is a question about the optimisation, so this test answers it there: 200
iterations of the committed bf16 step (bench.COMMITTED_SWITCHES["bf16"]: bf16 body, bf16-native MSDA, every hand-written kernel
family) and 200 of the fp32 step (the reference's arithmetic with the fp32 families), same initial weights, same cycle of 8 synthetic
batches, dropout off in both (its random streams differ by construction), same AdamW.

ACCEPTED DRIFT (the bar, with its reason): bf16 carries 8 mantissa bits, one rounding is 2^-9 = 0.2 % relative; the 26 loss terms are
means over hundreds of such values and the step integrates them, so the trajectories must agree far better than a single rounding
error would suggest if the roundings are unbiased, and must NOT separate with the iteration count if nothing systematic is wrong:
    * the smoothed total loss (window 20) stays within 2 % of the fp32 run's everywhere after the first window,
    * the mean over the last 50 iterations within 1 %,
    * and the bf16 run LEARNS: its last-50 mean is below 80 % of its first-10 mean, as the fp32 run's is.
Measured on MI355X (profiles/r04p_bf16_trajectory.json): worst smoothed deviation 0.68 %, last-50 means 35.458 (bf16) vs 35.471 (fp32)
= 0.04 %, both runs 70.5-70.9 -> 35.5: the bars leave a factor of three over the measurement, not thirty."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import monodetr_amd._runtime_env  # noqa: E402,F401  (before torch)

import pytest  # noqa: E402
import torch  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(precision, steps, size):
    import bench
    from model_init import disable_dropout_
    dev = torch.device("cuda", 0)
    step = bench.TrainStep(dev, 2, precision, size=size, switches=bench.committed_switches(precision)[0], seed=444)
    disable_dropout_(step.raw_model)
    batches = [step.make_inputs(9000 + i) for i in range(8)]
    out = []
    for i in range(steps):
        step.inputs = batches[i % 8]
        out.append(step())
        if i % 25 == 24:
            torch.cuda.synchronize()
    losses = [float(x.detach()) for x in out]
    del step
    torch.cuda.empty_cache()
    return losses


def _smooth(x, w):
    return [sum(x[i:i + w]) / w for i in range(0, len(x) - w + 1)]


def test_bf16_training_trajectory_follows_the_fp32_one():
    steps, size = 200, (192, 640)
    fp32 = _run("fp32", steps, size)
    bf16 = _run("bf16", steps, size)
    assert all(x == x and abs(x) < 1e6 for x in fp32 + bf16)
    s32, s16 = _smooth(fp32, 20), _smooth(bf16, 20)
    worst = max(abs(a - b) / abs(a) for a, b in zip(s32[20:], s16[20:]))
    tail32, tail16 = sum(fp32[-50:]) / 50, sum(bf16[-50:]) / 50
    head32, head16 = sum(fp32[:10]) / 10, sum(bf16[:10]) / 10
    rec = {"steps": steps, "size": list(size), "worst_smoothed_rel": round(worst, 4), "tail_rel": round(abs(tail16 - tail32) / tail32, 4),
           "fp32_first10_last50": [round(head32, 3), round(tail32, 3)], "bf16_first10_last50": [round(head16, 3), round(tail16, 3)],
           "fp32_every20": [round(x, 2) for x in s32[::20]], "bf16_every20": [round(x, 2) for x in s16[::20]]}
    print("BF16-TRAJECTORY " + json.dumps(rec))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(rec, open(os.path.join(out, "bf16_trajectory.json"), "w"))
    assert tail32 < 0.8 * head32 and tail16 < 0.8 * head16, rec          # both runs learn
    assert worst < 0.02, rec
    assert abs(tail16 - tail32) / tail32 < 0.01, rec
