"""Diagnostic (GPU): does the training step's MSDA backward use the `far` side buffer (corners beyond every block's reach)?  Runs a few
eager iterations of bench.TrainStep with a hook on the backward entry that reads the workspace header's `far` word after each call."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import monodetr_amd._runtime_env  # noqa: E402,F401
import torch  # noqa: E402

import bench  # noqa: E402
from monodetr_amd import msda_ext  # noqa: E402
from monodetr_amd.kernel_families import COMMITTED_SWITCHES  # noqa: E402

seen = []
orig = msda_ext._workspace


def spy(device, nbytes):
    ws = orig(device, nbytes)
    seen.append(ws)
    return ws


msda_ext._workspace = spy
step = bench.TrainStep(torch.device("cuda", 0), 8, "bf16", switches=COMMITTED_SWITCHES["bf16"], graph=False)
for it in range(14):
    seen.clear()
    step()
    torch.cuda.synchronize()
    flags = [int(ws.view(torch.int32)[2].item()) for ws in seen]          # Header: absmax_g, absmax_a, far
    print("iteration", it, "far flags of the", len(flags), "backward calls:", flags)
