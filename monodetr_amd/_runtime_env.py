"""Process-level settings of the ROCm runtime this package relies on.  Import BEFORE ``torch`` (bench.py, tools/train_val.py and
tests/conftest.py do): the HIP runtime reads its flags when libamdhip64 is loaded.

``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0``: ROCm 7's hipGraph launch path that replays pre-recorded AQL packets corrupted the replayed
training iteration whenever eagerly launched kernels ran on the launching stream between two replays -- non-finite gradients at
FIXED positions of a few tensors (an eighth of a convolution's weight gradient, single elements of bias gradients), the losses
intact; with the flag off, or with AMD_SERIALIZE_KERNEL / AMD_SERIALIZE_COPY = 3, twelve such iterations are clean
(tests/diag/graph_nan.py, profiles/r03_graph_replay_corruption.md).  ``helpers/step_helper.TrainIteration`` additionally launches
its graphs on their own stream.  Export the variable as 1 to get the packet path back."""
import os
import sys

GRAPH_PACKET_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_torch_was_loaded = "torch" in sys.modules
if GRAPH_PACKET_ENV not in os.environ and not _torch_was_loaded:
    os.environ[GRAPH_PACKET_ENV] = "0"
# IPC handles of RCCL / shared CUDA tensors need dmabuf mode on this stack (bench.py sets the same)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def graph_packets_off():
    """Was the packet path switched off early enough to be in effect?  (Set by us before torch was imported, or already in the
    environment -- then we trust that it was exported before the process started.)"""
    return os.environ.get(GRAPH_PACKET_ENV) == "0"
