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


def _exported_at_start(name):
    """The variable's value in the environment the PROCESS STARTED with (/proc/self/environ), not in os.environ, which Python code
    may have edited after the HIP runtime read its flags."""
    try:
        with open("/proc/self/environ", "rb") as f:
            for item in f.read().split(b"\0"):
                if item.startswith(name.encode() + b"="):
                    return item.split(b"=", 1)[1].decode(errors="replace")
    except OSError:
        pass
    return None


_at_start = _exported_at_start(GRAPH_PACKET_ENV)
_set_here = False
if GRAPH_PACKET_ENV not in os.environ and not _torch_was_loaded:
    os.environ[GRAPH_PACKET_ENV] = "0"
    _set_here = True
# IPC handles of RCCL / shared CUDA tensors need dmabuf mode on this stack (bench.py sets the same)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def graph_packets_off():
    """Was the packet path switched off EARLY ENOUGH to be in effect?  Either the process started with the variable exported as 0,
    or this module set it before torch (hence libamdhip64) was loaded.  A value that appeared in os.environ any other way -- set by
    Python code after `import torch` -- does not count: graph replay then stays off (helpers/step_helper.TrainIteration: eager
    launches and one log line)."""
    if os.environ.get(GRAPH_PACKET_ENV) != "0":
        return False
    return _at_start == "0" or (_set_here and not _torch_was_loaded)
