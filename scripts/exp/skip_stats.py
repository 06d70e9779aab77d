"""Experiment: at which residual sites does _TokenLinearSkip find the arriving gradient exclusively its own?"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import monodetr_amd  # noqa: F401,E402
import torch  # noqa: E402
import bench  # noqa: E402
from monodetr_amd.monodetr import linear  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
chosen, _ = bench.committed_switches("bf16")
step = bench.TrainStep(dev, 8, "bf16", graph=False, switches=chosen)
step()
linear.SKIP_STATS = []
step()
torch.cuda.synchronize()
print("baseline", linear._BASELINE)
for shape, counts, ok in linear.SKIP_STATS:
    print(shape, "holders", counts, "-> in place" if ok else "-> copy")
