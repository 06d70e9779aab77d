"""Where does a small grouped fp32 product spend its time?  Variations of the heads' one-slab input-gradient launch (B1: [4400, 6] x
[6, 256], masked), run under `rocprofv3 --kernel-trace`; scripts/exp/sgemm_probe.sh prints the per-launch durations in order."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import monodetr_amd._runtime_env  # noqa
import torch
from monodetr_amd import sgemm_ext as ext

dev = "cuda"
torch.manual_seed(0)


def run(tag, mode, probs):
    for _ in range(3):
        ext.grouped(mode, probs)
    torch.cuda.synchronize()
    print("CASE", tag, flush=True)


T = 4400
g6, w6, saved = torch.randn(T, 6, device=dev), torch.randn(6, 256, device=dev), torch.randn(T, 256, device=dev)
g32, w32 = torch.randn(T, 32, device=dev), torch.randn(32, 256, device=dev)
out = torch.empty(T, 256, device=dev)
P = ext.Problem
run("nn_k6_masked_4400", ext.NN, [P([(g6, w6)], out, mask=saved)])
run("nn_k6_plain_4400", ext.NN, [P([(g6, w6)], out)])
run("nn_k32_plain_4400(fast)", ext.NN, [P([(g32, w32)], out)])
run("nn_k32_plain_64rows(1 tile row)", ext.NN, [P([(g32[:64], w32)], out[:64])])
x, w = torch.randn(T, 256, device=dev), torch.randn(256, 256, device=dev)
run("nt_k256_n256_4400(fast, 8 slabs)", ext.NT, [P([(x, w)], out)])
run("nt_k256_n256_x4(L1-like)", ext.NT, [P([(x, w)], torch.empty(T, 256, device=dev)) for _ in range(4)])
xk = torch.randn(T, 1024, device=dev)
wk = torch.randn(1024, 256, device=dev)
run("nn_k1024_n256_4400(fast, 32 slabs)", ext.NN, [P([(xk, wk)], out)])
