"""Developer tool: what one ``FlatGradSync.sync()`` costs on this GPU with a ONE-rank RCCL process group (the N > 1 code path
as far as a 1-GPU box can run it): host time, GPU time, and the kernels behind it.

    python -m monodetr_amd.tools.syncbench [--iters 20]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import bench
    from monodetr_amd.helpers.dist_helper import FlatGradSync
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    step = bench.TrainStep(dev, 2, "bf16", switches=bench.committed_switches("bf16")[0])
    step._step()                                              # real gradients of the real model
    sync = FlatGradSync(step.raw_model.parameters())
    grads = [p.grad for p in sync.params if p.grad is not None]
    print("parameters with gradients: %d, %.1f MB" % (len(grads), sum(g.numel() * g.element_size() for g in grads) / 1e6))
    for _ in range(3):
        sync.sync()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.iters):
        sync.sync()
    e1.record()
    host = (time.perf_counter() - t0) / a.iters * 1e3
    torch.cuda.synchronize()
    print("sync(): host %.3f ms per call (enqueue), GPU %.3f ms per call" % (host, e0.elapsed_time(e1) / a.iters))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            sync.sync()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60)[:6000])
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
