"""Developer tool: wall-clock breakdown of one training step (forward / criterion / backward / optimizer),
each phase bracketed by torch.cuda.synchronize().  python -m monodetr_amd.tools.stepbreakdown [--precision bf16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    step = bench.TrainStep(dev, 8, a.precision)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    acc = dict(forward=0.0, criterion=0.0, backward=0.0, optimizer=0.0, criterion_host_only=0.0)
    images, calibs, img_sizes, targets = step.inputs
    for _ in range(a.iters):
        step.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.precision == "bf16-autocast"):
            out = step.model(images, calibs, targets, img_sizes)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            losses = step.criterion(out, targets)
            t1b = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        w = step.criterion.weight_dict
        total = step.criterion.weighted_total(losses)
        total.backward()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        step.optimizer.step()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        acc["forward"] += t1 - t0; acc["criterion"] += t2 - t1; acc["backward"] += t3 - t2; acc["optimizer"] += t4 - t3
        acc["criterion_host_only"] += t1b - t1
    print(json.dumps({k: round(v / a.iters * 1e3, 2) for k, v in acc.items()}))
    # host enqueue time of each phase (no synchronisation inside the step): if their sum is close to the
    # step's wall time the step is launch-bound, not GPU-bound
    host = dict(forward=0.0, criterion=0.0, backward=0.0, optimizer=0.0, wall=0.0)
    for _ in range(a.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.precision == "bf16-autocast"):
            out = step.model(images, calibs, targets, img_sizes)
            t1 = time.perf_counter()
            losses = step.criterion(out, targets)
        t2 = time.perf_counter()
        total = step.criterion.weighted_total(losses)
        total.backward()
        t3 = time.perf_counter()
        step.optimizer.step()
        t4 = time.perf_counter()
        torch.cuda.synchronize(); t5 = time.perf_counter()
        host["forward"] += t1 - t0; host["criterion"] += t2 - t1; host["backward"] += t3 - t2
        host["optimizer"] += t4 - t3; host["wall"] += t5 - t0
    print(json.dumps({"host_enqueue_ms": {k: round(v / a.iters * 1e3, 2) for k, v in host.items()}}))


if __name__ == "__main__":
    main()
