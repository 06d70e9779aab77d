"""Timing of the device image path of the input pipeline (csrc/kitti_prep.hip): kernel time for one batch of KITTI-sized
images against its HBM roofline, and images/s including the H2D copy of the decoded images.

    python -m monodetr_amd.tools.prepbench [--batch 8] [--iters 50] [--dtype fp32|bf16]

Synthetic images (random bytes, the four KITTI sizes), every stage of the distortion chain switched on, a mild crop.
Algorithmic bytes per image = H*W*3 read + 3*384*1280*e written."""
import argparse
import json

import numpy as np
import torch

from monodetr_amd import kitti_prep_ext as prep
from monodetr_amd.helpers.dataloader_helper import pack_images

SIZES = [(1242, 375), (1224, 370), (1238, 374), (1241, 376)]
HBM_PEAK = 8.0e12


def make_batch(n, seed=0):
    rs = np.random.RandomState(seed)
    images = []
    for k in range(n):
        w, h = SIZES[k % 4]
        d = np.zeros(1, dtype=prep.DESCRIPTOR)
        d['width'], d['height'] = w, h
        d['flags'] = 127 if k % 2 else 127 & ~prep.FLIP
        d['perm'] = [0x24, 0x18, 0x06][k % 3]
        d['brightness'], d['contrast'], d['saturation'], d['hue'] = rs.uniform(-32, 32), rs.uniform(0.5, 1.5), rs.uniform(0.5, 1.5), rs.uniform(-18, 18)
        s = rs.uniform(0.95, 1.05)
        d['inv'] = [s * w / 1280.0, 0, rs.uniform(-10, 10), 0, s * w / 1280.0, rs.uniform(-5, 5) + (h - s * w * 0.3) / 2]
        images.append({'pixels': rs.randint(0, 256, size=(h, w, 3), dtype=np.uint8), 'descriptor': d})
    return images


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dtype", default="fp32")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    images = make_batch(a.batch)
    packed, n = pack_images(images)
    head = n * prep.DESCRIPTOR.itemsize
    staged = packed.pin_memory()
    dev = staged.cuda()
    out = torch.empty((n, 3, 384, 1280), dtype=dt, device="cuda", memory_format=torch.channels_last)

    def kernel():
        prep.preprocess_batch(dev[head:], dev[:head], dtype=dt, out=out, channels_last=True)

    def with_copy():
        d = staged.to("cuda", non_blocking=True)
        prep.preprocess_batch(d[head:], d[:head], dtype=dt, out=out, channels_last=True)

    res = {}
    for tag, fn in (("kernel", kernel), ("h2d+kernel", with_copy)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        e1.synchronize()
        res[tag] = e0.elapsed_time(e1) / a.iters
    bytes_alg = packed.numel() - head + out.numel() * out.element_size()
    print(json.dumps({
        "batch": n, "dtype": a.dtype, "kernel_ms": round(res["kernel"], 4), "h2d_kernel_ms": round(res["h2d+kernel"], 4),
        "images_per_s_kernel": round(n / res["kernel"] * 1e3, 1), "images_per_s_with_h2d": round(n / res["h2d+kernel"] * 1e3, 1),
        "roofline": {"bound": "hbm", "achieved": round(bytes_alg / res["kernel"] / 1e6, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(bytes_alg / (res["kernel"] * 1e-3) / HBM_PEAK, 4), "algorithmic_bytes": bytes_alg}}))


if __name__ == "__main__":
    main()
