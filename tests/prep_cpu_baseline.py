"""CPU timing of the reference's image chain (oracle restatement: numpy distortion, PIL flip + affine warp,
normalisation) per image, the figure the device path of the input pipeline is reported beside.

    python tests/prep_cpu_baseline.py [--images 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import kitti_synth                                   # noqa: E402
from oracle import kitti_pipeline as okp             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    a = ap.parse_args()
    rs = np.random.RandomState(0)
    imgs = [kitti_synth.synth_image(rs, *kitti_synth.SIZES[k % 4]) for k in range(a.images)]
    np.random.seed(1)
    t0 = time.perf_counter()
    for img in imgs:
        src = okp.apply_photometric(img, okp.draw_photometric())
        size = np.array([img.shape[1], img.shape[0]])
        flip, center, crop_size, _ = okp.draw_geometry(size)
        okp.warp_and_normalise(src, flip, okp.affine_pair(center, crop_size)[1])
    dt = (time.perf_counter() - t0) / a.images
    print(json.dumps({"cpu_chain_ms_per_image": round(dt * 1e3, 2), "images_per_s_per_core": round(1 / dt, 1), "kind": "port",
                      "sample": "%d synthetic KITTI-sized images, full distortion chain, 1 thread" % a.images}))


if __name__ == "__main__":
    main()
