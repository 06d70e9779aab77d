"""Timing of the KITTI evaluation at validation-split size (3 769 frames) on synthetic annotations: the two
segmented overlap launches (bird's-eye view, 3-D) and the whole official evaluation of one class.

    python -m monodetr_amd.tools.evalbench [--frames 3769]
"""
import argparse
import json
import time

import numpy as np
import torch

from monodetr_amd.datasets.kitti.kitti_eval_python import eval as kitti_eval
from monodetr_amd.datasets.kitti.kitti_eval_python import rotate_iou


def synth_annos(n_frames, seed=0):
    rs = np.random.RandomState(seed)
    gts, dts = [], []
    for _ in range(n_frames):
        def anno(m, scores):
            loc = np.stack([rs.uniform(-15, 15, m), rs.uniform(1.2, 2.0, m), rs.uniform(5, 60, m)], 1)
            dims = np.stack([rs.uniform(3, 4.6, m), rs.uniform(1.3, 1.9, m), rs.uniform(1.4, 1.9, m)], 1)
            u = 609 + 721 * loc[:, 0] / loc[:, 2]
            h = 721 * dims[:, 1] / loc[:, 2]
            bbox = np.stack([u - h, 175 - h / 2, u + h, 175 + h / 2], 1)
            return {'name': np.array(['Car'] * m), 'truncated': np.zeros(m), 'occluded': rs.randint(0, 3, m), 'alpha': rs.uniform(-3, 3, m),
                    'bbox': bbox, 'dimensions': dims, 'location': loc, 'rotation_y': rs.uniform(-3, 3, m),
                    'score': rs.uniform(0.2, 1, m) if scores else np.zeros(m)}
        g = anno(rs.randint(1, 12), False)
        d = anno(rs.randint(1, 20), True)
        k = min(len(g['name']), len(d['name']))
        for key in ('location', 'dimensions', 'bbox'):                   # the first k detections sit near ground truths
            d[key][:k] = g[key][:k] + rs.normal(0, 0.15, g[key][:k].shape)
        d['rotation_y'][:k] = g['rotation_y'][:k] + rs.normal(0, 0.1, k)
        gts.append(g); dts.append(d)
    return gts, dts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3769)
    a = ap.parse_args()
    gt, dt = synth_annos(a.frames)
    bev = lambda an: np.concatenate([an['location'][:, [0, 2]], an['dimensions'][:, [0, 2]], an['rotation_y'][:, None]], 1)
    d3 = lambda an: np.concatenate([an['location'], an['dimensions'], an['rotation_y'][:, None]], 1)
    pairs = sum(len(g['name']) * len(d['name']) for g, d in zip(gt, dt))
    res = {"frames": a.frames, "pairs": pairs}
    for tag, fn, conv in (("bev", rotate_iou.segmented_rotate_iou, bev), ("3d", rotate_iou.segmented_box3d_overlap, d3)):
        b, q = [conv(x) for x in dt], [conv(x) for x in gt]
        fn(b, q)                                                          # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(b, q)
        torch.cuda.synchronize()
        res[tag + "_overlaps_ms_incl_transfers"] = round((time.perf_counter() - t0) * 1e3, 2)
    t0 = time.perf_counter()
    text, ret, moderate = kitti_eval.get_official_eval_result(gt, dt, 0)
    res["official_eval_one_class_s"] = round(time.perf_counter() - t0, 2)
    res["car_3d_moderate_R40"] = round(float(moderate), 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
