"""Synthetic detections for the evaluation tests: the ground-truth objects of a synthetic KITTI tree
(tests/kitti_synth.py), jittered, some dropped, some duplicated, plus false positives -- written in the KITTI result
format (16 fields) the evaluation reads."""
import os

import numpy as np


def make_results(root, ids, out_dir, seed=3):
    rs = np.random.RandomState(seed)
    os.makedirs(out_dir, exist_ok=True)
    for idx in ids:
        lines = [l.strip().split(' ') for l in open(os.path.join(root, 'training/label_2/%s.txt' % idx)).readlines() if l.strip()]
        out = []
        for f in lines:
            if f[0] == 'DontCare' or rs.rand() < 0.2:
                continue
            v = [float(x) for x in f[1:]]
            reps = 2 if rs.rand() < 0.15 else 1                       # duplicate detections of one object
            for _ in range(reps):
                q = rs.choice([0.02, 0.08, 0.3])                       # localisation quality
                box = [v[3] + rs.normal(0, 3 * q * 10), v[4] + rs.normal(0, 2 * q * 10), v[5] + rs.normal(0, 3 * q * 10), v[6] + rs.normal(0, 2 * q * 10)]
                dims = [v[7] * (1 + rs.normal(0, q)), v[8] * (1 + rs.normal(0, q)), v[9] * (1 + rs.normal(0, q))]
                loc = [v[10] + rs.normal(0, q * 3), v[11] + rs.normal(0, q), v[12] + rs.normal(0, q * 6)]
                ry = v[13] + rs.normal(0, q * 2)
                alpha = v[2] + rs.normal(0, q * 2)
                name = f[0] if rs.rand() > 0.05 else 'Car'
                out.append('%s 0.0 0 %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f' % (
                    name, alpha, box[0], box[1], box[2], box[3], dims[0], dims[1], dims[2], loc[0], loc[1], loc[2], ry,
                    float(np.clip(1 - q * 2 + rs.normal(0, 0.1), 0.01, 0.99))))
        for _ in range(rs.randint(0, 4)):                              # false positives
            z = rs.uniform(5, 60)
            x = rs.uniform(-0.4, 0.4) * z
            u, vv = 609 + 721 * x / z, 180 + rs.uniform(-20, 40)
            w, h = rs.uniform(20, 120), rs.uniform(25, 90)
            out.append('%s 0.0 0 %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f' % (
                rs.choice(['Car', 'Car', 'Pedestrian', 'Cyclist']), rs.uniform(-3, 3), u - w / 2, vv - h / 2, u + w / 2, vv + h / 2,
                rs.uniform(1.3, 1.9), rs.uniform(1.4, 1.8), rs.uniform(3, 4.5), x, rs.uniform(1.3, 1.9), z, rs.uniform(-3, 3), rs.uniform(0.05, 0.7)))
        with open(os.path.join(out_dir, '%s.txt' % idx), 'w') as fh:
            fh.write('\n'.join(out) + ('\n' if out else ''))


def decode_problem():
    """Random head outputs of two images + their calibration, shared with the tests."""
    import torch
    g = torch.Generator().manual_seed(11)
    B, Q = 2, 50
    boxes = torch.cat([torch.rand(B, Q, 2, generator=g) * 0.6 + 0.2, torch.rand(B, Q, 4, generator=g) * 0.08 + 0.01], -1)
    outputs = {'pred_logits': torch.randn(B, Q, 3, generator=g) * 1.5 - 2.5, 'pred_boxes': boxes,
               'pred_angle': torch.randn(B, Q, 24, generator=g), 'pred_3d_dim': torch.rand(B, Q, 3, generator=g) * 2 + 1,
               'pred_depth': torch.cat([torch.rand(B, Q, 1, generator=g) * 50 + 3, torch.randn(B, Q, 1, generator=g)], -1)}
    p2 = [np.array([[721.54, 0, 609.56, 44.857], [0, 721.54, 172.85, 0.2164], [0, 0, 1, 0.002746]], dtype=np.float32),
          np.array([[707.05, 0, 604.08, 45.757], [0, 707.05, 180.51, -0.3454], [0, 0, 1, 0.004981]], dtype=np.float32)]
    info = {'img_id': np.array([1, 4]), 'img_size': np.array([[1242, 375], [1224, 370]])}
    return outputs, p2, info
