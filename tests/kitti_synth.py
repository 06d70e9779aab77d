"""A tiny synthetic KITTI tree (image_2 PNGs, label_2, calib, ImageSets) generated deterministically from a seed, so
that pipeline tests need no dataset and golden fixtures need to store only outputs.  Image sizes, label statistics and
calibration values follow the real files' ranges (sizes 1242x375 / 1224x370 / 1238x374 / 1241x376)."""
import os

import numpy as np
from PIL import Image

SIZES = [(1242, 375), (1224, 370), (1238, 374), (1241, 376)]        # W, H
CLASSES = ['Car', 'Car', 'Car', 'Pedestrian', 'Cyclist', 'Van', 'DontCare', 'Truck']


def synth_image(rs, w, h):
    """Smooth structure + texture + saturated patches (so that the photometric chain hits its wrap-around cases)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), dtype=np.float32)
    for c in range(3):
        fx, fy, ph = rs.uniform(0.002, 0.02), rs.uniform(0.004, 0.03), rs.uniform(0, 6.28)
        img[..., c] = 120 + 90 * np.sin(xx * fx + ph) * np.cos(yy * fy - ph) + rs.uniform(-30, 30)
    img += rs.normal(0, 12, size=img.shape).astype(np.float32)
    for _ in range(6):                                                # bright / dark / grey rectangles
        x0, y0 = rs.randint(0, w - 40), rs.randint(0, h - 30)
        x1, y1 = x0 + rs.randint(10, 200), y0 + rs.randint(8, 120)
        img[y0:y1, x0:x1] = rs.choice([0, 255, 128, 250, 5])
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_label_lines(rs, w, h, n, occ_choices=None):
    lines = []
    for _ in range(n):
        cls = CLASSES[rs.randint(len(CLASSES))]
        trunc = -1 if cls == 'DontCare' else round(float(rs.choice([0, 0, 0.1, 0.3, 0.6])), 2)
        occ = -1 if cls == 'DontCare' else (int(rs.randint(0, 4)) if occ_choices is None else int(rs.choice(occ_choices)))
        z = float(rs.uniform(1.0, 75.0))
        x = float(rs.uniform(-0.5, 0.5) * z * 1.2)
        y = float(rs.uniform(1.2, 2.0))
        hh, ww, ll = float(rs.uniform(1.3, 2.0)), float(rs.uniform(1.4, 1.9)), float(rs.uniform(3.0, 4.8))
        ry = float(rs.uniform(-np.pi, np.pi))
        u = 609.56 + 721.54 * x / z
        v = 172.85 + 721.54 * (y - hh / 2) / z
        bw, bh = 721.54 * ll / z * rs.uniform(0.5, 1.0), 721.54 * hh / z
        x1, x2 = np.clip([u - bw / 2 + rs.uniform(-4, 4), u + bw / 2 + rs.uniform(-4, 4)], 0, w - 1)
        y1, y2 = np.clip([v - bh / 2 + rs.uniform(-3, 3), v + bh / 2 + rs.uniform(-3, 3)], 0, h - 1)
        alpha = ry - np.arctan2(x, z)
        alpha = (alpha + np.pi) % (2 * np.pi) - np.pi
        lines.append('%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f' % (
            cls, trunc, occ, alpha, x1, y1, x2, y2, hh, ww, ll, x, y, z, ry))
    return lines


def synth_calib_lines(rs):
    f = 721.5377 + rs.uniform(-15, 15)
    cu, cv = 609.5593 + rs.uniform(-8, 8), 172.854 + rs.uniform(-5, 5)
    p2 = [f, 0, cu, 44.85728 + rs.uniform(-1, 1), 0, f, cv, 0.2163791 + rs.uniform(-0.1, 0.1), 0, 0, 1, 0.002745884]
    fmt = lambda name, vals: name + ': ' + ' '.join('%.12e' % v for v in vals)
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    rt = [0, -1, 0, 0, 0, 0, -1, -0.08, 1, 0, 0, -0.27]
    return [fmt('P0', p2), fmt('P1', p2), fmt('P2', p2), fmt('P3', p2), fmt('R0_rect', eye),
            fmt('Tr_velo_to_cam', rt), fmt('Tr_imu_to_velo', rt)]


def make_tree(root, n_images=6, seed=7, images=True, occ_choices=None):
    """Writes root/{ImageSets/train.txt, training/{image_2,label_2,calib}/%06d.*}; returns the list of ids."""
    rs = np.random.RandomState(seed)
    for d in ('ImageSets', 'training/image_2', 'training/label_2', 'training/calib'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    ids = []
    for k in range(n_images):
        idx = 3 * k + 1
        w, h = SIZES[k % len(SIZES)]
        if images:
            Image.fromarray(synth_image(rs, w, h)).save(os.path.join(root, 'training/image_2/%06d.png' % idx))
        n = 0 if k == 4 else int(rs.randint(1, 9))                    # one image without objects
        with open(os.path.join(root, 'training/label_2/%06d.txt' % idx), 'w') as f:
            f.write('\n'.join(synth_label_lines(rs, w, h, n, occ_choices)) + ('\n' if n else ''))
        with open(os.path.join(root, 'training/calib/%06d.txt' % idx), 'w') as f:
            f.write('\n'.join(synth_calib_lines(rs)) + '\n')
        ids.append('%06d' % idx)
    for split in ('train', 'val', 'trainval', 'test'):
        with open(os.path.join(root, 'ImageSets/%s.txt' % split), 'w') as f:
            f.write('\n'.join(ids) + '\n')
    return ids
