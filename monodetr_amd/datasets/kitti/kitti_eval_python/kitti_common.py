"""Reading KITTI label / result files into annotation dicts -- mirror of the part of
``lib/datasets/kitti/kitti_eval_python/kitti_common.py`` the evaluation uses (:294-347)."""
import pathlib
import re

import numpy as np


def get_image_index_str(img_idx):
    return "{:06d}".format(img_idx)


def get_label_anno(label_path):
    """One file -> dict(name, truncated, occluded, alpha, bbox [n,4], dimensions [n,3] as (l, h, w), location [n,3],
    rotation_y, score); score is 0 for ground-truth files (15 fields per line)."""
    with open(label_path, 'r') as f:
        rows = [line.strip().split(' ') for line in f.readlines()]
    col = lambda a, b, conv=float: np.array([[conv(v) for v in r[a:b]] for r in rows]).reshape(-1, b - a)
    anno = {
        'name': np.array([r[0] for r in rows]),
        'truncated': col(1, 2).reshape(-1),
        'occluded': col(2, 3, int).reshape(-1),
        'alpha': col(3, 4).reshape(-1),
        'bbox': col(4, 8),
        'dimensions': col(8, 11)[:, [2, 0, 1]],          # file order h, w, l -> l, h, w
        'location': col(11, 14),
        'rotation_y': col(14, 15).reshape(-1),
    }
    if rows and len(rows[0]) == 16:
        anno['score'] = col(15, 16).reshape(-1)
    else:
        anno['score'] = np.zeros([len(anno['bbox'])])
    return anno


def get_label_annos(label_folder, image_ids=None):
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        pattern = re.compile(r'^\d{6}.txt$')
        image_ids = sorted(int(p.stem) for p in folder.glob('*.txt') if pattern.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    return [get_label_anno(folder / (get_image_index_str(idx) + '.txt')) for idx in image_ids]
