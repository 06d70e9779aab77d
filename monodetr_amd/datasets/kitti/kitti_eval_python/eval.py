"""KITTI official evaluation -- mirror of ``lib/datasets/kitti/kitti_eval_python/eval.py`` (``get_official_eval_result``,
``do_eval``, ``eval_class``, ``get_mAP``, ``get_mAP_R40`` keep their signatures and return values).

How the work is split here (reference in parentheses):
  * overlaps: image boxes vectorised in numpy; bird's-eye-view and 3-D boxes on the device, ONE segmented launch per
    metric over the whole split, only within-frame pairs (numba-CUDA kernel per 50-frame part on all pairs of the
    part + a CPU pass for the height overlap, eval.py:190-228, :404-486);
  * ignore flags (``clean_data``): vectorised per frame;
  * the greedy assignment / recall thresholds / per-threshold tp, fp, fn accumulation: one native call per
    (class, difficulty, overlap) cell, ``mdetr_kitti_pr_curve`` (numba-jitted functions driven from Python loops over
    frames x 41 thresholds, eval.py:563-620).
"""
import ctypes
import io as sysio

import numpy as np

from .... import _capi
from .rotate_iou import segmented_box3d_overlap, segmented_rotate_iou
from .rotate_iou import self_check as rotate_iou_self_check

N_SAMPLE_PTS = 41
CLASS_NAMES = ['car', 'pedestrian', 'cyclist', 'van', 'person_sitting', 'truck']
_MIN_HEIGHT = np.array([40, 25, 25])
_MAX_OCCLUSION = np.array([0, 1, 2])
_MAX_TRUNCATION = np.array([0.15, 0.3, 0.5])
_NEUTRAL_FOR = {'pedestrian': 'person_sitting', 'car': 'van'}


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """[N,4] x [K,4] axis-aligned boxes -> [N,K]: IoU (-1), intersection over the box's (0) / the query's (1) area."""
    b, q = boxes[:, None, :], query_boxes[None, :, :]
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0])
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    area_q = (q[..., 2] - q[..., 0]) * (q[..., 3] - q[..., 1])
    hit = (iw > 0) & (ih > 0)
    inter = np.where(hit, iw * ih, 0.0)
    if criterion == -1:
        ua = area_b + area_q - inter
    elif criterion == 0:
        ua = area_b + 0 * area_q
    elif criterion == 1:
        ua = area_q + 0 * area_b
    else:
        ua = np.ones_like(inter)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(hit, inter / ua, 0.0).astype(boxes.dtype)


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """-> (number of valid ground truths, ignored_gt, ignored_dt, don't-care boxes); flags 0 = counts, 1 = neutral
    (too hard for this difficulty, or a look-alike class), -1 = other class."""
    name = CLASS_NAMES[current_class]
    gt_names = np.char.lower(gt_anno['name'].astype(str)) if len(gt_anno['name']) else np.zeros(0, dtype=str)
    height = gt_anno['bbox'][:, 3] - gt_anno['bbox'][:, 1]
    same = gt_names == name
    lookalike = gt_names == _NEUTRAL_FOR.get(name, '\0')
    hard = ((gt_anno['occluded'] > _MAX_OCCLUSION[difficulty]) | (gt_anno['truncated'] > _MAX_TRUNCATION[difficulty])
            | (height <= _MIN_HEIGHT[difficulty]))
    ignored_gt = np.where(same & ~hard, 0, np.where(lookalike | (hard & same), 1, -1)).astype(np.int64)
    dc_bboxes = gt_anno['bbox'][gt_anno['name'] == 'DontCare'] if len(gt_anno['name']) else np.zeros((0, 4))
    dt_names = np.char.lower(dt_anno['name'].astype(str)) if len(dt_anno['name']) else np.zeros(0, dtype=str)
    dt_height = np.abs(dt_anno['bbox'][:, 3] - dt_anno['bbox'][:, 1])
    ignored_dt = np.where(dt_height < _MIN_HEIGHT[difficulty], 1, np.where(dt_names == name, 0, -1)).astype(np.int64)
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc_bboxes


def _boxes_for(anno, metric):
    if metric == 1:
        return np.concatenate([anno['location'][:, [0, 2]], anno['dimensions'][:, [0, 2]], anno['rotation_y'][..., np.newaxis]], axis=1)
    return np.concatenate([anno['location'], anno['dimensions'], anno['rotation_y'][..., np.newaxis]], axis=1)


def calculate_overlaps(gt_annos, dt_annos, metric):
    """Per frame the [num_dt, num_gt] float64 overlap matrix of ``metric`` (0 image, 1 bird's-eye view, 2 3-D)."""
    assert len(gt_annos) == len(dt_annos)
    if metric == 0:
        return [image_box_overlap(d['bbox'], g['bbox']).astype(np.float64) for g, d in zip(gt_annos, dt_annos)]
    if metric not in (1, 2):
        raise ValueError("unknown metric")
    dts, gts = [_boxes_for(d, metric) for d in dt_annos], [_boxes_for(g, metric) for g in gt_annos]
    rotate_iou_self_check()                                  # known-answer test of the overlap kernel, once per process
    fn = segmented_rotate_iou if metric == 1 else segmented_box3d_overlap
    return [o.astype(np.float64) for o in fn(dts, gts, -1)]


def _starts(counts):
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_parts=50,
               DIForDIS=True):
    """-> dict(recall, precision, orientation), each [num_class, num_difficulty, num_minoverlap, 41]
    (``min_overlaps`` is [num_minoverlap, metric, class]; ``num_parts`` is accepted and unused -- nothing is
    computed in parts here)."""
    if not DIForDIS:
        raise NotImplementedError("distance-based difficulty (clean_data_by_distance) is not mirrored")
    assert len(gt_annos) == len(dt_annos)
    lib = _capi.lib()
    overlaps = calculate_overlaps(gt_annos, dt_annos, metric)
    n_frames = len(gt_annos)
    ov_flat = np.ascontiguousarray(np.concatenate([o.reshape(-1) for o in overlaps]) if overlaps else np.zeros(0))
    ov_start = _starts([o.size for o in overlaps])
    gt_datas = np.ascontiguousarray(np.concatenate([np.concatenate([g['bbox'], g['alpha'][..., np.newaxis]], 1) for g in gt_annos], 0), dtype=np.float64)
    dt_datas = np.ascontiguousarray(np.concatenate([np.concatenate([d['bbox'], d['alpha'][..., np.newaxis], d['score'][..., np.newaxis]], 1)
                                                    for d in dt_annos], 0), dtype=np.float64)
    gt_start, dt_start = _starts([len(g['name']) for g in gt_annos]), _starts([len(d['name']) for d in dt_annos])
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    pr, ths, nth = np.zeros((N_SAMPLE_PTS, 4)), np.zeros(N_SAMPLE_PTS), ctypes.c_int(0)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            cleaned = [clean_data(g, d, current_class, difficulty) for g, d in zip(gt_annos, dt_annos)]
            total_valid = sum(c[0] for c in cleaned)
            ignored_gt = np.ascontiguousarray(np.concatenate([c[1] for c in cleaned]) if cleaned else np.zeros(0, np.int64))
            ignored_dt = np.ascontiguousarray(np.concatenate([c[2] for c in cleaned]) if cleaned else np.zeros(0, np.int64))
            dontcares = np.ascontiguousarray(np.concatenate([c[3].reshape(-1, 4) for c in cleaned], 0) if cleaned else np.zeros((0, 4)), dtype=np.float64)
            dc_start = _starts([len(c[3]) for c in cleaned])
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                rc = lib.mdetr_kitti_pr_curve(_ptr(ov_flat), _ptr(ov_start), _ptr(gt_datas), _ptr(dt_datas), _ptr(gt_start), _ptr(dt_start),
                                              _ptr(ignored_gt), _ptr(ignored_dt), _ptr(dontcares), _ptr(dc_start), n_frames, int(metric),
                                              float(min_overlap), int(total_valid), 1 if compute_aos else 0, N_SAMPLE_PTS,
                                              _ptr(pr), _ptr(ths), ctypes.byref(nth))
                if rc != 0:
                    raise RuntimeError("mdetr_kitti_pr_curve failed (%d): more than %d recall thresholds?" % (rc, N_SAMPLE_PTS))
                n = nth.value
                with np.errstate(divide='ignore', invalid='ignore'):
                    recall[m, l, k, :n] = pr[:n, 0] / (pr[:n, 0] + pr[:n, 2])
                    precision[m, l, k, :n] = pr[:n, 0] / (pr[:n, 0] + pr[:n, 1])
                    if compute_aos:
                        aos[m, l, k, :n] = pr[:n, 3] / (pr[:n, 0] + pr[:n, 1])
                # interpolated curves: value at threshold i = maximum over thresholds >= i (up to the array's end)
                for arr in (precision, recall) + ((aos,) if compute_aos else ()):
                    for i in range(n):
                        arr[m, l, k, i] = np.max(arr[m, l, k, i:], axis=-1)
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP(prec):
    """11-point interpolated AP (recall 0, 0.1, ..., 1)."""
    sums = 0
    for i in range(0, prec.shape[-1], 4):
        sums = sums + prec[..., i]
    return sums / 11 * 100


def get_mAP_R40(prec):
    """40-point interpolated AP (recall 1/40, ..., 1)."""
    sums = 0
    for i in range(1, prec.shape[-1]):
        sums = sums + prec[..., i]
    return sums / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, PR_detail_dict=None, DIForDIS=True):
    """-> (mAP_bbox, mAP_bev, mAP_3d, mAP_aos, and the same four |R40), each [num_class, 3 difficulties, num_minoverlap]."""
    difficultys = [0, 1, 2]
    out, out40 = {}, {}
    for metric, tag in ((0, 'bbox'), (1, 'bev'), (2, '3d')):
        ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos and metric == 0,
                         DIForDIS=DIForDIS)
        out[tag], out40[tag] = get_mAP(ret["precision"]), get_mAP_R40(ret["precision"])
        if PR_detail_dict is not None:
            PR_detail_dict[tag] = ret['precision']
        if metric == 0:
            out['aos'] = out40['aos'] = None
            if compute_aos:
                out['aos'], out40['aos'] = get_mAP(ret["orientation"]), get_mAP_R40(ret["orientation"])
                if PR_detail_dict is not None:
                    PR_detail_dict['aos'] = ret['orientation']
    return out['bbox'], out['bev'], out['3d'], out['aos'], out40['bbox'], out40['bev'], out40['3d'], out40['aos']


def print_str(value, *arg, sstream=None):
    if sstream is None:
        sstream = sysio.StringIO()
    sstream.truncate(0)
    sstream.seek(0)
    print(value, *arg, file=sstream)
    return sstream.getvalue()


_OVERLAP_STRICT = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3)
_OVERLAP_LOOSE = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])
_CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting', 5: 'Truck'}


def get_official_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None):
    """-> (report text, dict of the named APs, 3-D AP|R40 of the first class at moderate difficulty) -- the three
    values ``KITTI_Dataset.eval`` consumes (kitti_dataset.py:109-116)."""
    min_overlaps = np.stack([_OVERLAP_STRICT, _OVERLAP_LOOSE], axis=0)           # [2, metric, class]
    name_to_class = {v: n for n, v in _CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = min_overlaps[:, :, current_classes]
    compute_aos = False                                                          # orientation only if alpha is provided
    for anno in dt_annos:
        if anno['alpha'].shape[0] != 0:
            compute_aos = bool(anno['alpha'][0] != -10)
            break
    ap = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, PR_detail_dict=PR_detail_dict, DIForDIS=True)
    r11 = dict(zip(('bbox', 'bev', '3d', 'aos'), ap[:4]))
    r40 = dict(zip(('bbox', 'bev', '3d', 'aos'), ap[4:]))
    result, ret_dict = '', {}
    levels = ('easy', 'moderate', 'hard')
    for j, curcls in enumerate(current_classes):
        cname = _CLASS_TO_NAME[curcls]
        for i in range(min_overlaps.shape[0]):
            for table, title, suffix in ((r11, 'AP', ''), (r40, 'AP_R40', '_R40')):
                result += print_str("%s %s@{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j]) % (cname, title))
                for tag, label in (('bbox', 'bbox'), ('bev', 'bev '), ('3d', '3d  ')):
                    v = table[tag]
                    result += print_str("%s AP:%.4f, %.4f, %.4f" % (label, v[j, 0, i], v[j, 1, i], v[j, 2, i]))
                if compute_aos:
                    v = table['aos']
                    result += print_str("aos  AP:%.2f, %.2f, %.2f" % (v[j, 0, i], v[j, 1, i], v[j, 2, i]))
                    if i == 0:
                        for d, lv in enumerate(levels):
                            ret_dict['%s_aos_%s%s' % (cname, lv, suffix)] = v[j, d, 0]
                if i == 0:
                    for tag, key in (('3d', '3d'), ('bev', 'bev'), ('bbox', 'image')):
                        for d, lv in enumerate(levels):
                            ret_dict['%s_%s_%s%s' % (cname, key, lv, suffix)] = table[tag][j, d, 0]
    return result, ret_dict, r40['3d'][0, 1, 0]
