"""oracle/kitti_eval.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Plain-Python restatement of the reference's KITTI evaluation, SURVEY.md section 8 row f4:

    lib/datasets/kitti/kitti_eval_python/rotate_iou.py:17-258   rotated-box intersection (device functions)
    lib/datasets/kitti/kitti_eval_python/eval.py:9-27           get_thresholds
    lib/datasets/kitti/kitti_eval_python/eval.py:30-82          clean_data
    lib/datasets/kitti/kitti_eval_python/eval.py:160-228        image_box_overlap, bev / 3-D overlaps
    lib/datasets/kitti/kitti_eval_python/eval.py:231-345        compute_statistics_jit
    lib/datasets/kitti/kitti_eval_python/eval.py:524-640        eval_class, get_mAP, get_mAP_R40
    lib/datasets/kitti/kitti_eval_python/eval.py:727-825        get_official_eval_result (numbers and report text)
    lib/datasets/kitti/kitti_eval_python/kitti_common.py:294-347  get_label_anno(s)

PARITY STATUS.  Pinned on tests/golden/kitti_eval.npz, recorded by tests/golden/make_kitti_eval_golden.py from the
reference's own eval.py / kitti_common.py and the DEVICE functions of rotate_iou.py executed as plain Python (numba
is absent; its decorators are stubbed to identity).  Two caveats, both stated in that script: the CUDA launch wrapper
rotate_iou_gpu_eval cannot run and is replaced by a loop over the reference's devRotateIoUEval; and the device
functions then run on numpy float32 scalars -- float32 with one rounding per operation -- whereas numba-CUDA types a
few intermediates as float64 and may contract multiply-adds, so the fixture is "the reference's algorithm in float32",
not a recording of its GPU output.  (The difference is ~1e-7 in an overlap, except for exactly degenerate pairs --
identical or exactly touching boxes -- where the corner-inside tests sit on rounding.)  Against that fixture the
rotated overlaps are pinned bit-exactly, and so are average precisions, precision / recall curves and the report keys.

The rotated-box functions work in float32 with one rounding per operation (see the comment above them); the
statistics in float64, as the reference's.
"""
import math

import numpy as np

CLASS_NAMES = ['car', 'pedestrian', 'cyclist', 'van', 'person_sitting', 'truck']
MIN_HEIGHT, MAX_OCCLUSION, MAX_TRUNCATION = [40, 25, 25], [0, 1, 2], [0.15, 0.3, 0.5]
N_SAMPLE_PTS = 41


# ------------------------------------------------------------------------------------------------
# rotated boxes (x, y, dx, dy, angle) -- float32, one rounding per operation, no fused multiply-add: what the
# reference's device functions compute when executed on numpy float32 scalars (how the fixture was recorded)
# ------------------------------------------------------------------------------------------------
F = np.float32


def corners_of(box):
    """rbbox_to_corners (rotate_iou.py:206-229): 4 corners, clockwise, as a flat list of 8 float32."""
    x, y, dx, dy, ang = (F(v) for v in box)
    c, s = F(math.cos(float(ang))), F(math.sin(float(ang)))
    out = []
    for cx, cy in ((-dx / F(2), -dy / F(2)), (-dx / F(2), dy / F(2)), (dx / F(2), dy / F(2)), (dx / F(2), -dy / F(2))):
        out += [c * cx + s * cy + x, -s * cx + c * cy + y]
    return out


def inside_quad(px, py, q):
    """point_in_quadrilateral (rotate_iou.py:166-182): projections onto the two edges leaving corner 0."""
    ab0, ab1, ad0, ad1 = q[2] - q[0], q[3] - q[1], q[6] - q[0], q[7] - q[1]
    ap0, ap1 = px - q[0], py - q[1]
    abab, abap = ab0 * ab0 + ab1 * ab1, ab0 * ap0 + ab1 * ap1
    adad, adap = ad0 * ad0 + ad1 * ad1, ad0 * ap0 + ad1 * ap1
    return bool(abab >= abap and abap >= 0 and adad >= adap and adap >= 0)


def edge_crossing(p1, p2, i, j):
    """line_segment_intersection (rotate_iou.py:79-122): crossing point of edge i of p1 and edge j of p2, or None."""
    ax, ay, bx, by = p1[2 * i], p1[2 * i + 1], p1[2 * ((i + 1) % 4)], p1[2 * ((i + 1) % 4) + 1]
    cx, cy, dx, dy = p2[2 * j], p2[2 * j + 1], p2[2 * ((j + 1) % 4)], p2[2 * ((j + 1) % 4) + 1]
    ba0, ba1, da0, ca0, da1, ca1 = bx - ax, by - ay, dx - ax, cx - ax, dy - ay, cy - ay
    acd = da1 * ca0 > ca1 * da0
    bcd = (dy - by) * (cx - bx) > (cy - by) * (dx - bx)
    if acd == bcd:
        return None
    abc = ca1 * ba0 > ba1 * ca0
    abd = da1 * ba0 > ba1 * da0
    if abc == abd:
        return None
    dc0, dc1 = dx - cx, dy - cy
    abba, cddc = ax * by - bx * ay, cx * dy - dx * cy
    dh = ba1 * dc0 - ba0 * dc1
    with np.errstate(divide='ignore', invalid='ignore'):
        return (abba * dc0 - ba0 * cddc) / dh, (abba * dc1 - ba1 * cddc) / dh


def intersection_area(b1, b2):
    """inter (rotate_iou.py:232-246): polygon of the corner-in-other-box points and the edge crossings, ordered
    around its centroid by the reference's monotone angle key, area by a triangle fan."""
    p1, p2 = corners_of(b1), corners_of(b2)
    pts = []
    for i in range(4):                                           # quadrilateral_intersection (:185-203)
        if inside_quad(p1[2 * i], p1[2 * i + 1], p2):
            pts.append((p1[2 * i], p1[2 * i + 1]))
        if inside_quad(p2[2 * i], p2[2 * i + 1], p1):
            pts.append((p2[2 * i], p2[2 * i + 1]))
    for i in range(4):
        for j in range(4):
            x = edge_crossing(p1, p2, i, j)
            if x is not None:
                pts.append(x)
    n = len(pts)
    if n == 0:
        return F(0.0)
    cx, cy = F(0.0), F(0.0)
    for p in pts:
        cx, cy = cx + p[0], cy + p[1]
    cx, cy = cx / F(n), cy / F(n)
    keys = []                                                    # sort_vertex_in_convex_polygon (:33-72)
    with np.errstate(divide='ignore', invalid='ignore'):
        for (x, y) in pts:
            vx, vy = x - cx, y - cy
            d = F(math.sqrt(float(vx * vx + vy * vy)))
            vx, vy = vx / d, vy / d
            keys.append(F(-2) - vx if vy < 0 else vx)
    for i in range(1, n):                                        # the reference's insertion sort
        if keys[i - 1] > keys[i]:
            k, p, j = keys[i], pts[i], i
            while j > 0 and keys[j - 1] > k:
                keys[j], pts[j] = keys[j - 1], pts[j - 1]
                j -= 1
            keys[j], pts[j] = k, p
    total = F(0.0)                                               # area (:23-30)
    for i in range(n - 2):
        a, b, c = pts[0], pts[i + 1], pts[i + 2]
        total = total + abs(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / F(2.0))
    return total


def rotate_iou(boxes, query_boxes, criterion=-1):
    """rotate_iou_gpu_eval (rotate_iou.py:299-330) + devRotateIoUEval (:249-258): [N,5] x [K,5] -> [N,K].
    criterion -1: IoU; 0: inter / area(query box); 1: inter / area(box); 2: intersection area.
    (The kernel passes (query box, box) as (rbox1, rbox2), :293-296.)"""
    b = np.asarray(boxes).astype(np.float32)
    q = np.asarray(query_boxes).astype(np.float32)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    for n in range(b.shape[0]):
        for k in range(q.shape[0]):
            a1, a2 = q[k, 2] * q[k, 3], b[n, 2] * b[n, 3]
            ai = intersection_area(q[k], b[n])
            out[n, k] = ai / (a1 + a2 - ai) if criterion == -1 else (ai / a1 if criterion == 0 else (ai / a2 if criterion == 1 else ai))
    return out.astype(np.asarray(boxes).dtype)


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """eval.py:160-187."""
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = np.zeros((n, k), dtype=boxes.dtype)
    for j in range(k):
        qa = (query_boxes[j, 2] - query_boxes[j, 0]) * (query_boxes[j, 3] - query_boxes[j, 1])
        for i in range(n):
            iw = min(boxes[i, 2], query_boxes[j, 2]) - max(boxes[i, 0], query_boxes[j, 0])
            ih = min(boxes[i, 3], query_boxes[j, 3]) - max(boxes[i, 1], query_boxes[j, 1])
            if iw > 0 and ih > 0:
                ba = (boxes[i, 2] - boxes[i, 0]) * (boxes[i, 3] - boxes[i, 1])
                ua = ba + qa - iw * ih if criterion == -1 else (ba if criterion == 0 else (qa if criterion == 1 else 1.0))
                out[i, j] = iw * ih / ua
    return out


def d3_box_overlap(boxes, qboxes, criterion=-1):
    """eval.py:195-228: BEV intersection area x vertical overlap (camera frame: y down, box origin at its bottom)."""
    rinc = rotate_iou(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2)
    for i in range(boxes.shape[0]):
        for j in range(qboxes.shape[0]):
            if rinc[i, j] > 0:
                ih = min(boxes[i, 1], qboxes[j, 1]) - max(boxes[i, 1] - boxes[i, 4], qboxes[j, 1] - qboxes[j, 4])
                if ih > 0:
                    v1, v2 = boxes[i, 3] * boxes[i, 4] * boxes[i, 5], qboxes[j, 3] * qboxes[j, 4] * qboxes[j, 5]
                    inc = ih * rinc[i, j]
                    ua = v1 + v2 - inc if criterion == -1 else (v1 if criterion == 0 else (v2 if criterion == 1 else inc))
                    rinc[i, j] = inc / ua
                else:
                    rinc[i, j] = 0.0
    return rinc


# ------------------------------------------------------------------------------------------------
# annotations
# ------------------------------------------------------------------------------------------------
def read_label_file(path):
    """kitti_common.get_label_anno (:294-330)."""
    rows = [ln.strip().split(' ') for ln in open(path).readlines()]
    a = {'name': np.array([r[0] for r in rows]), 'truncated': np.array([float(r[1]) for r in rows]),
         'occluded': np.array([int(r[2]) for r in rows]), 'alpha': np.array([float(r[3]) for r in rows]),
         'bbox': np.array([[float(v) for v in r[4:8]] for r in rows]).reshape(-1, 4),
         'dimensions': np.array([[float(v) for v in r[8:11]] for r in rows]).reshape(-1, 3)[:, [2, 0, 1]],
         'location': np.array([[float(v) for v in r[11:14]] for r in rows]).reshape(-1, 3),
         'rotation_y': np.array([float(r[14]) for r in rows]).reshape(-1)}
    a['score'] = np.array([float(r[15]) for r in rows]) if rows and len(rows[0]) == 16 else np.zeros([len(a['bbox'])])
    return a


def clean_data(gt, dt, cls, difficulty):
    """eval.py:30-82 -> (number of valid gt, ignored_gt, ignored_dt, don't-care boxes)."""
    name = CLASS_NAMES[cls]
    ignored_gt, ignored_dt, dc, valid = [], [], [], 0
    for i in range(len(gt['name'])):
        g = gt['name'][i].lower()
        h = gt['bbox'][i][3] - gt['bbox'][i][1]
        match = 1 if g == name else (0 if (name == 'pedestrian' and g == 'person_sitting') or (name == 'car' and g == 'van') else -1)
        hard = (gt['occluded'][i] > MAX_OCCLUSION[difficulty] or gt['truncated'][i] > MAX_TRUNCATION[difficulty]
                or h <= MIN_HEIGHT[difficulty])
        if match == 1 and not hard:
            ignored_gt.append(0)
            valid += 1
        elif match == 0 or (hard and match == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
        if gt['name'][i] == 'DontCare':
            dc.append(gt['bbox'][i])
    for i in range(len(dt['name'])):
        h = abs(dt['bbox'][i, 3] - dt['bbox'][i, 1])
        ignored_dt.append(1 if h < MIN_HEIGHT[difficulty] else (0 if dt['name'][i].lower() == name else -1))
    return valid, ignored_gt, ignored_dt, dc


def statistics(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap, thresh=0.0,
               compute_fp=False, compute_aos=False):
    """compute_statistics_jit (eval.py:231-345) -> (tp, fp, fn, similarity, scores of the true positives)."""
    nd, ng = dt_datas.shape[0], gt_datas.shape[0]
    scores, dt_alpha, gt_alpha = dt_datas[:, -1], dt_datas[:, 4], gt_datas[:, 4]
    assigned = [False] * nd
    below = [compute_fp and scores[j] < thresh for j in range(nd)]
    NONE = -10000000
    tp = fp = fn = 0
    similarity = 0
    tp_scores, deltas = [], []
    for i in range(ng):
        if ignored_gt[i] == -1:
            continue
        best, best_score, max_ov, took_ignored = -1, NONE, 0, False
        for j in range(nd):
            if ignored_det[j] == -1 or assigned[j] or below[j]:
                continue
            ov = overlaps[j, i]
            if not compute_fp and ov > min_overlap and scores[j] > best_score:
                best, best_score = j, scores[j]
            elif compute_fp and ov > min_overlap and (ov > max_ov or took_ignored) and ignored_det[j] == 0:
                max_ov, best, best_score, took_ignored = ov, j, 1, False
            elif compute_fp and ov > min_overlap and best_score == NONE and ignored_det[j] == 1:
                best, best_score, took_ignored = j, 1, True
        if best_score == NONE and ignored_gt[i] == 0:
            fn += 1
        elif best_score != NONE and (ignored_gt[i] == 1 or ignored_det[best] == 1):
            assigned[best] = True
        elif best_score != NONE:
            tp += 1
            tp_scores.append(scores[best])
            if compute_aos:
                deltas.append(gt_alpha[i] - dt_alpha[best])
            assigned[best] = True
    if compute_fp:
        for j in range(nd):
            if not (assigned[j] or ignored_det[j] == -1 or ignored_det[j] == 1 or below[j]):
                fp += 1
        stuff = 0
        if metric == 0:
            ov_dc = image_box_overlap(dt_datas[:, :4], dc_bboxes, 0)
            for i in range(dc_bboxes.shape[0]):
                for j in range(nd):
                    if assigned[j] or ignored_det[j] in (-1, 1) or below[j]:
                        continue
                    if ov_dc[j, i] > min_overlap:
                        assigned[j] = True
                        stuff += 1
        fp -= stuff
        if compute_aos:
            similarity = sum((1.0 + math.cos(d)) / 2.0 for d in deltas) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, similarity, np.array(tp_scores)


def recall_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """get_thresholds (eval.py:9-27): the detection scores at which recall crosses the 41 sample points."""
    scores = np.sort(scores)[::-1]
    cur, out = 0, []
    for i, s in enumerate(scores):
        l_rec = (i + 1) / num_gt
        r_rec = (i + 2) / num_gt if i < len(scores) - 1 else l_rec
        if (r_rec - cur) < (cur - l_rec) and i < len(scores) - 1:
            continue
        out.append(s)
        cur += 1 / (num_sample_pts - 1.0)
    return out


def overlaps_per_frame(gt_annos, dt_annos, metric):
    """What calculate_iou_partly (eval.py:404-486) yields per frame when called as eval_class calls it (detections
    first): a [num_dt, num_gt] matrix."""
    out = []
    for g, d in zip(gt_annos, dt_annos):
        if metric == 0:
            out.append(image_box_overlap(d['bbox'], g['bbox']))
        else:
            def boxes(a):
                if metric == 1:
                    return np.concatenate([a['location'][:, [0, 2]], a['dimensions'][:, [0, 2]], a['rotation_y'][..., None]], 1)
                return np.concatenate([a['location'], a['dimensions'], a['rotation_y'][..., None]], 1)
            fn = rotate_iou if metric == 1 else d3_box_overlap
            out.append(fn(boxes(d), boxes(g)).astype(np.float64))
    return out


def eval_class(gt_annos, dt_annos, classes, difficulties, metric, min_overlaps, compute_aos=False):
    """eval_class (eval.py:524-640) -> dict(recall, precision, orientation), each [class, difficulty, overlap, 41]."""
    overlaps = overlaps_per_frame(gt_annos, dt_annos, metric)
    shape = [len(classes), len(difficulties), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(classes):
        for l, diff in enumerate(difficulties):
            prepared, total_valid = [], 0
            for g, d in zip(gt_annos, dt_annos):
                valid, ig, idt, dc = clean_data(g, d, cls, diff)
                total_valid += valid
                dc = np.stack(dc, 0).astype(np.float64) if dc else np.zeros((0, 4))
                gd = np.concatenate([g['bbox'], g['alpha'][..., None]], 1)
                dd = np.concatenate([d['bbox'], d['alpha'][..., None], d['score'][..., None]], 1)
                prepared.append((gd, dd, np.array(ig, dtype=np.int64), np.array(idt, dtype=np.int64), dc))
            for k, min_ov in enumerate(min_overlaps[:, metric, m]):
                tp_scores = []
                for ov, (gd, dd, ig, idt, dc) in zip(overlaps, prepared):
                    tp_scores += statistics(ov, gd, dd, ig, idt, dc, metric, min_ov, 0.0, False)[4].tolist()
                ths = np.array(recall_thresholds(np.array(tp_scores), total_valid))
                pr = np.zeros([len(ths), 4])
                for ov, (gd, dd, ig, idt, dc) in zip(overlaps, prepared):
                    for t, th in enumerate(ths):
                        tp, fp, fn, sim, _ = statistics(ov, gd, dd, ig, idt, dc, metric, min_ov, th, True, compute_aos)
                        pr[t, 0] += tp
                        pr[t, 1] += fp
                        pr[t, 2] += fn
                        if sim != -1:
                            pr[t, 3] += sim
                with np.errstate(invalid='ignore', divide='ignore'):
                    for i in range(len(ths)):
                        recall[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 2])
                        precision[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 1])
                        if compute_aos:
                            aos[m, l, k, i] = pr[i, 3] / (pr[i, 0] + pr[i, 1])
                for i in range(len(ths)):
                    precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                    recall[m, l, k, i] = np.max(recall[m, l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
    return {'recall': recall, 'precision': precision, 'orientation': aos}


def ap11(prec):
    return sum(prec[..., i] for i in range(0, prec.shape[-1], 4)) / 11 * 100        # eval.py:643-647


def ap40(prec):
    return sum(prec[..., i] for i in range(1, prec.shape[-1])) / 40 * 100           # eval.py:650-654


def official_min_overlaps(classes):
    hi = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3)
    lo = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])
    return np.stack([hi, lo], 0)[:, :, classes]                                    # eval.py:728-734


def official_result(gt_annos, dt_annos, cls):
    """The numbers of get_official_eval_result (eval.py:727-825) for one class: dict name -> value with the
    reference's keys ('Car_3d_moderate_R40', ...), and the moderate 3-D AP|R40 it returns."""
    names = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting', 5: 'Truck'}
    mo = official_min_overlaps([cls])
    aos_on = next((a['alpha'][0] != -10 for a in dt_annos if a['alpha'].shape[0] != 0), False)   # eval.py:753-758
    res = {}
    for metric, tag in ((0, 'image'), (1, 'bev'), (2, '3d')):
        r = eval_class(gt_annos, dt_annos, [cls], [0, 1, 2], metric, mo, compute_aos=aos_on and metric == 0)
        for suffix, fn in (('', ap11), ('_R40', ap40)):
            ap = fn(r['precision'])
            for d, dn in enumerate(('easy', 'moderate', 'hard')):
                res['%s_%s_%s%s' % (names[cls], tag, dn, suffix)] = ap[0, d, 0]
            if metric == 0 and aos_on:
                a = fn(r['orientation'])
                for d, dn in enumerate(('easy', 'moderate', 'hard')):
                    res['%s_aos_%s%s' % (names[cls], dn, suffix)] = a[0, d, 0]
    return res, res['%s_3d_moderate_R40' % names[cls]]
