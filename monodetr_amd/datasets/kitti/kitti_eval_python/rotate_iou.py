"""Rotated-box overlaps on the device (``csrc/rotate_iou.hip`` through ``mdetr_rotate_iou_eval`` /
``mdetr_box3d_overlap_eval``) -- mirror of ``lib/datasets/kitti/kitti_eval_python/rotate_iou.py``.

``rotate_iou_gpu_eval(boxes, query_boxes, criterion, device_id)`` keeps the reference's signature (numpy in, numpy
out, one all-pairs block).  The evaluation itself uses the segmented forms, which compute only within-frame pairs for
a whole split in one launch."""
import numpy as np
import torch

from .... import _capi

_backend = None               # tests substitute the CPU emulation of the same kernel source (tests/native_emul.py)


def _lib():
    return _backend if _backend is not None else _capi.lib()


def _device(device_id):
    if _backend is not None:
        return torch.device('cpu')
    if not torch.cuda.is_available():
        raise RuntimeError("Not implemented on the CPU")
    return torch.device('cuda', device_id)


def _segmented(entry, boxes_list, qboxes_list, width, dtype, criterion, device_id):
    """boxes_list[f] [n_f, width], qboxes_list[f] [k_f, width] -> list of [n_f, k_f] numpy arrays."""
    dev = _device(device_id)
    n = np.array([len(b) for b in boxes_list], dtype=np.int64)
    k = np.array([len(q) for q in qboxes_list], dtype=np.int64)
    starts = [np.concatenate([[0], np.cumsum(v)]).astype(np.int64) for v in (n, k, n * k)]
    total = int(starts[2][-1])
    np_dtype = np.float32 if dtype == torch.float32 else np.float64
    cat = lambda lst: np.concatenate([np.asarray(a, dtype=np_dtype).reshape(-1, width) for a in lst], 0) if lst else np.zeros((0, width), np_dtype)
    out = torch.empty(max(total, 1), dtype=dtype, device=dev)
    if total:
        b, q = (torch.from_numpy(np.ascontiguousarray(cat(l))).to(dev) for l in (boxes_list, qboxes_list))
        bs, qs, os_ = (torch.from_numpy(s).to(dev) for s in starts)
        cuda = dev.type == 'cuda'
        rc = getattr(_lib(), entry)(b.data_ptr(), q.data_ptr(), bs.data_ptr(), qs.data_ptr(), os_.data_ptr(), len(boxes_list), total,
                                    int(criterion), out.data_ptr(), dev.index if cuda else -1,
                                    torch.cuda.current_stream(dev).cuda_stream if cuda else None)
        if rc != 0:
            _capi.check(rc, entry)
    flat = out.cpu().numpy()
    return [flat[starts[2][f]:starts[2][f + 1]].reshape(int(n[f]), int(k[f])) for f in range(len(boxes_list))]


def segmented_rotate_iou(boxes_list, qboxes_list, criterion=-1, device_id=0):
    """Per frame: [n_f, 5] x [k_f, 5] (x, y, dx, dy, angle) -> float32 [n_f, k_f]."""
    return _segmented("mdetr_rotate_iou_eval", boxes_list, qboxes_list, 5, torch.float32, criterion, device_id)


def segmented_box3d_overlap(boxes_list, qboxes_list, criterion=-1, device_id=0):
    """Per frame: [n_f, 7] x [k_f, 7] camera-frame boxes (x, y, z, l, h, w, ry) -> float64 [n_f, k_f]
    (eval.py:224-228: bird's-eye-view intersection in float32, height overlap and ratio in float64)."""
    return _segmented("mdetr_box3d_overlap_eval", boxes_list, qboxes_list, 7, torch.float64, criterion, device_id)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """[N, 5] x [K, 5] -> [N, K] in the dtype of ``boxes`` (computed in float32), as the reference's function."""
    boxes, query_boxes = np.asarray(boxes), np.asarray(query_boxes)
    if boxes.shape[0] == 0 or query_boxes.shape[0] == 0:
        return np.zeros((boxes.shape[0], query_boxes.shape[0]), dtype=np.float32)
    return segmented_rotate_iou([boxes], [query_boxes], criterion, device_id)[0].astype(boxes.dtype)


# ---- known-answer self check ---------------------------------------------------------------------------------------
# A defect here would silently change the AP that picks checkpoint_best.  Before the first evaluation of a process the
# kernel computes 4 x 3 overlaps under the three criteria the evaluation uses; the float32 bit patterns must equal the
# table below, which tests/test_kitti_eval_cpu.py holds to the restatement of the reference's device functions
# (oracle/kitti_eval.py, pinned on the fixture recorded from the reference's rotate_iou.py).
_KAT_BOXES = np.array([[0, 0, 4, 2, 0.0], [1, 1, 3, 3, 0.5], [10, -2, 2.5, 1.5, -1.2], [0.3, 0.2, 4, 2, 3.0]], dtype=np.float32)
_KAT_QUERY = np.array([[0, 0, 4, 2, 0.0], [0.5, 0.25, 2, 4, 0.3], [9.5, -2.2, 2, 2, 0.4]], dtype=np.float32)
_KAT_BITS = {
    -1: [1065353216, 1052080449, 0, 1050640856, 1055091924, 0, 0, 0, 1056588329, 1060655630, 1053012594, 0],
    0: [1065353216, 1057356788, 0, 1057044538, 1059541364, 0, 0, 0, 1059248056, 1062622065, 1057854693, 0],
    1: [1065353216, 1057356788, 0, 1055242571, 1058322990, 0, 0, 0, 1059959526, 1062622065, 1057854693, 0],
}
_self_checked = set()


def self_check(device_id=0):
    key = (_backend is not None, device_id)
    if key in _self_checked:
        return
    for criterion, want in _KAT_BITS.items():
        got = rotate_iou_gpu_eval(_KAT_BOXES, _KAT_QUERY, criterion, device_id).astype(np.float32).view(np.uint32).reshape(-1).tolist()
        if got != want:
            raise RuntimeError("mdetr_rotate_iou_eval failed its known-answer check (criterion %d): the evaluation must not be trusted" % criterion)
    _self_checked.add(key)
