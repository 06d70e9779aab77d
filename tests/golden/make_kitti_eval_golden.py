"""Records tests/golden/kitti_eval.npz by running the REFERENCE's own evaluation
(/root/reference/lib/datasets/kitti/kitti_eval_python/{eval,rotate_iou,kitti_common}.py) on synthetic ground truth and
detections (tests/kitti_synth.py, tests/kitti_synth_dets.py).  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_kitti_eval_golden.py

numba is absent, so its decorators are stubbed to identity and the reference's functions run as plain Python:
eval.py entirely, and rotate_iou.py's DEVICE functions (devRotateIoUEval and everything below it) with
`cuda.local.array` -> a float32 numpy array.  The CUDA kernel launch itself (rotate_iou_gpu_eval) cannot run; it is
replaced by a loop that calls the reference's own devRotateIoUEval with the kernel's argument order
(rotate_iou.py:293-296: iou[n, k] = devRotateIoUEval(query_boxes[k], boxes[n])).
Caveat (stated in oracle/kitti_eval.py): under numba the device functions are typed float32 with a few float64
promotions; as plain Python on numpy scalars the promotions differ, so IoU values are pinned to ~1e-6, not bit-exactly.
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

import kitti_synth                                # noqa: E402
import kitti_synth_dets                           # noqa: E402


def install_stubs():
    ident = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    numba = types.ModuleType('numba')
    numba.jit = ident
    numba.float32 = np.float32
    cuda = types.ModuleType('numba.cuda')
    cuda.jit = ident
    cuda.local = types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype=np.float32))
    numba.cuda = cuda
    sys.modules['numba'], sys.modules['numba.cuda'] = numba, cuda
    sk = types.ModuleType('skimage')
    sk.io = types.ModuleType('skimage.io')
    sys.modules['skimage'], sys.modules['skimage.io'] = sk, sk.io
    cv2 = types.ModuleType('cv2')
    sys.modules.setdefault('cv2', cv2)
    tv = types.ModuleType('torchvision')
    tv.ops = types.ModuleType('torchvision.ops')
    tv.ops.boxes = types.ModuleType('torchvision.ops.boxes')
    tv.ops.boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    for name, mod in (('torchvision', tv), ('torchvision.ops', tv.ops), ('torchvision.ops.boxes', tv.ops.boxes)):
        sys.modules.setdefault(name, mod)


def sample_boxes(rs, n):
    """[n, 5] (x, z, dx, dz, angle) boxes around a common centre so that many pairs overlap."""
    return np.stack([rs.uniform(-3, 3, n), rs.uniform(10, 16, n), rs.uniform(1.4, 4.5, n), rs.uniform(1.4, 4.5, n),
                     rs.uniform(-np.pi, np.pi, n)], 1)


def main():
    install_stubs()
    sys.path.insert(0, '/root/reference')
    from lib.datasets.kitti.kitti_eval_python import eval as ref_eval
    from lib.datasets.kitti.kitti_eval_python import kitti_common as ref_common
    from lib.datasets.kitti.kitti_eval_python import rotate_iou as ref_riou

    def riou(boxes, query_boxes, criterion=-1, device_id=0):
        boxes32, q32 = boxes.astype(np.float32), query_boxes.astype(np.float32)
        out = np.zeros((boxes.shape[0], query_boxes.shape[0]), dtype=np.float32)
        for n in range(boxes.shape[0]):
            for k in range(query_boxes.shape[0]):
                out[n, k] = ref_riou.devRotateIoUEval(q32[k].copy(), boxes32[n].copy(), criterion)
        return out.astype(boxes.dtype)
    ref_eval.rotate_iou_gpu_eval = riou

    out = {}
    rs = np.random.RandomState(0)
    a, b = sample_boxes(rs, 9), sample_boxes(rs, 7)
    b[0] = a[0]                                                     # identical boxes
    b[1] = a[1] + [0, 0, 0, 0, np.pi / 2]                           # same centre, rotated by 90 degrees
    a[2, 4], b[2, :] = 0.0, [a[2, 0] + 0.5, a[2, 1], a[2, 2], a[2, 3], 0.0]   # axis-aligned, shifted
    b[3] = [40, 40, 2, 2, 0.3]                                      # disjoint from everything
    out['riou_boxes'], out['riou_qboxes'] = a, b
    for crit in (-1, 0, 1, 2):
        out['riou_c%d' % crit] = riou(a, b, crit)
    b3a = np.concatenate([a[:, :1], rs.uniform(1.2, 2.0, (9, 1)), a[:, 1:2], a[:, 2:3], rs.uniform(1.3, 2.0, (9, 1)), a[:, 3:4], a[:, 4:]], 1)
    b3b = np.concatenate([b[:, :1], rs.uniform(1.2, 2.0, (7, 1)), b[:, 1:2], b[:, 2:3], rs.uniform(1.3, 2.0, (7, 1)), b[:, 3:4], b[:, 4:]], 1)
    out['d3_boxes'], out['d3_qboxes'] = b3a, b3b
    for crit in (-1, 0, 1):
        out['d3_c%d' % crit] = ref_eval.d3_box_overlap(b3a, b3b, crit)

    with tempfile.TemporaryDirectory() as root:
        ids = kitti_synth.make_tree(root, n_images=40, seed=21, images=False, occ_choices=[0, 0, 0, 1, 2, 3])
        res = os.path.join(root, 'results')
        kitti_synth_dets.make_results(root, ids, res, seed=3)
        dt = ref_common.get_label_annos(res)
        gt = ref_common.get_label_annos(os.path.join(root, 'training', 'label_2'), [int(i) for i in ids])
        for cls in (0, 1, 2):
            text, ret, car_mod = ref_eval.get_official_eval_result(gt, dt, cls)
            out['cls%d_text' % cls] = np.array(text)
            out['cls%d_keys' % cls] = np.array(sorted(ret))
            out['cls%d_vals' % cls] = np.array([ret[k] for k in sorted(ret)], dtype=np.float64)
            out['cls%d_ap3d_r40_moderate' % cls] = np.array(car_mod)
            print(text)
        # precision curves of one configuration, for a finer comparison than the averaged APs
        ret = ref_eval.eval_class(gt, dt, [0], [0, 1, 2], 2, np.array([[[0.7], [0.7], [0.7]], [[0.7], [0.5], [0.5]]]), compute_aos=False)
        out['car_3d_precision'], out['car_3d_recall'] = ret['precision'], ret['recall']
        ret = ref_eval.eval_class(gt, dt, [0], [0, 1, 2], 0, np.array([[[0.7], [0.7], [0.7]], [[0.7], [0.5], [0.5]]]), compute_aos=True)
        out['car_bbox_precision'], out['car_bbox_aos'] = ret['precision'], ret['orientation']
    # ---- decode path: lib/helpers/decode_helper.py (extract_dets_from_outputs, decode_detections) ----
    from lib.helpers.decode_helper import decode_detections, extract_dets_from_outputs
    from lib.datasets.kitti.kitti_utils import Calibration
    outputs, p2, info = kitti_synth_dets.decode_problem()
    dets = extract_dets_from_outputs(outputs, K=50, topk=50).numpy()
    out['decode_dets'] = dets.copy()
    calibs = [Calibration({'P2': p, 'R0': np.eye(3, dtype=np.float32), 'Tr_velo2cam': np.zeros((3, 4), dtype=np.float32)}) for p in p2]
    res = decode_detections(dets.copy(), info, calibs, np.zeros((3, 3), dtype=np.float32), 0.2)
    for img_id, preds in res.items():
        out['decode_img%d' % img_id] = np.array(preds, dtype=np.float64).reshape(-1, 14)
        print('decoded', img_id, len(preds))
    np.savez_compressed(os.path.join(HERE, 'kitti_eval.npz'), **out)
    print('wrote', os.path.join(HERE, 'kitti_eval.npz'))


if __name__ == '__main__':
    main()
