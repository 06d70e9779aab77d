"""KITTI evaluation, CPU side: the oracle restatement against the fixture recorded from
the reference's own eval.py / rotate_iou.py device functions (tests/golden/make_kitti_eval_golden.py)."""
import os

import numpy as np
import pytest

import kitti_synth
import kitti_synth_dets
from oracle import kitti_eval as oke

GOLD = os.path.join(os.path.dirname(__file__), "golden", "kitti_eval.npz")


@pytest.fixture(scope="module")
def annos(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("kitti_eval"))
    ids = kitti_synth.make_tree(root, n_images=40, seed=21, images=False, occ_choices=[0, 0, 0, 1, 2, 3])
    res = os.path.join(root, 'results')
    kitti_synth_dets.make_results(root, ids, res, seed=3)
    return root, ids, res


def test_oracle_rotated_overlaps_match_the_reference_device_functions():
    g = np.load(GOLD)
    for crit in (-1, 0, 1, 2):
        got = oke.rotate_iou(g['riou_boxes'], g['riou_qboxes'], crit)
        assert np.array_equal(got, g['riou_c%d' % crit]), crit          # same float32 operations: bit-exact
    assert abs(oke.rotate_iou(g['riou_boxes'][:1], g['riou_qboxes'][:1])[0, 0] - 1.0) < 1e-6          # identical boxes
    assert np.all(g['riou_c-1'][:, 3] == 0)                                                            # the disjoint box
    for crit in (-1, 0, 1):
        got = oke.d3_box_overlap(g['d3_boxes'], g['d3_qboxes'], crit)
        assert np.array_equal(got, g['d3_c%d' % crit]), crit


def test_oracle_official_result_matches_the_reference_evaluation(annos):
    root, ids, res = annos
    g = np.load(GOLD)
    gt = [oke.read_label_file(os.path.join(root, 'training/label_2/%s.txt' % i)) for i in ids]
    dt = [oke.read_label_file(os.path.join(res, '%s.txt' % i)) for i in ids]
    for cls in (0, 1, 2):
        got, moderate = oke.official_result(gt, dt, cls)
        keys, vals = list(g['cls%d_keys' % cls]), g['cls%d_vals' % cls]
        assert sorted(got) == keys
        for k, v in zip(keys, vals):
            assert (np.isnan(v) and np.isnan(got[k])) or abs(got[k] - v) < 1e-9, (cls, k, got[k], v)
        assert abs(moderate - float(g['cls%d_ap3d_r40_moderate' % cls])) < 1e-9
    mo = np.array([[[0.7], [0.7], [0.7]], [[0.7], [0.5], [0.5]]])
    r = oke.eval_class(gt, dt, [0], [0, 1, 2], 2, mo)
    assert np.allclose(r['precision'], g['car_3d_precision'], atol=1e-12, equal_nan=True)
    assert np.allclose(r['recall'], g['car_3d_recall'], atol=1e-12, equal_nan=True)
    r = oke.eval_class(gt, dt, [0], [0, 1, 2], 0, mo, compute_aos=True)
    assert np.allclose(r['precision'], g['car_bbox_precision'], atol=1e-12, equal_nan=True)
    assert np.allclose(r['orientation'], g['car_bbox_aos'], atol=1e-12, equal_nan=True)


# ---- product: device kernel source on the CPU shim + the native statistics of the real library -----------------------
@pytest.fixture()
def emulated_overlaps():
    import native_emul
    from monodetr_amd.datasets.kitti.kitti_eval_python import rotate_iou
    rotate_iou._backend = native_emul.lib()
    yield rotate_iou
    rotate_iou._backend = None


def test_rotated_overlap_kernels_are_bit_identical_to_the_reference_device_functions(emulated_overlaps):
    g = np.load(GOLD)
    for crit in (-1, 0, 1, 2):
        got = emulated_overlaps.rotate_iou_gpu_eval(g['riou_boxes'], g['riou_qboxes'], crit)
        assert got.dtype == np.float64 and np.array_equal(got, g['riou_c%d' % crit]), crit
    for crit in (-1, 0, 1):
        got = emulated_overlaps.segmented_box3d_overlap([g['d3_boxes']], [g['d3_qboxes']], crit)[0]
        assert np.array_equal(got, g['d3_c%d' % crit]), crit
    # segmentation: three frames of different sizes (one empty) in one launch == frame-by-frame results
    a, b = g['riou_boxes'], g['riou_qboxes']
    parts = emulated_overlaps.segmented_rotate_iou([a[:4], a[4:4], a[4:]], [b[:2], b[2:5], b[5:]], -1)
    assert [p.shape for p in parts] == [(4, 2), (0, 3), (5, 2)]
    assert np.array_equal(parts[0], g['riou_c-1'][:4, :2].astype(np.float32)) and np.array_equal(parts[2], g['riou_c-1'][4:, 5:].astype(np.float32))
    assert emulated_overlaps.rotate_iou_gpu_eval(a[:0], b).shape == (0, 7)


def test_official_evaluation_matches_the_reference_report(annos, emulated_overlaps):
    from monodetr_amd.datasets.kitti.kitti_eval_python import eval as kitti_eval
    from monodetr_amd.datasets.kitti.kitti_eval_python import kitti_common
    root, ids, res = annos
    g = np.load(GOLD)
    dt = kitti_common.get_label_annos(res)
    gt = kitti_common.get_label_annos(os.path.join(root, 'training', 'label_2'), [int(i) for i in ids])
    for cls in (0, 1, 2):
        text, ret, moderate = kitti_eval.get_official_eval_result(gt, dt, cls)
        assert text == str(g['cls%d_text' % cls])                       # the report, character for character
        keys, vals = list(g['cls%d_keys' % cls]), g['cls%d_vals' % cls]
        assert sorted(ret) == keys
        for k, v in zip(keys, vals):
            assert (np.isnan(v) and np.isnan(ret[k])) or abs(ret[k] - v) < 1e-9, (cls, k, ret[k], v)
        assert abs(moderate - float(g['cls%d_ap3d_r40_moderate' % cls])) < 1e-12
    mo = np.array([[[0.7], [0.7], [0.7]], [[0.7], [0.5], [0.5]]])
    r = kitti_eval.eval_class(gt, dt, [0], [0, 1, 2], 2, mo)
    assert np.allclose(r['precision'], g['car_3d_precision'], atol=1e-12, equal_nan=True)
    assert np.allclose(r['recall'], g['car_3d_recall'], atol=1e-12, equal_nan=True)
    r = kitti_eval.eval_class(gt, dt, [0], [0, 1, 2], 0, mo, compute_aos=True)
    assert np.allclose(r['precision'], g['car_bbox_precision'], atol=1e-12, equal_nan=True)
    assert np.allclose(r['orientation'], g['car_bbox_aos'], atol=1e-9, equal_nan=True)
    # by class name, several classes at once
    text, ret, _ = kitti_eval.get_official_eval_result(gt, dt, ['Car', 'Cyclist'])
    assert 'Car_3d_moderate_R40' in ret and 'Cyclist_bev_hard' in ret and text.count('AP_R40@') == 4


def test_product_overlaps_refuse_to_run_without_a_gpu():
    from monodetr_amd.datasets.kitti.kitti_eval_python import rotate_iou
    assert rotate_iou._backend is None
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        rotate_iou.rotate_iou_gpu_eval(np.zeros((1, 5)), np.zeros((1, 5)))


def test_detection_decoding_matches_the_reference_helpers(tmp_path):
    """extract_dets_from_outputs + decode_detections against the arrays recorded from lib/helpers/decode_helper.py, and
    the result-file format of Tester.save_results read back by the evaluation's parser."""
    import logging
    from monodetr_amd.datasets.kitti.kitti_utils import Calibration
    from monodetr_amd.datasets.kitti.kitti_eval_python import kitti_common
    from monodetr_amd.helpers.decode_helper import decode_detections, extract_dets_from_outputs
    from monodetr_amd.helpers.tester_helper import Tester
    g = np.load(GOLD)
    outputs, p2, info = kitti_synth_dets.decode_problem()
    dets = extract_dets_from_outputs(outputs, K=50, topk=50).numpy()
    assert dets.shape == (2, 50, 37) and np.array_equal(dets, g['decode_dets'])
    calibs = [Calibration({'P2': p, 'R0': np.eye(3, dtype=np.float32), 'Tr_velo2cam': np.zeros((3, 4), dtype=np.float32)}) for p in p2]
    res = decode_detections(dets.copy(), info, calibs, np.zeros((3, 3), dtype=np.float32), 0.2)
    kept = 0
    for img_id, preds in res.items():
        want = g['decode_img%d' % img_id]
        kept += len(preds)
        assert len(preds) == len(want) and np.allclose(np.array(preds, dtype=np.float64).reshape(-1, 14), want, rtol=1e-6, atol=1e-6)
    assert 0 < kept < 100                                              # the score threshold removed some

    class _DS:
        max_objs, class_name = 50, ['Pedestrian', 'Car', 'Cyclist']
    class _DL:
        dataset = _DS()
    t = Tester({'type': 'KITTI'}, None, _DL(), logging.getLogger('t'), train_cfg={'save_path': str(tmp_path).lstrip('/')}, model_name='m')
    t.output_dir = str(tmp_path)
    t.save_results(res)
    back = kitti_common.get_label_annos(os.path.join(str(tmp_path), 'outputs', 'data'))
    assert len(back) == 2 and len(back[0]['name']) == len(res[1])
    assert set(back[0]['name']) <= {'Pedestrian', 'Car', 'Cyclist'}
    assert np.allclose(back[0]['score'], [round(p[-1], 2) for p in res[1]], atol=0.006)
    assert np.allclose(back[0]['dimensions'][:, [1, 2, 0]], [[round(v, 2) for v in p[6:9]] for p in res[1]], atol=0.006)   # file order h, w, l


def test_overlap_kernel_self_check_table_is_what_the_reference_functions_compute(emulated_overlaps):
    """The known-answer test that precedes the first evaluation of a process (rotate_iou.self_check): its table equals the
    restatement of the reference's device functions bit for bit, the kernel source reproduces it, and a kernel that
    computes anything else is refused."""
    from monodetr_amd.datasets.kitti.kitti_eval_python import rotate_iou as R
    for crit, want in R._KAT_BITS.items():
        ref = oke.rotate_iou(R._KAT_BOXES, R._KAT_QUERY, crit).astype(np.float32)
        assert ref.view(np.uint32).reshape(-1).tolist() == want, crit
    R._self_checked.clear()
    R.self_check(0)
    saved = R._KAT_BITS[0][1]
    R._KAT_BITS[0][1] ^= 1
    R._self_checked.clear()
    try:
        with pytest.raises(RuntimeError, match="known-answer"):
            R.self_check(0)
    finally:
        R._KAT_BITS[0][1] = saved
        R._self_checked.clear()
