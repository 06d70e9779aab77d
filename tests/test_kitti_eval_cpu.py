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
