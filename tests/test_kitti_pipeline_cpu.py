"""Input pipeline, CPU side: the oracle restatement against the fixture recorded from the
reference's own KITTI_Dataset class (tests/golden/make_kitti_golden.py), and PIL's affine/bilinear resampling
restated as array arithmetic (the form the device kernel uses) against PIL itself."""
import hashlib
import os

import numpy as np
import pytest
from PIL import Image

import kitti_synth
from oracle import kitti_pipeline as okp

GOLD = os.path.join(os.path.dirname(__file__), "golden", "kitti_pipeline.npz")
TARGET_KEYS = ['calibs', 'indices', 'img_size', 'labels', 'boxes', 'boxes_3d', 'depth', 'size_2d', 'size_3d',
               'src_size_3d', 'heading_bin', 'heading_res', 'mask_2d']


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("kitti"))
    ids = kitti_synth.make_tree(root, n_images=6, seed=7)
    return root, ids


def load_raw(root, idx):
    img = np.array(Image.open(os.path.join(root, 'training/image_2/%s.png' % idx)))
    labels = open(os.path.join(root, 'training/label_2/%s.txt' % idx)).readlines()
    calib = open(os.path.join(root, 'training/calib/%s.txt' % idx)).readlines()
    return img, labels, calib


def test_oracle_reproduces_the_reference_dataset_class_sample_for_sample(tree):
    root, ids = tree
    g = np.load(GOLD)
    n_aug = 0
    for n, seed in enumerate(g['seeds']):
        pre = 's%02d_' % n
        img, labels, calib = load_raw(root, ids[int(g[pre + 'item'])])
        np.random.seed(int(seed))
        inputs, p2, targets, params = okp.training_sample(img, labels, calib, augment=n < 10)
        n_aug += params['pd'] is not None and params['flip']
        inputs = np.ascontiguousarray(inputs, dtype=np.float32)
        assert inputs.shape == (3, 384, 1280)
        assert np.array_equal(inputs[:, 5::24, 7::40], g[pre + 'sub'])
        assert hashlib.sha256(inputs.tobytes()).digest() == g[pre + 'sha256'].tobytes()       # bit-exact image
        assert np.array_equal(p2, g[pre + 'p2'])
        for k in TARGET_KEYS:
            want = g[pre + 't_' + k]
            assert targets[k].dtype == want.dtype and np.array_equal(targets[k], want), (n, k)
    assert n_aug >= 2                                                  # the fixture exercises flip + distortion together


def test_fixture_covers_the_cases_that_matter():
    g = np.load(GOLD)
    kept = [int((g['s%02d_t_labels' % n] != 0).sum() + (g['s%02d_t_mask_2d' % n]).sum()) for n in range(len(g['seeds']))]
    assert max(kept) >= 2 and min(kept) == 0                           # images with several kept objects and with none
    sizes = {tuple(g['s%02d_img_size' % n]) for n in range(len(g['seeds']))}
    assert len(sizes) == 4                                             # all four KITTI image sizes


@pytest.mark.parametrize("seed", range(6))
def test_restated_pil_affine_bilinear_is_what_pil_computes(seed):
    rs = np.random.RandomState(seed)
    w, h = kitti_synth.SIZES[seed % 4]
    img = kitti_synth.synth_image(rs, w, h)
    np.random.seed(100 + seed)
    flip, center, crop_size, crop_scale = okp.draw_geometry(np.array([w, h]), random_crop=1.0, scale=0.4, shift=0.1)
    _, inv = okp.affine_pair(center, crop_size)
    if seed == 5:                                                      # a general matrix: rotation + shear, leaves the image
        inv = np.array([[0.93, 0.21, -40.3], [-0.17, 1.08, 25.9]])
    want = np.array(Image.fromarray(img).transform((1280, 384), method=Image.AFFINE, data=tuple(inv.reshape(-1).tolist()),
                                                   resample=Image.BILINEAR))
    got = okp.pil_affine_bilinear(img, inv.reshape(-1), 1280, 384)
    assert np.array_equal(got, want)


def test_hsv_round_trip_restatement_properties():
    rs = np.random.RandomState(0)
    img = rs.uniform(0, 255, size=(64, 96, 3)).astype(np.float32)
    img[:8] = img[:8, :, :1]                                           # grey rows: s = 0
    hsv = okp.bgr2hsv_f32(img)
    assert hsv[..., 0].min() >= 0 and hsv[..., 0].max() < 360.0 + 1e-3
    assert np.all(hsv[:8, :, 1] == 0) and np.allclose(hsv[..., 2], img.max(-1))
    back = okp.hsv2bgr_f32(hsv)
    assert np.abs(back - img).max() < 2e-3                             # float32 round trip of 0..255 values
    # pure colours land on the documented hues (channel 0 is "B")
    pure = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]]], dtype=np.float32)
    assert np.allclose(okp.bgr2hsv_f32(pure)[0, :, 0], [240.0, 120.0, 0.0], atol=1e-3)


def test_uint8_cast_is_the_wrapping_c_cast_the_reference_relies_on():
    x = np.array([300.7, -3.2, 255.9, 256.0, -0.5, 511.9], dtype=np.float32)
    with np.errstate(invalid='ignore'):
        assert np.array_equal(x.astype(np.uint8), x.astype(np.int32).astype(np.uint8))
    assert list(x.astype(np.int32).astype(np.uint8)) == [44, 253, 255, 0, 0, 255]
