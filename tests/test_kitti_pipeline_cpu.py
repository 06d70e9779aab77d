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
        inputs, p2, targets, params = okp.training_sample(img, labels, calib, augment=not 10 <= n < 12, aug_calib=n >= 12,
                                                          random_flip=0.8 if n >= 12 else 0.5)
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


def test_hsv_restatements_against_the_standard_library_colorsys():
    """cv2 is absent, so the float HSV conversions are restated from OpenCV's documented formulas; Python's `colorsys` is an
    independent implementation of the same hexcone model (hue as a fraction of a turn instead of degrees, no epsilon in the
    denominators): both directions agree to float32 rounding on random colours, greys, ties and pure colours."""
    import colorsys
    rs = np.random.RandomState(3)
    bgr = rs.uniform(0, 255, size=(400, 3)).astype(np.float32)
    bgr[:20] = bgr[:20, :1]                                            # greys
    bgr[20:40, 1] = bgr[20:40, 2]                                      # ties between two channels
    bgr[40:43] = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255]], dtype=np.float32)
    hsv = okp.bgr2hsv_f32(bgr[None])[0]
    for (b, g, r), (h, s_, v) in zip(bgr.astype(np.float64), hsv.astype(np.float64)):
        hc, sc, vc = colorsys.rgb_to_hsv(r, g, b)
        dh = abs(h - hc * 360.0)
        assert min(dh, 360.0 - dh) <= 2e-3 and abs(s_ - sc) <= 2e-6 and v == vc, ((b, g, r), (h, s_, v), (hc * 360, sc, vc))
    # the other direction on arbitrary (hue, saturation, value) triples, hue outside [0, 360) included
    hsv_in = np.stack([rs.uniform(-300, 700, 400), rs.uniform(0, 1, 400), rs.uniform(0, 255, 400)], -1).astype(np.float32)
    back = okp.hsv2bgr_f32(hsv_in[None])[0]
    for (h, s_, v), (b, g, r) in zip(hsv_in.astype(np.float64), back.astype(np.float64)):
        rc, gc, bc = colorsys.hsv_to_rgb((h / 360.0) % 1.0, s_, v)
        assert max(abs(r - rc), abs(g - gc), abs(b - bc)) <= 2e-3 * max(1.0, v), ((h, s_, v), (b, g, r), (bc, gc, rc))


def test_affine_matrix_restatement_is_the_unique_map_of_its_three_points():
    """cv2.getAffineTransform, restated: three non-collinear point pairs determine ONE affine map, so mapping the points is the
    whole specification (the inputs are float32, as cv2 requires; the solve is float64, as the library's)."""
    rs = np.random.RandomState(5)
    for _ in range(50):
        src = rs.uniform(-500, 1500, size=(3, 2)).astype(np.float32)
        dst = rs.uniform(-500, 1500, size=(3, 2)).astype(np.float32)
        if abs(np.linalg.det(np.c_[src.astype(np.float64), np.ones(3)])) < 1.0:
            continue
        m = okp.get_affine_matrix(src, dst)
        assert m.shape == (2, 3) and m.dtype == np.float64
        mapped = src.astype(np.float64) @ m[:, :2].T + m[:, 2]
        assert np.abs(mapped - dst.astype(np.float64)).max() <= 1e-8
        # ... and an independent route to the same matrix: least squares on homogeneous coordinates
        ls = np.linalg.lstsq(np.c_[src.astype(np.float64), np.ones(3)], dst.astype(np.float64), rcond=None)[0].T
        assert np.abs(ls - m).max() <= 1e-8 * max(1.0, np.abs(m).max())


def test_uint8_cast_is_the_wrapping_c_cast_the_reference_relies_on():
    x = np.array([300.7, -3.2, 255.9, 256.0, -0.5, 511.9], dtype=np.float32)
    with np.errstate(invalid='ignore'):
        assert np.array_equal(x.astype(np.uint8), x.astype(np.int32).astype(np.uint8))
    assert list(x.astype(np.int32).astype(np.uint8)) == [44, 253, 255, 0, 0, 255]


# ---- product host side + the kernel arithmetic (host build) ------------------------------------------------------
CFG = {'type': 'KITTI', 'aug_pd': True, 'aug_crop': True, 'random_flip': 0.5, 'random_crop': 0.5, 'scale': 0.05,
       'shift': 0.05, 'writelist': ['Car'], 'depth_scale': 'normal', 'train_split': 'train', 'test_split': 'val',
       'batch_size': 3}


@pytest.fixture(params=["host", "emul"])
def host_backend(request):
    import backends
    from monodetr_amd import kitti_prep_ext
    kitti_prep_ext._backend = backends.get(request.param)
    yield kitti_prep_ext
    kitti_prep_ext._backend = None


def run_host(prep, image, dtype=None):
    import torch
    from monodetr_amd.helpers.dataloader_helper import pack_images
    packed, n = pack_images([image])
    head = n * prep.DESCRIPTOR.itemsize
    return prep.preprocess_batch(packed[head:], packed[:head], dtype=dtype or torch.float32)


def test_dataset_and_kernel_arithmetic_reproduce_the_reference_sample_for_sample(tree, host_backend):
    """The product path -- KITTI_Dataset's draws / descriptor / targets on the host and the device kernel's
    per-pixel arithmetic (here: its host build) -- against the fixture recorded from the reference class:
    images bit-exact (SHA-256), every target array equal."""
    from monodetr_amd.datasets.kitti import KITTI_Dataset
    root, ids = tree
    g = np.load(GOLD)
    cfg = dict(CFG, root_dir=root)
    train, val = KITTI_Dataset('train', cfg), KITTI_Dataset('val', cfg)
    train_calib = KITTI_Dataset('train', dict(cfg, aug_calib=True, random_flip=0.8))
    seen_flags, refits = 0, 0
    for n, seed in enumerate(g['seeds']):
        pre = 's%02d_' % n
        np.random.seed(int(seed))
        image, p2, targets, info = (train if n < 10 else (val if n < 12 else train_calib))[int(g[pre + 'item'])]
        seen_flags |= int(image['descriptor']['flags'][0])
        refits += n >= 12 and bool(image['descriptor']['flags'][0] & 1)
        out = run_host(host_backend, image)[0].numpy()
        assert np.array_equal(out[:, 5::24, 7::40], g[pre + 'sub'])
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest() == g[pre + 'sha256'].tobytes()
        assert np.array_equal(p2, g[pre + 'p2']) and np.array_equal(info['img_size'], g[pre + 'img_size'])
        for k in TARGET_KEYS:
            want = g[pre + 't_' + k]
            assert targets[k].dtype == want.dtype and np.array_equal(targets[k], want), (n, k)
    assert seen_flags == 127                                            # every stage of the chain was exercised
    assert refits >= 2                                                  # aug_calib: P2 re-fitted for flipped images


@pytest.mark.parametrize("flags_off", [0, 8, 16, 32, 64, 4, 2])
def test_kernel_arithmetic_matches_the_oracle_for_each_stage_combination(host_backend, flags_off):
    """Stage by stage (one switched off at a time, both contrast positions, every channel permutation, extreme
    parameters that drive values through the uint8 wrap-around) on a small image, vs the numpy/PIL oracle."""
    prep = host_backend
    rs = np.random.RandomState(flags_off)
    img = kitti_synth.synth_image(rs, 200, 90)
    for trial in range(6):
        d = np.zeros(1, dtype=prep.DESCRIPTOR)
        flags = 127 & ~flags_off
        if trial % 2:
            flags ^= prep.CONTRAST_FIRST
        perm = okp.PERMS[trial]
        pd = {'brightness': rs.uniform(-32, 32), 'contrast': rs.uniform(0.5, 1.5), 'saturation': rs.uniform(0.5, 1.5),
              'hue': rs.uniform(-18, 18), 'perm': perm, 'contrast_first': bool(flags & prep.CONTRAST_FIRST)}
        if trial == 5:
            pd.update(brightness=32.0, contrast=1.5, hue=18.0)
        for name, bit in (('brightness', 8), ('contrast', 16), ('saturation', 32), ('hue', 64)):
            d[name] = pd[name]
            if not flags & bit:
                pd[name] = None
        inv = np.array([[0.16 + 0.01 * trial, 0.004, -3.0 + trial], [-0.003, 0.23, -2.5]])
        d['width'], d['height'], d['flags'], d['inv'] = 200, 90, flags, inv.reshape(-1)
        d['perm'] = perm[0] | (perm[1] << 2) | (perm[2] << 4)
        src = okp.apply_photometric(img, pd) if flags & prep.DISTORT else img
        want = okp.warp_and_normalise(src, bool(flags & prep.FLIP), inv)
        got = run_host(prep, {'pixels': img, 'descriptor': d})[0].numpy()
        assert np.array_equal(got, want), (flags, trial)


def test_bf16_output_is_the_rounded_float32_output(host_backend):
    import torch
    rs = np.random.RandomState(3)
    img = kitti_synth.synth_image(rs, 160, 80)
    d = np.zeros(1, dtype=host_backend.DESCRIPTOR)
    d['width'], d['height'], d['perm'], d['inv'] = 160, 80, host_backend.IDENTITY_PERM, [0.125, 0, 0, 0, 0.2083, 0]
    f32 = run_host(host_backend, {'pixels': img, 'descriptor': d})
    bf = run_host(host_backend, {'pixels': img, 'descriptor': d}, dtype=torch.bfloat16)
    assert bf.dtype == torch.bfloat16 and torch.equal(bf, f32.to(torch.bfloat16))


def test_loader_batches_like_the_reference_loader(tree, host_backend):
    """build_dataloader: the 4-tuple of the reference's loop (trainer_helper.py:122), ragged image sizes in one
    batch, workers, and the test split's repeated image."""
    import torch
    from monodetr_amd.helpers.dataloader_helper import build_dataloader
    root, ids = tree
    cfg = dict(CFG, root_dir=root)
    train, val = build_dataloader(cfg, workers=0, device='cpu')
    assert len(train) == 2 and len(val) == 2
    batches = list(val)                                                 # no augmentation, no shuffling: deterministic
    inputs, calibs, targets, info = batches[0]
    assert inputs.shape == (3, 3, 384, 1280) and inputs.dtype == torch.float32
    assert calibs.shape == (3, 3, 4) and targets['boxes_3d'].shape == (3, 50, 6) and targets['labels'].dtype == torch.int8
    assert info['img_id'].tolist() == [int(i) for i in ids[:3]] and info['img_size'].shape == (3, 2)
    for b in range(3):                                                  # each image equals its single-sample result
        image, _, t, _ = val.dataset[b]
        assert torch.equal(inputs[b], run_host(host_backend, image)[0])
        assert np.array_equal(targets['depth'][b].numpy(), t['depth'])
    # worker processes: same batches
    _, val2 = build_dataloader(cfg, workers=2, device='cpu')
    for (a, _, ta, _), (b, _, tb, _) in zip(batches, val2):
        assert torch.equal(a, b) and torch.equal(ta['boxes'], tb['boxes'])
    # test split
    from monodetr_amd.datasets.kitti import KITTI_Dataset
    from monodetr_amd.helpers.dataloader_helper import DeviceLoader, collate_packed
    dl = torch.utils.data.DataLoader(KITTI_Dataset('test', dict(cfg, root_dir=root)) if os.path.isdir(os.path.join(root, 'testing'))
                                     else _as_test_split(KITTI_Dataset('val', cfg)), batch_size=2, collate_fn=collate_packed)
    x, calib, again, info = next(iter(DeviceLoader(dl, 'cpu')))
    assert again is x and x.shape == (2, 3, 384, 1280)


def _as_test_split(ds):
    ds.split = 'test'                                                   # same files, the test split's return convention
    return ds


def test_product_refuses_to_run_without_the_device_library():
    import torch
    from monodetr_amd import kitti_prep_ext
    assert kitti_prep_ext._backend is None
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        kitti_prep_ext.preprocess_batch(torch.zeros(30, dtype=torch.uint8), torch.zeros(88, dtype=torch.uint8))


def test_install_aliases_the_reference_module_names_of_the_pipeline():
    import subprocess
    import sys
    code = ("import monodetr_amd; monodetr_amd.install();"
            "from lib.helpers.dataloader_helper import build_dataloader;"
            "from lib.datasets.kitti.kitti_dataset import KITTI_Dataset;"
            "from lib.datasets.kitti.kitti_utils import Calibration, get_affine_transform, affine_transform, get_objects_from_label;"
            "from lib.datasets.utils import angle2class, class2angle; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stderr


def test_channels_last_output_holds_the_same_values(host_backend):
    import torch
    from monodetr_amd.helpers.dataloader_helper import pack_images
    rs = np.random.RandomState(8)
    images = []
    for k, (w, h) in enumerate([(200, 90), (161, 77)]):
        d = np.zeros(1, dtype=host_backend.DESCRIPTOR)
        d['width'], d['height'], d['flags'], d['perm'] = w, h, 127, [0x24, 0x06][k]
        d['brightness'], d['contrast'], d['saturation'], d['hue'] = 11.0, 1.2, 0.8, -7.0
        d['inv'] = [w / 128.0, 0.01, -1, 0.002, h / 48.0, -0.5]
        images.append({'pixels': kitti_synth.synth_image(rs, w, h), 'descriptor': d})
    packed, n = pack_images(images)
    head = n * host_backend.DESCRIPTOR.itemsize
    for dtype in (torch.float32, torch.bfloat16):
        a = host_backend.preprocess_batch(packed[head:], packed[:head], out_hw=(48, 128), dtype=dtype)
        b = host_backend.preprocess_batch(packed[head:], packed[:head], out_hw=(48, 128), dtype=dtype, channels_last=True)
        assert b.shape == a.shape and b.is_contiguous(memory_format=torch.channels_last) and not b.is_contiguous()
        assert torch.equal(a, b)


def test_self_check_constant_is_what_the_reference_chain_computes(host_backend):
    """The known-answer test the DeviceLoader runs before its first batch (kitti_prep_ext.self_check): its constant is the
    hash of what the numpy / PIL oracle of the reference chain produces for the fixed input, the kernel arithmetic
    reproduces it, and a kernel that computes anything else is refused."""
    import torch
    prep = host_backend
    img, d = prep.self_check_inputs()
    pd = {'brightness': 11.5, 'contrast': 1.25, 'saturation': 0.75, 'hue': -9.0, 'perm': (1, 2, 0), 'contrast_first': False}
    want = okp.warp_and_normalise(okp.apply_photometric(img, pd), True, np.array(d['inv'][0]).reshape(2, 3), resolution=(40, 24))
    assert hashlib.sha256(np.ascontiguousarray(want[None]).tobytes()).hexdigest() == prep._SELF_CHECK_SHA256
    prep._self_checked.clear()
    prep.self_check("cpu")                                   # (through the substituted backend)
    assert torch.device("cpu") in prep._self_checked
    prep._self_checked.clear()
    saved = prep._SELF_CHECK_SHA256
    prep._SELF_CHECK_SHA256 = "00" * 32
    try:
        with pytest.raises(RuntimeError, match="known-answer"):
            prep.self_check("cpu")
    finally:
        prep._SELF_CHECK_SHA256 = saved
        prep._self_checked.clear()
