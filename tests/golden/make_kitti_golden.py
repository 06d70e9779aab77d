"""Records tests/golden/kitti_pipeline.npz by running the REFERENCE's own dataset class
(/root/reference/lib/datasets/kitti/kitti_dataset.py: KITTI_Dataset.__getitem__) on the synthetic tree of
tests/kitti_synth.py.  Run in the build container only (the reference is not on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_kitti_golden.py

cv2, numba, skimage and torchvision are absent here and are stubbed: numba / skimage / torchvision are imported by
the reference at module level only; cv2's two functions on this path (cvtColor float32 BGR<->HSV,
getAffineTransform) are routed to oracle/kitti_pipeline.py's restatements -- so the fixture pins everything the
reference computes with numpy / PIL / its own Python, and NOT those two functions (stated in the oracle's header).

Per sample the fixture stores: the numpy seed, a SHA-256 of the float32 [3,384,1280] input, a strided sub-sample
of it, P2 and the 13 target arrays.
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from oracle import kitti_pipeline as okp          # noqa: E402
import kitti_synth                                # noqa: E402

SEEDS = [444, 445, 446, 447, 448, 449, 450, 451, 452, 453, 454, 455, 456, 457, 458, 459]
N_TRAIN, N_VAL = 10, 2          # then: train split with aug_calib (P2 re-fitted for flipped images)


def install_stubs():
    cv2 = types.ModuleType('cv2')
    cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 40, 54

    def cvt(img, code):
        assert img.dtype == np.float32
        return okp.bgr2hsv_f32(img) if code == cv2.COLOR_BGR2HSV else okp.hsv2bgr_f32(img)
    cv2.cvtColor = cvt
    cv2.getAffineTransform = okp.get_affine_matrix
    sys.modules['cv2'] = cv2
    numba = types.ModuleType('numba')
    numba.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    numba.cuda = types.ModuleType('numba.cuda')
    numba.cuda.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules['numba'], sys.modules['numba.cuda'] = numba, numba.cuda
    sk = types.ModuleType('skimage')
    sk.io = types.ModuleType('skimage.io')
    sys.modules['skimage'], sys.modules['skimage.io'] = sk, sk.io
    tv = types.ModuleType('torchvision')
    tv.transforms = types.ModuleType('torchvision.transforms')
    sys.modules['torchvision'], sys.modules['torchvision.transforms'] = tv, tv.transforms


def main():
    install_stubs()
    sys.path.insert(0, '/root/reference')
    from lib.datasets.kitti.kitti_dataset import KITTI_Dataset
    out = {}
    with tempfile.TemporaryDirectory() as root:
        ids = kitti_synth.make_tree(root, n_images=6, seed=7)
        cfg = {'root_dir': root, 'aug_pd': True, 'aug_crop': True, 'random_flip': 0.5, 'random_crop': 0.5,
               'scale': 0.05, 'shift': 0.05, 'writelist': ['Car'], 'depth_scale': 'normal'}
        train = KITTI_Dataset('train', cfg)
        val = KITTI_Dataset('val', cfg)
        train_calib = KITTI_Dataset('train', dict(cfg, aug_calib=True, random_flip=0.8))
        out['seeds'] = np.array(SEEDS)
        for n, seed in enumerate(SEEDS):
            item = n % len(ids)
            ds = train if n < N_TRAIN else (val if n < N_TRAIN + N_VAL else train_calib)
            np.random.seed(seed)
            inputs, p2, targets, info = ds[item]
            inputs = np.ascontiguousarray(inputs, dtype=np.float32)
            pre = 's%02d_' % n
            out[pre + 'item'] = np.array(item)
            out[pre + 'sha256'] = np.frombuffer(hashlib.sha256(inputs.tobytes()).digest(), dtype=np.uint8)
            out[pre + 'sub'] = inputs[:, 5::24, 7::40].copy()
            out[pre + 'p2'] = np.asarray(p2)
            for k, v in targets.items():
                out[pre + 't_' + k] = np.asarray(v)
            out[pre + 'img_size'] = np.asarray(info['img_size'])
    np.savez_compressed(os.path.join(HERE, 'kitti_pipeline.npz'), **out)
    print('wrote', os.path.join(HERE, 'kitti_pipeline.npz'), os.path.getsize(os.path.join(HERE, 'kitti_pipeline.npz')), 'bytes')


if __name__ == '__main__':
    main()
