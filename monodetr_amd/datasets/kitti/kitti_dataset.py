"""``KITTI_Dataset(split, cfg)`` -- mirror of ``lib/datasets/kitti/kitti_dataset.py`` with the per-pixel work moved
to the device.

Same constructor, configuration keys, file layout (``ImageSets/<split>.txt``, ``{training,testing}/{image_2,calib,
label_2}``), random decisions (numpy's global RNG, drawn in the reference's order, so a seed reproduces the
reference's sample) and target encoding.  The one difference: where the reference's ``__getitem__`` returns the
finished float32 ``[3, 384, 1280]`` image (kitti_dataset.py:127-163: distortion, flip, PIL warp, normalisation on
the worker), this one returns the DECODED image and an 88-byte descriptor of what to do with it:

    ({'pixels': uint8 [H, W, 3], 'descriptor': DESCRIPTOR[1]}, P2, targets, info)

``helpers/dataloader_helper.py`` batches those, ships 1.4 MB instead of 5.9 MB per image over PCIe and produces the
reference's ``inputs`` tensor on the GPU in one kernel launch (bit-identical, ``csrc/kitti_prep.hip``).
``KITTI_Dataset.eval`` runs the official evaluation of ``kitti_eval_python`` (device overlaps, native statistics).
"""
import os

import numpy as np
import torch.utils.data as data
from PIL import Image, ImageFile

from ... import kitti_prep_ext as prep
from ..utils import angle2class
from .kitti_utils import Calibration, affine_transform, get_affine_transform, get_objects_from_label

ImageFile.LOAD_TRUNCATED_IMAGES = True

_PERMS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))          # pd.py:146-148


def draw_photometric(desc):
    """The random decisions of ``PhotometricDistort.__call__`` (pd.py:376-398) in call order, written into the
    descriptor record instead of being applied to an image."""
    rnd = np.random
    flags = prep.DISTORT

    def maybe(lo, hi, bit, field):
        nonlocal flags
        if rnd.randint(2):
            desc[field] = rnd.uniform(lo, hi)
            flags |= bit

    maybe(-32, 32, prep.BRIGHTNESS, 'brightness')                  # RandomBrightness
    if rnd.randint(2):                                             # distortion order (pd.py:392-395)
        flags |= prep.CONTRAST_FIRST
        maybe(0.5, 1.5, prep.CONTRAST, 'contrast')
        maybe(0.5, 1.5, prep.SATURATION, 'saturation')
        maybe(-18.0, 18.0, prep.HUE, 'hue')
    else:
        maybe(0.5, 1.5, prep.SATURATION, 'saturation')
        maybe(-18.0, 18.0, prep.HUE, 'hue')
        maybe(0.5, 1.5, prep.CONTRAST, 'contrast')
    if rnd.randint(2):                                             # RandomLightingNoise
        p = _PERMS[rnd.randint(len(_PERMS))]
        desc['perm'] = p[0] | (p[1] << 2) | (p[2] << 4)
    desc['flags'] |= flags


class KITTI_Dataset(data.Dataset):
    # configuration keys of the `dataset:` section and their defaults (kitti_dataset.py:24-84)
    _OPTIONS = (('use_3d_center', True), ('bbox2d_type', 'anno'), ('meanshape', False), ('class_merging', False),
                ('use_dontcare', False), ('aug_pd', False), ('aug_crop', False), ('aug_calib', False), ('random_flip', 0.5),
                ('random_crop', 0.5), ('scale', 0.4), ('shift', 0.1), ('depth_scale', 'normal'), ('clip_2d', False))
    _MEAN_SIZE = ((1.76255119, 0.66068622, 0.84422524),            # h, w, l of Pedestrian / Car / Cyclist (meanshape)
                  (1.52563191462, 1.62856739989, 3.88311640418),
                  (1.73698127, 0.59706367, 1.76282397))

    def __init__(self, split, cfg):
        assert split in ['train', 'val', 'trainval', 'test']
        self.split, self.root_dir = split, cfg.get('root_dir')
        for key, default in self._OPTIONS:
            setattr(self, key, cfg.get(key, default))
        assert self.bbox2d_type in ['anno', 'proj']
        self.num_classes, self.max_objs, self.downsample = 3, 50, 32
        self.class_name = ['Pedestrian', 'Car', 'Cyclist']
        self.cls2id = {name: i for i, name in enumerate(self.class_name)}
        self.resolution = np.array([1280, 384])                    # W, H
        self.writelist = list(cfg.get('writelist', ['Car']))
        if self.class_merging:
            self.writelist.extend(['Van', 'Truck'])
        if self.use_dontcare:
            self.writelist.extend(['DontCare'])

        with open(os.path.join(self.root_dir, 'ImageSets', split + '.txt')) as f:
            self.idx_list = [x.strip() for x in f.readlines()]
        self.data_dir = os.path.join(self.root_dir, 'testing' if split == 'test' else 'training')
        self.image_dir, self.calib_dir, self.label_dir = (os.path.join(self.data_dir, d) for d in ('image_2', 'calib', 'label_2'))
        self.data_augmentation = split in ['train', 'trainval']

        self.mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
        self.std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
        self.cls_mean_size = np.array(self._MEAN_SIZE) if self.meanshape else np.zeros((3, 3), dtype=np.float32)

    # ---- files -------------------------------------------------------------------------------
    def get_image(self, idx):
        path = os.path.join(self.image_dir, '%06d.png' % idx)
        assert os.path.exists(path)
        return Image.open(path)

    def get_label(self, idx):
        path = os.path.join(self.label_dir, '%06d.txt' % idx)
        assert os.path.exists(path)
        return get_objects_from_label(path)

    def get_calib(self, idx):
        path = os.path.join(self.calib_dir, '%06d.txt' % idx)
        assert os.path.exists(path)
        return Calibration(path)

    def __len__(self):
        return len(self.idx_list)

    def eval(self, results_dir, logger):
        """Official KITTI evaluation of the result files in ``results_dir`` against this split's labels; logs the
        report per class of the write-list and returns the car 3-D AP|R40 at moderate difficulty
        (kitti_dataset.py:101-116)."""
        from .kitti_eval_python import kitti_common as kitti
        from .kitti_eval_python.eval import get_official_eval_result
        logger.info("==> Loading detections and GTs...")
        img_ids = [int(i) for i in self.idx_list]
        dt_annos = kitti.get_label_annos(results_dir)
        gt_annos = kitti.get_label_annos(self.label_dir, img_ids)
        test_id = {'Car': 0, 'Pedestrian': 1, 'Cyclist': 2}
        logger.info('==> Evaluating (official) ...')
        car_moderate = 0
        for category in self.writelist:
            results_str, results_dict, mAP3d_R40 = get_official_eval_result(gt_annos, dt_annos, test_id[category])
            if category == 'Car':
                car_moderate = mAP3d_R40
            logger.info(results_str)
        return car_moderate

    # ---- one sample --------------------------------------------------------------------------
    def __getitem__(self, item):
        index = int(self.idx_list[item])
        img = self.get_image(index)
        img_size = np.array(img.size)
        pixels = np.asarray(img.convert('RGB') if img.mode != 'RGB' else img, dtype=np.uint8)
        desc = np.zeros(1, dtype=prep.DESCRIPTOR)
        d = desc[0]
        d['width'], d['height'], d['perm'] = img_size[0], img_size[1], prep.IDENTITY_PERM

        center = np.array(img_size) / 2
        crop_size, crop_scale, flipped = img_size, 1, False
        if self.data_augmentation:                                 # kitti_dataset.py:133-152, same draw order
            if self.aug_pd:
                draw_photometric(d)
            if np.random.random() < self.random_flip:
                flipped = True
                d['flags'] |= prep.FLIP
            if self.aug_crop and np.random.random() < self.random_crop:
                crop_scale = np.clip(np.random.randn() * self.scale + 1, 1 - self.scale, 1 + self.scale)
                crop_size = img_size * crop_scale
                for axis in (0, 1):
                    center[axis] += img_size[axis] * np.clip(np.random.randn() * self.shift, -2 * self.shift, 2 * self.shift)
        trans, trans_inv = get_affine_transform(center, crop_size, 0, self.resolution, inv=1)
        d['inv'] = trans_inv.reshape(-1)
        image = {'pixels': pixels, 'descriptor': desc}
        info = {'img_id': index, 'img_size': img_size,
                'bbox_downsample_ratio': img_size / (self.resolution // self.downsample)}
        calib = self.get_calib(index)
        if self.split == 'test':
            return image, calib.P2, image, info

        objects = self.get_label(index)
        if flipped:
            if self.aug_calib:
                calib.flip(img_size)
            for obj in objects:
                obj.mirror(img_size[0])
                if self.aug_calib:
                    obj.pos[0] *= -1
        targets = self._encode(objects, calib, img_size, trans, crop_scale, flipped)
        return image, calib.P2, targets, info

    # ---- targets (kitti_dataset.py:192-312) ---------------------------------------------------
    def _keep(self, obj):
        return (obj.cls_type in self.writelist and obj.level_str != 'UnKnown' and 2 <= obj.pos[-1] <= 65)

    def _encode(self, objects, calib, img_size, trans, crop_scale, flipped):
        n = self.max_objs
        f32 = lambda *shape: np.zeros((n,) + shape, dtype=np.float32)
        t = {'calibs': f32(3, 4), 'indices': np.zeros(n, dtype=np.int64), 'img_size': img_size,
             'labels': np.zeros(n, dtype=np.int8), 'boxes': f32(4), 'boxes_3d': f32(6), 'depth': f32(1),
             'size_2d': f32(2), 'size_3d': f32(3), 'src_size_3d': f32(3),
             'heading_bin': np.zeros((n, 1), dtype=np.int64), 'heading_res': f32(1), 'mask_2d': np.zeros(n, dtype=bool)}
        res = self.resolution
        for i, obj in enumerate(objects[:n]):
            if not self._keep(obj):
                continue
            box = obj.box2d.copy()                                 # corners through the crop transform
            box[:2] = affine_transform(box[:2], trans)
            box[2:] = affine_transform(box[2:], trans)
            centre_2d = np.array([(box[0] + box[2]) / 2, (box[1] + box[3]) / 2], dtype=np.float32)
            centre_3d = (obj.pos + [0, -obj.h / 2, 0]).reshape(-1, 3)        # mid-height point of the 3-D box
            centre_3d = calib.rect_to_img(centre_3d)[0][0]
            if flipped and not self.aug_calib:
                centre_3d[0] = img_size[0] - centre_3d[0]
            centre_3d = affine_transform(centre_3d.reshape(-1), trans)
            if not (0 <= centre_3d[0] < res[0] and 0 <= centre_3d[1] < res[1]):
                continue                                           # projected centre left the canvas

            cls_id = self.cls2id[obj.cls_type]
            t['labels'][i] = cls_id
            t['size_2d'][i] = 1. * (box[2] - box[0]), 1. * (box[3] - box[1])
            centre_2d_n, size_2d_n = centre_2d / res, t['size_2d'][i] / res
            box[0:2] = box[0:2] / res
            box[2:4] = box[2:4] / res
            centre_3d_n = centre_3d / res
            left, right = centre_3d_n[0] - box[0], box[2] - centre_3d_n[0]
            top, bottom = centre_3d_n[1] - box[1], box[3] - centre_3d_n[1]
            if left < 0 or right < 0 or top < 0 or bottom < 0:
                if not self.clip_2d:
                    continue
                left, right, top, bottom = (np.clip(v, 0, 1) for v in (left, right, top, bottom))
            t['boxes'][i] = centre_2d_n[0], centre_2d_n[1], size_2d_n[0], size_2d_n[1]
            t['boxes_3d'][i] = centre_3d_n[0], centre_3d_n[1], left, right, top, bottom

            z = obj.pos[-1]
            t['depth'][i] = {'normal': z * crop_scale, 'inverse': z / crop_scale}.get(self.depth_scale, z)
            heading = calib.ry2alpha(obj.ry, (obj.box2d[0] + obj.box2d[2]) / 2)
            if heading > np.pi:
                heading -= 2 * np.pi
            if heading < -np.pi:
                heading += 2 * np.pi
            t['heading_bin'][i], t['heading_res'][i] = angle2class(heading)
            t['src_size_3d'][i] = np.array([obj.h, obj.w, obj.l], dtype=np.float32)
            t['size_3d'][i] = t['src_size_3d'][i] - self.cls_mean_size[cls_id]
            if obj.trucation <= 0.5 and obj.occlusion <= 2:
                t['mask_2d'][i] = 1
            t['calibs'][i] = calib.P2
        return t
