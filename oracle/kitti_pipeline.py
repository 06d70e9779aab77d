"""oracle/kitti_pipeline.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement (numpy + PIL) of the reference's training input pipeline, SURVEY.md section 8 row f3:

    lib/datasets/kitti/kitti_dataset.py:121-330   KITTI_Dataset.__getitem__
    lib/datasets/kitti/pd.py:112-225, 376-398     the photometric distortion chain
    lib/datasets/kitti/kitti_utils.py:13-52       Object3d (one label_2 line)
    lib/datasets/kitti/kitti_utils.py:118-330     calibration file, projection, flip
    lib/datasets/kitti/kitti_utils.py:332-391     get_affine_transform / affine_transform
    lib/datasets/utils.py:8-16                    angle2class

Third-party arithmetic the reference calls and this container does not have (opencv-python, unpinned in the
reference's requirements.txt) is restated from OpenCV's published algorithms:

    cv2.cvtColor(float32, COLOR_BGR2HSV / COLOR_HSV2BGR)  -> bgr2hsv_f32 / hsv2bgr_f32 (imgproc color_hsv: the
        scalar float path, hue range 360)
    cv2.getAffineTransform                                 -> get_affine_matrix (6x6 linear system in float64)

PARITY STATUS.  Pinned: everything computed by numpy / PIL / the reference's own Python -- tests/golden/
kitti_pipeline.npz is recorded by running the reference's KITTI_Dataset class itself on a synthetic KITTI tree
(tests/golden/make_kitti_golden.py), with cv2 / numba / skimage / torchvision stubbed because they are absent here.
UNPINNED against OpenCV itself: the two cv2 functions above -- the golden run routes them through this file's restatements.
What holds them instead (tests/test_kitti_pipeline_cpu.py): both HSV directions against the standard library's `colorsys` (an
independent implementation of the same hexcone model; float32 rounding), and the affine matrix by uniqueness -- three
non-collinear point pairs have exactly one affine map, the test checks that the matrix maps them (and a least-squares route
to the same matrix).  The last bits of OpenCV's own float evaluation order remain unchecked.

Everything is a function of (decoded image, label lines, calibration, the draws of numpy's global RNG); the draws are
made in the reference's order, so `np.random.seed(s)` reproduces the reference sample for sample.
"""
import math

import numpy as np
from PIL import Image

FLT_EPSILON = np.float32(1.1920929e-07)
RESOLUTION = np.array([1280, 384])                 # W, H   (kitti_dataset.py:32)
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
MAX_OBJS = 50
CLS2ID = {'Pedestrian': 0, 'Car': 1, 'Cyclist': 2}
PERMS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))      # pd.py:146-148


# ------------------------------------------------------------------------------------------------
# OpenCV restatements (float32 throughout, no fused multiply-add: numpy evaluates op by op)
# ------------------------------------------------------------------------------------------------
def bgr2hsv_f32(img):
    """cv2.cvtColor(img, COLOR_BGR2HSV) for float32 HxWx3: H in [0, 360), S = (V - min) / (|V| + eps), V = max."""
    img = np.asarray(img, dtype=np.float32)
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = (v - vmin).astype(np.float32)
    s = diff / (np.abs(v) + FLT_EPSILON)
    scale = np.float32(60.0) / (diff + FLT_EPSILON)
    h = np.where(v == r, (g - b) * scale,
                 np.where(v == g, (b - r) * scale + np.float32(120.0), (r - g) * scale + np.float32(240.0)))
    h = np.where(h < 0, h + np.float32(360.0), h).astype(np.float32)
    return np.stack([h, s.astype(np.float32), v], axis=-1)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def hsv2bgr_f32(img):
    """cv2.cvtColor(img, COLOR_HSV2BGR) for float32 HxWx3 (hue range 360)."""
    img = np.asarray(img, dtype=np.float32)
    h, s, v = img[..., 0], img[..., 1], img[..., 2]
    one = np.float32(1.0)
    hh = h * np.float32(6.0 / 360.0)
    # bring into [0, 6): the library loops "+= 6" / "-= 6"; one step suffices for the reference's |hue| < 720
    for _ in range(4):
        hh = np.where(hh < 0, hh + np.float32(6.0), hh)
        hh = np.where(hh >= 6, hh - np.float32(6.0), hh)
    hh = hh.astype(np.float32)
    sector = np.floor(hh).astype(np.int32)
    frac = (hh - sector.astype(np.float32)).astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    frac = np.where(bad, np.float32(0.0), frac)
    tab = np.stack([v, v * (one - s), v * (one - s * frac), v * (one - s * (one - frac))], axis=-1).astype(np.float32)
    idx = _SECTOR[sector]                                   # [..., 3] -> which tab entry is b, g, r
    out = np.take_along_axis(tab, idx, axis=-1)
    grey = (s == 0)[..., None]
    return np.where(grey, v[..., None], out).astype(np.float32)


def get_affine_matrix(src, dst):
    """cv2.getAffineTransform: the 2x3 float64 matrix mapping three float32 points src -> dst."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    a = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        a[i, 0:2], a[i, 2] = src[i], 1.0
        a[i + 3, 3:5], a[i + 3, 5] = src[i], 1.0
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


# ------------------------------------------------------------------------------------------------
# photometric distortion (pd.py:376-398), draws from numpy's global RNG in the reference's order
# ------------------------------------------------------------------------------------------------
def draw_photometric():
    """The random decisions of PhotometricDistort.__call__, in call order.  Returns a dict:
    brightness (float or None), contrast_first (bool), contrast / saturation / hue (float or None), perm (tuple or None)."""
    rnd = np.random
    p = {}
    p['brightness'] = rnd.uniform(-32, 32) if rnd.randint(2) else None              # pd.py:184-187
    p['contrast_first'] = bool(rnd.randint(2))                                     # pd.py:392
    if p['contrast_first']:
        p['contrast'] = rnd.uniform(0.5, 1.5) if rnd.randint(2) else None           # pd.py:171-174
    p['saturation'] = rnd.uniform(0.5, 1.5) if rnd.randint(2) else None             # pd.py:120-122
    p['hue'] = rnd.uniform(-18.0, 18.0) if rnd.randint(2) else None                 # pd.py:133-137
    if not p['contrast_first']:
        p['contrast'] = rnd.uniform(0.5, 1.5) if rnd.randint(2) else None
    p['perm'] = PERMS[rnd.randint(len(PERMS))] if rnd.randint(2) else None          # pd.py:150-154
    return p


def apply_photometric(img_u8, p):
    """float32 chain on an HxWx3 uint8 image, then the reference's `.astype(np.uint8)` (kitti_dataset.py:138-139):
    a C cast, i.e. truncation toward zero and wrap modulo 256 for values outside [0, 256) on x86-64."""
    im = np.asarray(img_u8).astype(np.float32)
    if p['brightness'] is not None:
        im = im + np.float32(p['brightness'])
    if p['contrast_first'] and p['contrast'] is not None:
        im = im * np.float32(p['contrast'])
    im = bgr2hsv_f32(im)
    if p['saturation'] is not None:
        im[..., 1] = im[..., 1] * np.float32(p['saturation'])
    if p['hue'] is not None:
        hch = im[..., 0] + np.float32(p['hue'])
        hch = np.where(hch > 360.0, hch - np.float32(360.0), hch)
        hch = np.where(hch < 0.0, hch + np.float32(360.0), hch)
        im[..., 0] = hch
    im = hsv2bgr_f32(im)
    if not p['contrast_first'] and p['contrast'] is not None:
        im = im * np.float32(p['contrast'])
    if p['perm'] is not None:
        im = im[:, :, list(p['perm'])]
    return im.astype(np.int32).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# geometry
# ------------------------------------------------------------------------------------------------
def affine_pair(center, crop_size, resolution=RESOLUTION):
    """get_affine_transform(center, crop_size, 0, resolution, inv=1) (kitti_utils.py:347-384), rot = 0."""
    scale = np.asarray(crop_size)
    if scale.ndim == 0:
        scale = np.array([scale, scale], dtype=np.float32)
    src_w, dst_w, dst_h = scale[0], resolution[0], resolution[1]
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0] = center
    src[1] = center + np.array([0.0, src_w * -0.5])        # get_dir with rot_rad = 0: sn = 0, cs = 1
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    for pts in (src, dst):
        d = pts[0] - pts[1]
        pts[2] = pts[1] + np.array([-d[1], d[0]], dtype=np.float32)
    return get_affine_matrix(src, dst), get_affine_matrix(dst, src)


def warp_point(pt, t):
    """affine_transform (kitti_utils.py:387-390): float32 point, float64 matrix."""
    return np.dot(t, np.array([pt[0], pt[1], 1.0], dtype=np.float32).T)[:2]


def draw_geometry(img_size, random_flip=0.5, random_crop=0.5, scale=0.05, shift=0.05, aug_crop=True):
    """kitti_dataset.py:141-152: flip flag, crop scale, crop centre."""
    center = np.array(img_size) / 2
    crop_size, crop_scale = img_size, 1
    flip = bool(np.random.random() < random_flip)
    if aug_crop and np.random.random() < random_crop:
        crop_scale = np.clip(np.random.randn() * scale + 1, 1 - scale, 1 + scale)
        crop_size = img_size * crop_scale
        center[0] += img_size[0] * np.clip(np.random.randn() * shift, -2 * shift, 2 * shift)
        center[1] += img_size[1] * np.clip(np.random.randn() * shift, -2 * shift, 2 * shift)
    return flip, center, crop_size, crop_scale


def warp_and_normalise(img_u8, flip, trans_inv, resolution=RESOLUTION):
    """flip, PIL affine warp with bilinear resampling, /255, (x - mean) / std, CHW (kitti_dataset.py:141-163)."""
    img = Image.fromarray(np.ascontiguousarray(img_u8))
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    img = img.transform(tuple(int(v) for v in resolution), method=Image.AFFINE,
                        data=tuple(trans_inv.reshape(-1).tolist()), resample=Image.BILINEAR)
    x = np.array(img).astype(np.float32) / 255.0
    x = (x - MEAN) / STD
    return x.transpose(2, 0, 1)


def pil_affine_bilinear(img_u8, a, out_w, out_h):
    """What PIL's Image.transform(AFFINE, BILINEAR) computes for an 8-bit RGB image, restated (libImaging
    Geometry.c: affine_transform + bilinear_filter32RGB): float64 arithmetic, output truncated to uint8, pixels
    whose source coordinate falls outside [0, w) x [0, h) are 0.  Checked against PIL itself in the tests; this
    is the form the device kernel implements."""
    img = np.asarray(img_u8)
    h, w, _ = img.shape
    xs = np.arange(out_w, dtype=np.float64)[None, :] + 0.5
    ys = np.arange(out_h, dtype=np.float64)[:, None] + 0.5
    xin = a[0] * xs + a[1] * ys + a[2]
    yin = a[3] * xs + a[4] * ys + a[5]
    inside = (xin >= 0.0) & (xin < w) & (yin >= 0.0) & (yin < h)
    xin, yin = xin - 0.5, yin - 0.5
    x0f, y0f = np.floor(xin), np.floor(yin)
    dx, dy = xin - x0f, yin - y0f
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    xa, xb = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    ya = np.clip(y0, 0, h - 1)
    yb_ok = (y0 + 1 >= 0) & (y0 + 1 < h)
    yb = np.clip(y0 + 1, 0, h - 1)
    out = np.zeros((out_h, out_w, 3), dtype=np.uint8)
    for c in range(3):
        ch = img[..., c].astype(np.float64)
        p00, p01 = ch[ya, xa], ch[ya, xb]
        v1 = p00 + (p01 - p00) * dx
        p10, p11 = ch[yb, xa], ch[yb, xb]
        v2 = np.where(yb_ok, p10 + (p11 - p10) * dx, v1)
        v = v1 + (v2 - v1) * dy
        out[..., c] = np.where(inside, v, 0.0).astype(np.uint8)
    return out


# ------------------------------------------------------------------------------------------------
# labels and calibration
# ------------------------------------------------------------------------------------------------
class LabelLine:
    """One line of a label_2 file (kitti_utils.py:13-52)."""

    def __init__(self, line):
        f = line.strip().split(' ')
        self.cls_type = f[0]
        self.truncation, self.occlusion, self.alpha = float(f[1]), float(f[2]), float(f[3])
        self.box2d = np.array([float(v) for v in f[4:8]], dtype=np.float32)
        self.h, self.w, self.l = float(f[8]), float(f[9]), float(f[10])
        self.pos = np.array([float(v) for v in f[11:14]], dtype=np.float32)
        self.ry = float(f[14])
        height = float(self.box2d[3]) - float(self.box2d[1]) + 1
        if self.truncation == -1:
            self.level = 'DontCare'
        elif height >= 40 and self.truncation <= 0.15 and self.occlusion <= 0:
            self.level = 'Easy'
        elif height >= 25 and self.truncation <= 0.3 and self.occlusion <= 1:
            self.level = 'Moderate'
        elif height >= 25 and self.truncation <= 0.5 and self.occlusion <= 2:
            self.level = 'Hard'
        else:
            self.level = 'UnKnown'


def read_p2(calib_lines):
    """kitti_utils.py:118-135: P2 is the third line of the calibration file, float32 3x4."""
    return np.array(calib_lines[2].strip().split(' ')[1:], dtype=np.float32).reshape(3, 4)


def project_rect_to_img(p2, pts):
    """Calibration.rect_to_img (kitti_utils.py:176-185) for float32 points [N,3]."""
    hom = np.hstack((pts, np.ones((pts.shape[0], 1), dtype=np.float32)))
    q = np.dot(hom, p2.T)
    return (q[:, 0:2].T / hom[:, 2]).T


def flip_p2(p2, img_size):
    """Calibration.flip (kitti_utils.py:286-326): P2 of the mirrored image, re-fitted by SVD through 8 points."""
    cu, cv, fu, fv = p2[0, 2], p2[1, 2], p2[0, 0], p2[1, 1]
    tx, ty = p2[0, 3] / (-fu), p2[1, 3] / (-fv)
    wsize, hsize = 4, 2
    p2ds = np.concatenate([np.expand_dims(np.tile(np.expand_dims(np.linspace(0, img_size[0], wsize), 0), [hsize, 1]), -1),
                           np.expand_dims(np.tile(np.expand_dims(np.linspace(0, img_size[1], hsize), 1), [1, wsize]), -1),
                           np.linspace(2, 78, wsize * hsize).reshape(hsize, wsize, 1)], -1).reshape(-1, 3)
    u, v, d = p2ds[:, 0:1], p2ds[:, 1:2], p2ds[:, 2:3]
    x = ((u - cu) * d) / fu + tx
    y = ((v - cv) * d) / fv + ty
    p3ds = np.concatenate((x.reshape(-1, 1), y.reshape(-1, 1), d.reshape(-1, 1)), axis=1)
    p3ds[:, 0] *= -1
    p2ds[:, 0] = img_size[0] - p2ds[:, 0]
    m = np.zeros([wsize * hsize, 2, 7])
    m[:, 0, 0] = p3ds[:, 0]
    m[:, 0, 1] = m[:, 1, 2] = p3ds[:, 2]
    m[:, 1, 0] = p3ds[:, 1]
    m[:, 0, 3] = m[:, 1, 4] = 1
    m[:, :, -2] = -p2ds[:, :2]
    m[:, :, -1] = (-p2ds[:, :2] * p3ds[:, 2:3])
    sol = np.linalg.svd(m.reshape(-1, 7))[-1][-1]
    sol /= sol[-1]
    new = np.zeros([4, 3]).astype(np.float32)
    new[0, 0] = new[1, 1] = sol[0]
    new[2, 0:2] = sol[1:3]
    new[3, :] = sol[3:6]
    new[-1, -1] = p2[-1, -1]
    return new.T


def ry_to_alpha(p2, ry, u):
    """Calibration.ry2alpha (kitti_utils.py:276-284)."""
    alpha = ry - np.arctan2(u - p2[0, 2], p2[0, 0])
    if alpha > np.pi:
        alpha -= 2 * np.pi
    if alpha < -np.pi:
        alpha += 2 * np.pi
    return alpha


def angle_to_class(angle, bins=12):
    """lib/datasets/utils.py:8-16."""
    angle = angle % (2 * np.pi)
    per = 2 * np.pi / float(bins)
    shifted = (angle + per / 2) % (2 * np.pi)
    cid = int(shifted / per)
    return cid, shifted - (cid * per + per / 2)


def encode_targets(objects, p2, img_size, flip, trans, crop_scale, writelist=('Car',), clip_2d=False,
                   depth_scale='normal', resolution=RESOLUTION, aug_calib=False):
    """kitti_dataset.py:173-312: the 13 target arrays of one sample (aug_calib is off in the shipped config).
    `objects` are LabelLine objects; they are modified in place by the flip exactly as the reference does.
    Returns (targets, P2) -- P2 changes when aug_calib re-fits it for a flipped image."""
    if flip:                                                                       # :177-190
        if aug_calib:
            p2 = flip_p2(p2, img_size)
        for o in objects:
            if aug_calib:
                o.pos[0] *= -1
            x1, x2 = o.box2d[0], o.box2d[2]
            o.box2d[0], o.box2d[2] = img_size[0] - x2, img_size[0] - x1
            o.alpha = np.pi - o.alpha
            o.ry = np.pi - o.ry
            if o.alpha > np.pi:
                o.alpha -= 2 * np.pi
            if o.alpha < -np.pi:
                o.alpha += 2 * np.pi
            if o.ry > np.pi:
                o.ry -= 2 * np.pi
            if o.ry < -np.pi:
                o.ry += 2 * np.pi
    t = {
        'calibs': np.zeros((MAX_OBJS, 3, 4), dtype=np.float32), 'indices': np.zeros(MAX_OBJS, dtype=np.int64),
        'img_size': img_size, 'labels': np.zeros(MAX_OBJS, dtype=np.int8),
        'boxes': np.zeros((MAX_OBJS, 4), dtype=np.float32), 'boxes_3d': np.zeros((MAX_OBJS, 6), dtype=np.float32),
        'depth': np.zeros((MAX_OBJS, 1), dtype=np.float32), 'size_2d': np.zeros((MAX_OBJS, 2), dtype=np.float32),
        'size_3d': np.zeros((MAX_OBJS, 3), dtype=np.float32), 'src_size_3d': np.zeros((MAX_OBJS, 3), dtype=np.float32),
        'heading_bin': np.zeros((MAX_OBJS, 1), dtype=np.int64), 'heading_res': np.zeros((MAX_OBJS, 1), dtype=np.float32),
        'mask_2d': np.zeros(MAX_OBJS, dtype=bool),
    }
    for i in range(min(len(objects), MAX_OBJS)):
        o = objects[i]
        if o.cls_type not in writelist or o.level == 'UnKnown' or o.pos[-1] < 2 or o.pos[-1] > 65:
            continue
        bbox = o.box2d.copy()
        bbox[:2] = warp_point(bbox[:2], trans)
        bbox[2:] = warp_point(bbox[2:], trans)
        center_2d = np.array([(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2], dtype=np.float32)
        c3 = (o.pos + [0, -o.h / 2, 0]).reshape(-1, 3)
        c3 = project_rect_to_img(p2, c3)[0]
        if flip and not aug_calib:
            c3[0] = img_size[0] - c3[0]
        c3 = warp_point(c3.reshape(-1), trans)
        if c3[0] < 0 or c3[0] >= resolution[0] or c3[1] < 0 or c3[1] >= resolution[1]:
            continue
        t['labels'][i] = CLS2ID[o.cls_type]
        w, h = bbox[2] - bbox[0], bbox[3] - bbox[1]
        t['size_2d'][i] = 1. * w, 1. * h
        center_2d_norm = center_2d / resolution
        size_2d_norm = t['size_2d'][i] / resolution
        corner = bbox
        corner[0:2] = corner[0:2] / resolution
        corner[2:4] = corner[2:4] / resolution
        c3n = c3 / resolution
        l, r = c3n[0] - corner[0], corner[2] - c3n[0]
        tt, b = c3n[1] - corner[1], corner[3] - c3n[1]
        if l < 0 or r < 0 or tt < 0 or b < 0:
            if clip_2d:
                l, r, tt, b = (np.clip(v, 0, 1) for v in (l, r, tt, b))
            else:
                continue
        t['boxes'][i] = center_2d_norm[0], center_2d_norm[1], size_2d_norm[0], size_2d_norm[1]
        t['boxes_3d'][i] = c3n[0], c3n[1], l, r, tt, b
        if depth_scale == 'normal':
            t['depth'][i] = o.pos[-1] * crop_scale
        elif depth_scale == 'inverse':
            t['depth'][i] = o.pos[-1] / crop_scale
        else:
            t['depth'][i] = o.pos[-1]
        heading = ry_to_alpha(p2, o.ry, (o.box2d[0] + o.box2d[2]) / 2)
        if heading > np.pi:
            heading -= 2 * np.pi
        if heading < -np.pi:
            heading += 2 * np.pi
        t['heading_bin'][i], t['heading_res'][i] = angle_to_class(heading)
        t['src_size_3d'][i] = np.array([o.h, o.w, o.l], dtype=np.float32)
        t['size_3d'][i] = t['src_size_3d'][i]                                     # meanshape off: mean size is 0
        if o.truncation <= 0.5 and o.occlusion <= 2:
            t['mask_2d'][i] = 1
        t['calibs'][i] = p2
    return t, p2


def training_sample(img_u8, label_lines, calib_lines, aug_pd=True, aug_crop=True, random_flip=0.5, random_crop=0.5,
                    scale=0.05, shift=0.05, augment=True, aug_calib=False):
    """One `__getitem__` of the train split: (inputs [3,384,1280] float32, P2, targets dict, params dict)."""
    img_u8 = np.asarray(img_u8)
    img_size = np.array([img_u8.shape[1], img_u8.shape[0]])
    params = {'pd': None, 'flip': False}
    flip, center, crop_size, crop_scale = False, np.array(img_size) / 2, img_size, 1
    if augment:
        if aug_pd:
            params['pd'] = draw_photometric()
            img_u8 = apply_photometric(img_u8, params['pd'])
        flip, center, crop_size, crop_scale = draw_geometry(img_size, random_flip, random_crop, scale, shift, aug_crop)
    trans, trans_inv = affine_pair(center, crop_size)
    params.update(flip=flip, center=center, crop_size=crop_size, crop_scale=crop_scale, trans=trans, trans_inv=trans_inv)
    inputs = warp_and_normalise(img_u8, flip, trans_inv)
    p2 = read_p2(calib_lines)
    objects = [LabelLine(l) for l in label_lines]
    targets, p2 = encode_targets(objects, p2, img_size, flip, trans, crop_scale, aug_calib=aug_calib)
    return inputs, p2, targets, params
