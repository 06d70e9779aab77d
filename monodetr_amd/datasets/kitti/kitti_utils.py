"""KITTI label / calibration records and the crop geometry -- mirror of ``lib/datasets/kitti/kitti_utils.py``
(names, attributes and numerics; the LiDAR / BEV helpers of that file are not on this path and are not mirrored).

The reference calls ``cv2.getAffineTransform`` for the crop matrices (kitti_utils.py:379-384); here the 6x6 system
it solves is solved directly in float64 (no OpenCV dependency)."""
import numpy as np

_LEVELS = ((40, 0.15, 0, 'Easy', 1), (25, 0.3, 1, 'Moderate', 2), (25, 0.5, 2, 'Hard', 3))


def _wrap_pi(a):
    if a > np.pi:
        a -= 2 * np.pi
    if a < -np.pi:
        a += 2 * np.pi
    return a


class Object3d(object):
    """One line of a ``label_2`` file (kitti_utils.py:13-52).  The reference's attribute spelling ``trucation`` is
    part of its interface and kept."""

    def __init__(self, line):
        f = line.strip().split(' ')
        self.src = line
        self.cls_type = f[0]
        self.trucation, self.occlusion, self.alpha = (float(v) for v in f[1:4])
        self.box2d = np.array([float(v) for v in f[4:8]], dtype=np.float32)
        self.h, self.w, self.l = (float(v) for v in f[8:11])
        self.pos = np.array([float(v) for v in f[11:14]], dtype=np.float32)
        self.dis_to_cam = np.linalg.norm(self.pos)
        self.ry = float(f[14])
        self.score = float(f[15]) if len(f) == 16 else -1.0
        self.level_str, self.level = self._difficulty()

    def _difficulty(self):
        if self.trucation == -1:
            return 'DontCare', 0
        height = float(self.box2d[3]) - float(self.box2d[1]) + 1
        for min_h, max_trunc, max_occ, name, code in _LEVELS:
            if height >= min_h and self.trucation <= max_trunc and self.occlusion <= max_occ:
                return name, code
        return 'UnKnown', 4

    def get_obj_level(self):
        return self.level

    def mirror(self, width):
        """Horizontal flip of the annotation (kitti_dataset.py:181-190)."""
        x1, x2 = self.box2d[0], self.box2d[2]
        self.box2d[0], self.box2d[2] = width - x2, width - x1
        self.alpha = _wrap_pi(np.pi - self.alpha)
        self.ry = _wrap_pi(np.pi - self.ry)

    def to_kitti_format(self):
        b, p = self.box2d, self.pos
        return '%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f' % (
            self.cls_type, self.trucation, int(self.occlusion), self.alpha, b[0], b[1], b[2], b[3],
            self.h, self.w, self.l, p[0], p[1], p[2], self.ry)


def get_objects_from_label(label_file):
    with open(label_file, 'r') as f:
        return [Object3d(line) for line in f.readlines()]


def get_calib_from_file(calib_file):
    with open(calib_file) as f:
        rows = [ln.strip().split(' ')[1:] for ln in f.readlines()[:6]]
    mat = lambda r, shape: np.array(rows[r], dtype=np.float32).reshape(shape)
    return {'P2': mat(2, (3, 4)), 'P3': mat(3, (3, 4)), 'R0': mat(4, (3, 3)), 'Tr_velo2cam': mat(5, (3, 4))}


class Calibration(object):
    """Camera-2 projection of one frame (kitti_utils.py:137-330, the image <-> rectified-camera part)."""

    def __init__(self, calib_file):
        calib = get_calib_from_file(calib_file) if isinstance(calib_file, str) else calib_file
        self.R0, self.V2C = calib['R0'], calib['Tr_velo2cam']
        self._set_p2(calib['P2'])

    def _set_p2(self, p2):
        self.P2 = p2
        self.cu, self.cv, self.fu, self.fv = p2[0, 2], p2[1, 2], p2[0, 0], p2[1, 1]
        self.tx, self.ty = p2[0, 3] / (-self.fu), p2[1, 3] / (-self.fv)

    def rect_to_img(self, pts_rect):
        """[N,3] rectified-camera points -> ([N,2] pixels, [N] depth)."""
        hom = np.hstack((pts_rect, np.ones((pts_rect.shape[0], 1), dtype=np.float32)))
        proj = np.dot(hom, self.P2.T)
        return (proj[:, 0:2].T / hom[:, 2]).T, proj[:, 2] - self.P2.T[3, 2]

    def img_to_rect(self, u, v, depth_rect):
        x = ((u - self.cu) * depth_rect) / self.fu + self.tx
        y = ((v - self.cv) * depth_rect) / self.fv + self.ty
        return np.concatenate((x.reshape(-1, 1), y.reshape(-1, 1), depth_rect.reshape(-1, 1)), axis=1)

    def alpha2ry(self, alpha, u):
        return _wrap_pi(alpha + np.arctan2(u - self.cu, self.fu))

    def ry2alpha(self, ry, u):
        return _wrap_pi(ry - np.arctan2(u - self.cu, self.fu))

    def flip(self, img_size):
        """Re-fit P2 for the mirrored image (kitti_utils.py:286-326): 8 pixel/depth grid points are lifted to 3-D,
        mirrored in x and in u, and the 7 unknowns of the projection are the null vector of the stacked
        constraints (smallest right singular vector)."""
        nw, nh = 4, 2
        us = np.tile(np.linspace(0, img_size[0], nw)[None, :], [nh, 1])
        vs = np.tile(np.linspace(0, img_size[1], nh)[:, None], [1, nw])
        ds = np.linspace(2, 78, nw * nh).reshape(nh, nw)
        p2d = np.stack([us, vs, ds], -1).reshape(-1, 3)
        p3d = self.img_to_rect(p2d[:, 0:1], p2d[:, 1:2], p2d[:, 2:3])
        p3d[:, 0] *= -1
        p2d[:, 0] = img_size[0] - p2d[:, 0]
        rows = np.zeros([nw * nh, 2, 7])
        rows[:, 0, 0], rows[:, 1, 0] = p3d[:, 0], p3d[:, 1]
        rows[:, 0, 1] = rows[:, 1, 2] = p3d[:, 2]
        rows[:, 0, 3] = rows[:, 1, 4] = 1
        rows[:, :, -2] = -p2d[:, :2]
        rows[:, :, -1] = -p2d[:, :2] * p3d[:, 2:3]
        sol = np.linalg.svd(rows.reshape(-1, 7))[-1][-1]
        sol /= sol[-1]
        m = np.zeros([4, 3]).astype(np.float32)
        m[0, 0] = m[1, 1] = sol[0]
        m[2, 0:2] = sol[1:3]
        m[3, :] = sol[3:6]
        m[-1, -1] = self.P2[-1, -1]
        self._set_p2(m.T)


def _solve_affine(src, dst):
    """The 2x3 float64 matrix taking three float32 points src -> dst (what cv2.getAffineTransform returns)."""
    s = np.asarray(src, dtype=np.float32).astype(np.float64)
    d = np.asarray(dst, dtype=np.float32).astype(np.float64)
    a, b = np.zeros((6, 6)), np.zeros(6)
    for i in range(3):
        a[i, 0:2], a[i, 2] = s[i], 1.0
        a[i + 3, 3:5], a[i + 3, 5] = s[i], 1.0
        b[i], b[i + 3] = d[i, 0], d[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def _triangle(origin, up):
    """Three float32 points: origin, origin + up, and the third corner of the right-angled isosceles triangle."""
    pts = np.zeros((3, 2), dtype=np.float32)
    pts[0, :] = origin
    pts[1, :] = origin + up
    leg = pts[0, :] - pts[1, :]
    pts[2:, :] = pts[1, :] + np.array([-leg[1], leg[0]], dtype=np.float32)
    return pts


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """Crop window (centre, size) -> output canvas; same arguments and return values as kitti_utils.py:347-384."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    half = scale[0] * -0.5
    src_up = [0 * cs - half * sn, 0 * sn + half * cs]
    dst_w, dst_h = output_size[0], output_size[1]
    src = _triangle(center + scale * shift, src_up)
    dst = _triangle(np.array([dst_w * 0.5, dst_h * 0.5], np.float32), np.array([0, dst_w * -0.5], np.float32))
    trans = _solve_affine(src, dst)
    return (trans, _solve_affine(dst, src)) if inv else trans


def affine_transform(pt, t):
    return np.dot(t, np.array([pt[0], pt[1], 1.], dtype=np.float32).T)[:2]
