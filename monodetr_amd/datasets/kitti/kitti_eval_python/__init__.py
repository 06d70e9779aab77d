"""KITTI object-detection evaluation (AP|R11, AP|R40 for image / bird's-eye-view / 3-D boxes, orientation similarity)
-- mirror of ``lib/datasets/kitti/kitti_eval_python``: rotated overlaps on the device, the serial statistics native."""
