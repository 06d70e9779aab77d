from .kitti_dataset import KITTI_Dataset

__all__ = ["KITTI_Dataset"]
