"""monodetr_amd -- MI355X-native implementation of MonoDETR's training hot path.

    monodetr_amd.monodetr          mirror of the reference's ``lib/models/monodetr`` package
    monodetr_amd.msda_ext          mirror of its native extension ``MultiScaleDeformableAttention``
    monodetr_amd.utils / .losses   the pieces of ``utils`` / ``lib.losses`` the model path imports
    monodetr_amd.csrc              HIP kernels + C ABI (libmonodetr_amd.so, include/monodetr_amd.h)

``install()`` registers these under the reference's module names so its unchanged
``tools/train_val.py`` / ``lib/helpers`` import them.
"""
import importlib
import sys

from . import _runtime_env  # noqa: F401  (runtime flags: effective when this package is imported before torch)

__all__ = ["install", "build_monodetr"]


def build_monodetr(cfg):
    from .monodetr import build_monodetr as _b
    return _b(cfg)


def install(force=False):
    """Alias this package's modules under the names the reference code imports:
    ``MultiScaleDeformableAttention`` (ops/functions/ms_deform_attn_func.py:18),
    ``lib.models.monodetr`` (lib/helpers/model_helper.py:1), ``lib.losses.focal_loss``,
    ``utils.misc`` / ``utils.box_ops``, and the input pipeline ``lib.helpers.dataloader_helper`` /
    ``lib.datasets.kitti.kitti_dataset`` (lib/helpers/dataloader_helper.py:4, tools/train_val.py).
    Existing entries are kept unless force=True."""
    aliases = {
        "MultiScaleDeformableAttention": ".msda_ext",
        "lib.models.monodetr": ".monodetr",
        "lib.losses.focal_loss": ".losses.focal_loss",
        "utils.misc": ".utils.misc",
        "utils.box_ops": ".utils.box_ops",
        "lib.helpers.dataloader_helper": ".helpers.dataloader_helper",
        "lib.helpers.model_helper": ".helpers.model_helper",
        "lib.helpers.utils_helper": ".helpers.utils_helper",
        "lib.helpers.scheduler_helper": ".helpers.scheduler_helper",
        "lib.helpers.trainer_helper": ".helpers.trainer_helper",
        "lib.helpers.optimizer_helper": ".helpers.optimizer_helper",
        "lib.helpers.decode_helper": ".helpers.decode_helper",
        "lib.helpers.tester_helper": ".helpers.tester_helper",
        "lib.helpers.save_helper": ".helpers.save_helper",
        "lib.datasets.kitti.kitti_eval_python.eval": ".datasets.kitti.kitti_eval_python.eval",
        "lib.datasets.kitti.kitti_eval_python.kitti_common": ".datasets.kitti.kitti_eval_python.kitti_common",
        "lib.datasets.kitti.kitti_eval_python.rotate_iou": ".datasets.kitti.kitti_eval_python.rotate_iou",
        "lib.datasets.utils": ".datasets.utils",
        "lib.datasets.kitti.kitti_utils": ".datasets.kitti.kitti_utils",
        "lib.datasets.kitti.kitti_dataset": ".datasets.kitti.kitti_dataset",
    }
    for name, target in aliases.items():
        if force or name not in sys.modules:
            sys.modules[name] = importlib.import_module(target, __name__)
    return sorted(aliases)
