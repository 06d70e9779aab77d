"""Depth-map supervision of the depth predictor (mirror of ``lib/models/monodetr/depth_predictor/ddn_loss``): the
package exposes ``DDNLoss`` only; the fused device kernel lives in ``monodetr_amd/ddn_loss_ext.py``."""
from . import ddn_loss as _impl

DDNLoss = _impl.DDNLoss
__all__ = ["DDNLoss"]
