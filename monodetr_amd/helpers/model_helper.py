"""Model construction entry point used by the training script -- mirror of ``lib/helpers/model_helper.py``.

``build_model(cfg)`` takes the ``model:`` section of the yaml file and returns ``(model, criterion)``: the MonoDETR
network and its ``SetCriterion`` (with the Hungarian matcher inside), both still on the CPU; the caller moves them to
its device.  Only the ``monodetr`` model family exists in the reference, so there is nothing to dispatch on.
"""
from .. import monodetr as _family


def build_model(cfg):
    model, criterion = _family.build_monodetr(cfg)
    return model, criterion
