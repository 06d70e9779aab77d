"""``build_model(cfg)`` -- mirror of ``lib/helpers/model_helper.py``: ``(model, criterion)`` for the ``model:`` section."""
from ..monodetr import build_monodetr


def build_model(cfg):
    return build_monodetr(cfg)
