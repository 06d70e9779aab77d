"""Mirror of lib/models/monodetr: ``build_monodetr(cfg) -> (model, criterion)`` (__init__.py:1-5)."""
from .monodetr import build


def build_monodetr(cfg):
    return build(cfg)
