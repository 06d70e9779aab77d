"""Model package with the reference's entry point: ``build_monodetr(cfg) -> (model, criterion)``
(lib/models/monodetr/__init__.py), where ``cfg`` is the ``model:`` section of configs/monodetr.yaml."""
from . import monodetr as _impl

__all__ = ["build_monodetr"]


def build_monodetr(cfg):
    model, criterion = _impl.build(cfg)
    return model, criterion
