from .ddn_loss import DDNLoss  # noqa: F401

__all__ = ["DDNLoss"]
