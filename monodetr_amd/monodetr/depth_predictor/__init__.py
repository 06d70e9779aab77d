from .depth_predictor import DepthPredictor  # noqa: F401  (depth_predictor/__init__.py:1)
