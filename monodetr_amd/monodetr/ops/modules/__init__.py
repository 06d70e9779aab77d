# same three names as lib/models/monodetr/ops/modules/__init__.py:9-11
from .ms_deform_attn import MSDeformAttn, MSDeformAttn_cross, MultiheadAttention  # noqa: F401
