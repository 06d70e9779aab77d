"""Heading-angle binning -- mirror of ``lib/datasets/utils.py:6-26`` (the heat-map helpers of that file belong to
the centre-net style heads MonoDETR does not use and are not mirrored)."""
import numpy as np

num_heading_bin = 12
_BIN = 2 * np.pi / float(num_heading_bin)


def angle2class(angle):
    """angle (rad) -> (bin in 0..11, residual from the bin centre); bin k is centred at k * 30 degrees."""
    angle = angle % (2 * np.pi)
    assert 0 <= angle <= 2 * np.pi
    shifted = (angle + _BIN / 2) % (2 * np.pi)
    k = int(shifted / _BIN)
    return k, shifted - (k * _BIN + _BIN / 2)


def class2angle(cls, residual, to_label_format=False):
    angle = cls * _BIN + residual
    if to_label_format and angle > np.pi:
        angle = angle - 2 * np.pi
    return angle
