"""Seeding and logging for the training script -- mirror of ``lib/helpers/utils_helper.py``.

``set_random_seed(s)`` seeds Python with s, numpy with s^2, torch's CPU generator with s^3 and the device generators
with s^4 -- the reference's scheme, so a run here draws the same augmentations and initial weights as a reference run
with the same ``random_seed``.  (The reference also forces cuDNN into deterministic mode; MIOpen has no such switch.)
"""
import logging
import random

import numpy as np
import torch

_FORMAT = '%(asctime)s  %(levelname)5s  %(message)s'


def set_random_seed(seed):
    for seeder, power in ((random.seed, 1), (np.random.seed, 2), (torch.manual_seed, 3)):
        seeder(seed ** power)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed ** 4)


def create_logger(log_file, rank=0):
    """File + console logger; ranks other than 0 only report errors."""
    level = logging.INFO if rank == 0 else logging.ERROR
    logging.basicConfig(level=level, format=_FORMAT, filename=log_file)
    logger = logging.getLogger(__name__)
    has_console = any(type(h) is logging.StreamHandler for h in logger.handlers)
    if not has_console:
        console = logging.StreamHandler()
        console.setFormatter(logging.Formatter(_FORMAT))
        console.setLevel(level)
        logger.addHandler(console)
    return logger
