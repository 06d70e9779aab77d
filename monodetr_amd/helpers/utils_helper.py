"""Logger and seeding -- mirror of ``lib/helpers/utils_helper.py`` (seed, seed^2, seed^3, seed^4 for Python / numpy /
torch CPU / torch device generators, so that a run is comparable with a reference run of the same ``random_seed``)."""
import logging
import random

import numpy as np
import torch


def create_logger(log_file, rank=0):
    fmt = '%(asctime)s  %(levelname)5s  %(message)s'
    level = logging.INFO if rank == 0 else logging.ERROR
    logging.basicConfig(level=level, format=fmt, filename=log_file)
    logger = logging.getLogger(__name__)
    if not any(isinstance(h, logging.StreamHandler) and not isinstance(h, logging.FileHandler) for h in logger.handlers):
        console = logging.StreamHandler()
        console.setLevel(level)
        console.setFormatter(logging.Formatter(fmt))
        logger.addHandler(console)
    return logger


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed ** 2)
    torch.manual_seed(seed ** 3)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed ** 4)
