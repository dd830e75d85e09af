"""Small helpers with the reference's names (utilities/utils.py:10-36)."""
import logging
import random
from datetime import datetime

import numpy as np
import torch


def generate_id(prefix=None, postfix=None) -> str:
    """Timestamp id used for run folders, e.g. 2024-11-15_9-3-27.123456 (utilities/utils.py:10-18)."""
    now = datetime.now()
    uid = f'{now.year}-{now.month}-{now.day}_{now.hour}-{now.minute}-{now.second}.{now.microsecond}'
    return '_'.join(part for part in (prefix, uid, postfix) if part is not None)


def reproducible(seed: int):
    """Seed python, torch (host + every HIP device) and numpy (utilities/utils.py:21-26)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)  # also seeds the HIP generators


def log_info_results(metrics: dict):
    for name, value in metrics.items():
        logging.info('%-10s : %.5f', name, value)
