"""seed_everything (reference: promptttspp/utils/seed.py:22-28)."""
import os
import random

import numpy as np
import torch


def seed_everything(seed=1234):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    from .. import functional as PF

    PF.manual_seed(seed)  # the HIP kernels' counter-based dropout stream
