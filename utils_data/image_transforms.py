"""`ArrayToTensor` - the one transform the sampling plug-in builds (val_TDiff.py:95; utils_data/image_transforms.py:44-62):
numpy HWC -> torch CHW, as float when asked, NOT normalised."""
from __future__ import annotations

import numpy as np
import torch


class ArrayToTensor:
    def __init__(self, get_float=True):
        self.get_float = get_float

    def __call__(self, array):
        array = np.ascontiguousarray(np.transpose(np.asarray(array), (2, 0, 1)))
        t = torch.from_numpy(array)
        return t.float() if self.get_float else t
