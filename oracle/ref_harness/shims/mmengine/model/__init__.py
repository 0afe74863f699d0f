"""Shim of mmengine.model.BaseModule: an nn.Module accepting init_cfg."""
from torch import nn


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
