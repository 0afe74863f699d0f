"""The warp call of the sampling path in the reference's shape (datasets/utils/warping.py:14-23,50-73):

    reg_model_bilin = register_model2((512, 512), 'bilinear')
    warped = reg_model_bilin([img, grid])          # img [N,C,Hin,Win] f32, grid [N,2,H,W] in [-1,1], channel 0 = x

i.e. `F.grid_sample(img, grid.permute(0,2,3,1), mode='bilinear', padding_mode='zeros', align_corners=True)`.
The modules hold no parameters and no compute of their own: `forward` hands the two device tensors to
`dvd_grid_sample_bilinear_zeros_ac` (include/dvd_hip.h; LDS-tiled gather kernel, warp.hip) on the current HIP
stream.  Like every product path there is no CPU route: host tensors are rejected."""
from __future__ import annotations

import torch
from torch import nn

from dvd_amd import ops


class SpatialTransformer2(nn.Module):
    """`forward(src, flow)`: flow is the NORMALISED sampling grid, NCHW with 2 channels (x first); no base grid is
    added and nothing is rescaled (warping.py:50-73)."""

    def __init__(self, size, mode="bilinear"):
        super().__init__()
        if mode != "bilinear":
            raise NotImplementedError(f"mode {mode!r}: the sampling path warps bilinearly (gaussian_diffusion.py:20,218)")
        self.size, self.mode = tuple(size), mode      # `size` is unused by the reference's forward as well

    def forward(self, src, flow):
        if src.dim() != 4 or flow.dim() != 4 or flow.shape[1] != 2:
            raise ValueError(f"expected src [N,C,Hin,Win] and flow [N,2,H,W], got {tuple(src.shape)} / {tuple(flow.shape)}")
        if flow.shape[0] != src.shape[0]:
            raise ValueError(f"batch sizes differ: src {src.shape[0]}, flow {flow.shape[0]}")   # F.grid_sample's rule
        if flow.device != src.device:
            raise ValueError(f"src on {src.device}, flow on {flow.device}")
        return ops.grid_sample(src.to(torch.float32).contiguous(), flow.to(torch.float32).contiguous())


class register_model2(nn.Module):
    """`forward([img, flow])` - the list-taking wrapper every call site uses (warping.py:14-23)."""

    def __init__(self, img_size=(64, 1024, 1024), mode="bilinear"):
        super().__init__()
        self.spatial_trans = SpatialTransformer2(img_size, mode)

    def forward(self, x):
        img, flow = x[0], x[1]
        return self.spatial_trans(img, flow)
