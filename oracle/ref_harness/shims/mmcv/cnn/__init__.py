"""Shim of mmcv.cnn.ConvModule (cross_attn.py:9): conv -> norm -> activation, with
sub-modules named .conv / .bn / .activate as in mmcv 2.x (bias='auto' => no conv bias
when a norm layer follows)."""
from torch import nn


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, bias="auto", conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), inplace=True, **unused):
        super().__init__()
        has_norm = norm_cfg is not None
        if bias == "auto":
            bias = not has_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups, bias=bias)
        if has_norm:
            assert norm_cfg.get("type") == "BN"
            self.bn = nn.BatchNorm2d(out_channels)
        else:
            self.bn = None
        if act_cfg is not None:
            assert act_cfg.get("type") == "ReLU"
            self.activate = nn.ReLU(inplace=inplace)
        else:
            self.activate = None

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.activate is not None:
            x = self.activate(x)
        return x
