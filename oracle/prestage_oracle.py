"""CPU oracle for the PRE-STAGE conditioning nets of the DvD sampling path.

TEST INFRASTRUCTURE - NOT PRODUCT CODE (same rule as oracle/dvd_oracle.py: only tests/, smoke() and bench.py's
cpu_baseline leg may import it).

A from-scratch functional restatement (PyTorch-CPU fp32 over reference-named state dicts) of
  U2NETP                 train_settings/models/geotr/geotr_core.py:24-46 (REBNCONV, _upsample_like),
                         :48-330 (RSU7/6/5/4/4F), :745-845 (U2NETP.forward)
  Seg / GeoTr_Seg_Inf    geotr_core.py:984-995, :997-1019   (the GeoTr branch of GeoTr_Seg_Inf is dead on the live
                         configuration: its output `ref_bm` is only read when env.use_init_flow is set, and val_TDiff.py
                         never loads weights into it, :57-58)
  UNet                   train_settings/models/geotr/unet_model.py:4-37, unet_parts.py:8-77
  the glue               train_settings/dvd/evaluation.py:162-216
Parity with the real reference is PINNED by tests/golden/prestage_g16.npz (oracle/ref_harness/gen_golden.py G8,
made by running the reference's own modules on synthetic weights); tests/test_oracle_golden.py checks it.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd, k):
    v = sd[k]
    return v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))


def rebnconv(sd, p, x, dirate):
    """REBNCONV (geotr_core.py:24-36): conv3x3(padding = dilation = dirate) -> BatchNorm(eval) -> ReLU."""
    y = F.conv2d(x, _t(sd, p + "conv_s1.weight"), _t(sd, p + "conv_s1.bias"), padding=dirate, dilation=dirate)
    y = F.batch_norm(y, _t(sd, p + "bn_s1.running_mean"), _t(sd, p + "bn_s1.running_var"), _t(sd, p + "bn_s1.weight"),
                     _t(sd, p + "bn_s1.bias"), False, 0.0, 1e-5)
    return F.relu(y)


def up_like(src, tar):
    """_upsample_like (geotr_core.py:42-45): bilinear, align_corners=False, to tar's spatial size."""
    return F.interpolate(src, size=tar.shape[2:], mode="bilinear", align_corners=False)


def rsu(sd, p, x, depth):
    """RSU-7/6/5/4 (geotr_core.py:48-296): encoder with ceil-mode 2x2 max-pools after stages 1..depth-2, a dilated
    (dirate 2) bottom stage, decoder over concatenations with bilinear up-sampling, residual on rebnconvin."""
    hxin = rebnconv(sd, p + "rebnconvin.", x, 1)
    enc = []
    hx = hxin
    for k in range(1, depth):
        hk = rebnconv(sd, p + f"rebnconv{k}.", hx, 1)
        enc.append(hk)
        hx = F.max_pool2d(hk, 2, stride=2, ceil_mode=True) if k <= depth - 2 else hk
    d = rebnconv(sd, p + f"rebnconv{depth}.", hx, 2)
    for k in range(depth - 1, 0, -1):
        d = rebnconv(sd, p + f"rebnconv{k}d.", torch.cat((d, enc[k - 1]), 1), 1)
        if k > 1:
            d = up_like(d, enc[k - 2])
    return d + hxin


def rsu4f(sd, p, x):
    """RSU-4F (geotr_core.py:300-332): no pooling, dilations 1,2,4,8 down and 4,2,1 up."""
    hxin = rebnconv(sd, p + "rebnconvin.", x, 1)
    h1 = rebnconv(sd, p + "rebnconv1.", hxin, 1)
    h2 = rebnconv(sd, p + "rebnconv2.", h1, 2)
    h3 = rebnconv(sd, p + "rebnconv3.", h2, 4)
    h4 = rebnconv(sd, p + "rebnconv4.", h3, 8)
    h3d = rebnconv(sd, p + "rebnconv3d.", torch.cat((h4, h3), 1), 4)
    h2d = rebnconv(sd, p + "rebnconv2d.", torch.cat((h3d, h2), 1), 2)
    h1d = rebnconv(sd, p + "rebnconv1d.", torch.cat((h2d, h1), 1), 1)
    return h1d + hxin


def u2netp(sd, x, prefix=""):
    """U2NETP.forward (geotr_core.py:779-845) -> (sigmoid(d0), hx6, hx5d, hx4d, hx3d, hx2d, hx1d)."""
    P = prefix
    pool = lambda t: F.max_pool2d(t, 2, stride=2, ceil_mode=True)  # noqa: E731
    hx1 = rsu(sd, P + "stage1.", x, 7)
    hx2 = rsu(sd, P + "stage2.", pool(hx1), 6)
    hx3 = rsu(sd, P + "stage3.", pool(hx2), 5)
    hx4 = rsu(sd, P + "stage4.", pool(hx3), 4)
    hx5 = rsu4f(sd, P + "stage5.", pool(hx4))
    hx6 = rsu4f(sd, P + "stage6.", pool(hx5))
    hx5d = rsu4f(sd, P + "stage5d.", torch.cat((up_like(hx6, hx5), hx5), 1))
    hx4d = rsu(sd, P + "stage4d.", torch.cat((up_like(hx5d, hx4), hx4), 1), 4)
    hx3d = rsu(sd, P + "stage3d.", torch.cat((up_like(hx4d, hx3), hx3), 1), 5)
    hx2d = rsu(sd, P + "stage2d.", torch.cat((up_like(hx3d, hx2), hx2), 1), 6)
    hx1d = rsu(sd, P + "stage1d.", torch.cat((up_like(hx2d, hx1), hx1), 1), 7)
    side = lambda k, t: F.conv2d(t, _t(sd, P + f"side{k}.weight"), _t(sd, P + f"side{k}.bias"), padding=1)  # noqa: E731
    d1 = side(1, hx1d)
    ds = [d1] + [up_like(side(k, t), d1) for k, t in ((2, hx2d), (3, hx3d), (4, hx4d), (5, hx5d), (6, hx6))]
    d0 = F.conv2d(torch.cat(ds, 1), _t(sd, P + "outconv.weight"), _t(sd, P + "outconv.bias"))
    return torch.sigmoid(d0), hx6, hx5d, hx4d, hx3d, hx2d, hx1d


def double_conv(sd, p, x):
    """DoubleConv (unet_parts.py:8-26): (conv3x3 pad 1 -> BN(eval) -> ReLU) x 2."""
    for idx in (0, 3):
        x = F.conv2d(x, _t(sd, p + f"double_conv.{idx}.weight"), _t(sd, p + f"double_conv.{idx}.bias"), padding=1)
        q = p + f"double_conv.{idx + 1}."
        x = F.relu(F.batch_norm(x, _t(sd, q + "running_mean"), _t(sd, q + "running_var"), _t(sd, q + "weight"),
                                _t(sd, q + "bias"), False, 0.0, 1e-5))
    return x


def unet(sd, x):
    """UNet.forward (unet_model.py:25-37) -> (x [N,64,H,W], logits [N,1,H,W]); bilinear=True: Up = 2x bilinear
    (align_corners=True) + zero pad to the skip's size + cat([skip, up]) + DoubleConv (unet_parts.py:44-68)."""
    x1 = double_conv(sd, "inc.", x)
    skips = [x1]
    h = x1
    for k in range(1, 5):
        h = double_conv(sd, f"down{k}.maxpool_conv.1.", F.max_pool2d(h, 2))
        skips.append(h)
    for k in range(1, 5):
        skip = skips[4 - k]
        up = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        h = double_conv(sd, f"up{k}.conv.", torch.cat([skip, up], 1))
    logits = F.conv2d(h, _t(sd, "outc.conv.weight"), _t(sd, "outc.conv.bias"))
    return h, logits


def prestage(sd_dewarp_msk, sd_seg, sd_line, source512, grid, mask_override=None):
    """evaluation.py:162-216 on the live configuration (use_gt_mask False, use_line_mask True, use_init_flow False).
    source512 [N,3,512,512] in 0..1.  sd_dewarp_msk: U2NETP keys without prefix (what reload_segmodel leaves in
    GeoTr_Seg_Inf.msk); sd_seg: Seg keys ('msk.'-prefixed); sd_line: UNet keys.
    Returns dict(mask_cat [N,1,512,512], mask_y512 [N,384,G,G], line_msk [N,64,G,G], d0, mskx, source_288).
    mask_override: a [N,1,288,288] 0/1 mask used in place of (d0 > 0.5) - tests pass the device's own threshold
    decision so that a probability within rounding of 0.5 cannot flip the comparison of everything downstream."""
    src288 = F.interpolate(source512, size=288, mode="bilinear", align_corners=True)                    # :162
    msk_a = u2netp(sd_dewarp_msk, src288)[0]                                                           # geotr_core.py:1004
    mask_x = F.interpolate(msk_a, size=512, mode="bilinear", align_corners=True)                        # :1010
    d0, hx6, hx5d, hx4d, hx3d, hx2d, hx1d = u2netp(sd_seg, src288, "msk.")                              # :988
    d1 = (d0 > 0.5).float() if mask_override is None else mask_override
    mskx = d1 * src288                                                                                 # :990
    feats = [F.interpolate(t, size=grid, mode="bilinear", align_corners=False) for t in (hx6, hx5d, hx4d, hx3d, hx2d, hx1d)]
    seg_map_all = torch.cat(feats, dim=1)                                                              # :203-210
    line_map, _ = unet(sd_line, mskx)                                                                  # :214
    line_msk = F.interpolate(line_map, size=grid, mode="bilinear", align_corners=False)                 # :215
    return {"mask_cat": mask_x, "mask_y512": seg_map_all, "line_msk": line_msk, "d0": d0, "d0_dewarp": msk_a,
            "mskx": mskx, "source_288": src288, "hx": [hx6, hx5d, hx4d, hx3d, hx2d, hx1d], "line_map": line_map}
