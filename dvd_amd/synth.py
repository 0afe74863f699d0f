"""Deterministic synthetic weights and inputs for the DvD sampling path.

There is no checkpoint in the reference tree (checkpoints/ only holds a download note,
README.md:46-58) and no network here, so every run - golden-vector generation with the
real reference, the CPU oracle, the HIP engine, bench.py - draws its tensors from the
same counter-based generator:  value(key, i) = f(seed, fnv1a(key), i).

The generator is pure integer arithmetic on uint64 (splitmix64 finaliser) followed by an
exact int->float32 conversion, so it produces bit-identical tensors on every platform
and never needs the 607 MB of weights to be stored anywhere.

`state_dict_spec()` lists the tensors of the live model
(`DiT_models2['DiT-S/2'](input_size=G, in_channels=2, tv=True)`,
reference train_settings/dvd/improved_diffusion/cross_model.py:361-460 and
cross_attn.py:399-458) with the checkpoint key names, so a real `model1852000.pt`
and the synthetic weights go through the same loader.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def uniform01(key: str, n: int, seed: int = 0, offset: int = 0) -> np.ndarray:
    """n float32 values in [0, 1) on a 2^-24 lattice, element i = f(seed, key, offset+i)."""
    base = np.uint64((fnv1a64(key) ^ ((seed * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF))
                     & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        bits = _splitmix64(_splitmix64(ctr ^ base) + base)
    return (bits >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def uniform(key: str, shape, lo: float, hi: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(key, n, seed)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32).reshape(shape)


def normalish(key: str, shape, seed: int = 0) -> np.ndarray:
    """Approximately N(0,1): sum of 4 uniforms (Irwin-Hall), exact float32 arithmetic order."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float32)
    for j in range(4):
        acc = acc + uniform01(key + f"#ih{j}", n, seed)
    # mean 2, variance 4/12
    return ((acc - np.float32(2.0)) * np.float32(math.sqrt(3.0))).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# Model tensor inventory (reference key names)
# --------------------------------------------------------------------------------------

HID = 384          # DiT-S hidden size                       cross_model.py:766-767
HEADS = 6
DEPTH = 12         # 12 blocks exist, only blocks[-1] is live  cross_model.py:615-616
DEC = 4 * HID      # decoder d_model = 1536                   cross_model.py:446
DEC_INNER = 2048   # locality-aware FFN hidden                cross_model.py:448
DEC_LAYERS = 6
TFREQ = 256        # time_frequency_embedding_size            cross_model.py:375

_PYR = [("level_0.0", 4, 64), ("level_1.0", 64, 64), ("level_2.0", 64, 128), ("level_2.2", 128, 128),
        ("level_3.0", 128, 256), ("level_3.2", 256, 256), ("level_3.4", 256, 256)]
_EMB = [("obs_embedder", 2), ("r_embedder", 258), ("c_embedder", 256), ("m_embedder", 384),
        ("l_embedder", 64)]


def state_dict_spec(grid: int = 64, blocks=range(DEPTH)):
    """Ordered {key: (shape, kind)} for the live model at coordinate-grid size `grid`.

    kind drives the synthetic distribution: 'w' weight with fan-in = prod(shape[1:]),
    'b' bias, 'ln_w'/'ln_b' LayerNorm affine, 'bn_w','bn_b','bn_m','bn_v','bn_n' BatchNorm,
    'pos' fixed sin-cos table (computed, not random), 'dec_h'/'dec_w' decoder sinusoid
    buffers (computed).
    """
    T = (grid // 2) ** 2
    npos = grid // 2
    s = OrderedDict()
    s["noised_obs_pos_embed"] = ((1, T, HID), "pos")
    for name, cin, cout in _PYR:
        s[f"pyramid.{name}.weight"] = ((cout, cin, 3, 3), "w")
        s[f"pyramid.{name}.bias"] = ((cout,), "b")
    for name, cin in _EMB:
        s[f"{name}.proj.weight"] = ((HID, cin, 2, 2), "w")
        s[f"{name}.proj.bias"] = ((HID,), "b")
    s["t_embedder.mlp.0.weight"] = ((HID, TFREQ), "w")
    s["t_embedder.mlp.0.bias"] = ((HID,), "b")
    s["t_embedder.mlp.2.weight"] = ((HID, HID), "w")
    s["t_embedder.mlp.2.bias"] = ((HID,), "b")
    for i in blocks:
        p = f"blocks.{i}."
        s[p + "attn.qkv.weight"] = ((3 * HID, HID), "w")
        s[p + "attn.qkv.bias"] = ((3 * HID,), "b")
        s[p + "attn.proj.weight"] = ((HID, HID), "w")
        s[p + "attn.proj.bias"] = ((HID,), "b")
        s[p + "mlp.fc1.weight"] = ((4 * HID, HID), "w")
        s[p + "mlp.fc1.bias"] = ((4 * HID,), "b")
        s[p + "mlp.fc2.weight"] = ((HID, 4 * HID), "w")
        s[p + "mlp.fc2.bias"] = ((HID,), "b")
        s[p + "adaLN_modulation.1.weight"] = ((6 * HID, HID), "w_mod")
        s[p + "adaLN_modulation.1.bias"] = ((6 * HID,), "b_mod")
        s[p + "cross_attn.in_proj_weight"] = ((3 * HID, HID), "w")
        s[p + "cross_attn.in_proj_bias"] = ((3 * HID,), "b")
        s[p + "cross_attn.out_proj.weight"] = ((HID, HID), "w")
        s[p + "cross_attn.out_proj.bias"] = ((HID,), "b")
    d = "decoder.position_dec."
    s[d + "h_position_encoder"] = ((1, DEC, npos, 1), "dec_h")
    s[d + "w_position_encoder"] = ((1, DEC, 1, npos), "dec_w")
    for hw in ("h_scale", "w_scale"):
        for j in (0, 2):
            s[d + f"{hw}.{j}.weight"] = ((DEC, DEC, 1, 1), "w")
            s[d + f"{hw}.{j}.bias"] = ((DEC,), "b")
    for j in range(DEC_LAYERS):
        p = f"decoder.layer_stack.{j}."
        s[p + "norm1.weight"] = ((DEC,), "ln_w")
        s[p + "norm1.bias"] = ((DEC,), "ln_b")
        for q in ("linear_q", "linear_k", "linear_v", "fc"):
            s[p + f"attn.{q}.weight"] = ((DEC, DEC), "w")
        s[p + "norm2.weight"] = ((DEC,), "ln_w")
        s[p + "norm2.bias"] = ((DEC,), "ln_b")
        for cname, shape in (("conv1", (DEC_INNER, DEC, 1, 1)), ("depthwise_conv", (DEC_INNER, 1, 3, 3)),
                             ("conv2", (DEC, DEC_INNER, 1, 1))):
            c = p + f"feed_forward.{cname}."
            s[c + "conv.weight"] = (shape, "w")
            s[c + "bn.weight"] = ((shape[0],), "bn_w")
            s[c + "bn.bias"] = ((shape[0],), "bn_b")
            s[c + "bn.running_mean"] = ((shape[0],), "bn_m")
            s[c + "bn.running_var"] = ((shape[0],), "bn_v")
            s[c + "bn.num_batches_tracked"] = ((), "bn_n")
    s["decoder.layer_norm.weight"] = ((DEC,), "ln_w")
    s["decoder.layer_norm.bias"] = ((DEC,), "ln_b")
    s["final_layer2.linear.weight"] = ((8, DEC), "w_out")
    s["final_layer2.linear.bias"] = ((8,), "b")
    s["final_layer2.adaLN_modulation.1.weight"] = ((2 * DEC, DEC), "w_mod")
    s["final_layer2.adaLN_modulation.1.bias"] = ((2 * DEC,), "b_mod")
    return s


def sincos_pos_embed_2d(dim: int, side: int) -> np.ndarray:
    """Fixed 2-D sin-cos table [side*side, dim]; first half of the channels encodes the
    column (w) index, second half the row (h) index; each half is [sin | cos].
    Restates cross_model.py:677-722 (meshgrid 'w first')."""
    def one_d(d, pos):
        omega = np.arange(d // 2, dtype=np.float64) / (d / 2.0)
        omega = 1.0 / 10000 ** omega
        out = pos.reshape(-1).astype(np.float64)[:, None] * omega[None, :]
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    ys, xs = np.meshgrid(np.arange(side, dtype=np.float32), np.arange(side, dtype=np.float32),
                         indexing="ij")
    emb = np.concatenate([one_d(dim // 2, xs), one_d(dim // 2, ys)], axis=1)
    return emb.astype(np.float32)


def decoder_sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """[n_position, d_hid] table of cross_attn.py:122-134: angle = pos / 10000^(2*(j//2)/d),
    sin on even j, cos on odd j.  The reference builds the denominator in float64 then
    stores it in a float32 tensor and multiplies by a float32 position."""
    j = np.arange(d_hid)
    denom = (1.0 / np.power(10000, 2 * (j // 2) / d_hid)).astype(np.float32)
    pos = np.arange(n_position, dtype=np.float32)[:, None]
    tab = pos * denom[None, :]
    out = tab.copy()
    out[:, 0::2] = np.sin(tab[:, 0::2])
    out[:, 1::2] = np.cos(tab[:, 1::2])
    return out.astype(np.float32)


def synth_tensor(key: str, shape, kind: str, seed: int = 0) -> np.ndarray:
    if kind == "w":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        a = math.sqrt(3.0 / fan_in)          # std = 1/sqrt(fan_in)
        return uniform(key, shape, -a, a, seed)
    if kind == "w_mod":                       # adaLN projections: zero in a fresh reference model
        fan_in = int(np.prod(shape[1:]))      # (cross_model.py:535-545); small but non-zero here
        a = 0.5 * math.sqrt(3.0 / fan_in)
        return uniform(key, shape, -a, a, seed)
    if kind == "w_out":                       # final linear (zero-init in the reference)
        fan_in = int(np.prod(shape[1:]))
        a = 0.25 * math.sqrt(3.0 / fan_in)
        return uniform(key, shape, -a, a, seed)
    if kind in ("b", "b_mod"):
        return uniform(key, shape, -0.05, 0.05, seed)
    if kind == "ln_w":
        return uniform(key, shape, 0.9, 1.1, seed)
    if kind == "ln_b":
        return uniform(key, shape, -0.05, 0.05, seed)
    if kind == "bn_w":
        return uniform(key, shape, 0.6, 1.2, seed)
    if kind == "bn_b":
        return uniform(key, shape, -0.2, 0.1, seed)
    if kind == "bn_m":
        return uniform(key, shape, -0.1, 0.1, seed)
    if kind == "bn_v":
        return uniform(key, shape, 0.5, 1.5, seed)
    if kind == "bn_n":
        return np.asarray(1, dtype=np.int64)
    raise ValueError(f"no synthetic rule for kind {kind!r} ({key})")


def tame_gain(steps: int) -> float:
    """`out_gain` of the TAME weight family for an S-step roll-out.  The denoiser's output is `o + init_flow` with
    init_flow = the previous step's x0 (idf/cross_model.py:645-646, idf/gaussian_diffusion.py:618-620), so under random
    weights x0 grows roughly linearly with the step count: with the plain family (out_gain 1) 60-90 % of the last
    step's pixels lie outside [-1, 1] and the final clamp hides their error.  A trained model's coordinates stay inside
    (-1, 1); 1.6 / S keeps the last-step x0 at a standard deviation of 0.3-0.5 with no saturated pixel, so the parity
    tests can assert on the un-clamped, un-averaged x0."""
    return 1.6 / steps


def synth_state_dict(grid: int = 64, seed: int = 0, blocks=range(DEPTH), keys=None, out_gain: float = 1.0):
    """name -> numpy array for every tensor of the spec (or only `keys`).  out_gain scales the final linear layer
    (weight and bias): 1.0 is the family the golden vectors were made with, `tame_gain(S)` the tame family."""
    spec = state_dict_spec(grid, blocks)
    out = OrderedDict()
    for k, (shape, kind) in spec.items():
        if keys is not None and k not in keys:
            continue
        if kind == "pos":
            out[k] = sincos_pos_embed_2d(HID, grid // 2)[None]
        elif kind == "dec_h":
            out[k] = decoder_sinusoid_table(grid // 2, DEC).T.reshape(1, DEC, grid // 2, 1).copy()
        elif kind == "dec_w":
            out[k] = decoder_sinusoid_table(grid // 2, DEC).T.reshape(1, DEC, 1, grid // 2).copy()
        else:
            out[k] = synth_tensor(k, shape, kind, seed)
            if out_gain != 1.0 and k.startswith("final_layer2.linear."):
                out[k] = (out[k] * np.float32(out_gain)).astype(np.float32)
    return out


# --------------------------------------------------------------------------------------
# Synthetic per-document inputs; value ranges mirror the real tensors
# --------------------------------------------------------------------------------------

def synth_document(doc_idx: int, grid: int, seed: int = 1234, full_res=None):
    """Conditioning tensors for one document (numpy float32):
       y512 [3,512,512] in [0,1)           (evaluation.py:189  source image / 255)
       mask_cat [1,512,512] in [0,1)       (sigmoid output of the doc-mask net)
       mask_y512 [384,G,G] >= 0            (resized post-ReLU U2-Net features)
       line_msk [64,G,G]  >= 0
       full_res [H,W,3] uint8 when full_res=(H,W) is given."""
    tag = f"doc{doc_idx}/"
    d = {
        "y512": uniform01(tag + "y512", 3 * 512 * 512, seed).reshape(3, 512, 512),
        "mask_cat": uniform01(tag + "mask_cat", 512 * 512, seed).reshape(1, 512, 512),
        "mask_y512": np.maximum(normalish(tag + "mask_y512", (384, grid, grid), seed), 0),
        "line_msk": np.maximum(normalish(tag + "line_msk", (64, grid, grid), seed), 0),
    }
    if full_res is not None:
        h, w = full_res
        u = uniform01(tag + "src", h * w * 3, seed)
        d["src_u8"] = np.minimum((u * np.float32(256.0)).astype(np.int32), 255).astype(np.uint8).reshape(h, w, 3)
    return d


def synth_noise(doc_idx: int, n_hyp: int, grid: int, seed: int = 1234, step=None) -> np.ndarray:
    """x_T (step=None) or the per-step DDPM noise: [n_hyp, 2, G, G] ~ N(0,1)-ish."""
    tag = f"doc{doc_idx}/noise" + ("" if step is None else f"/s{step}")
    return normalish(tag, (n_hyp, 2, grid, grid), seed)


# --------------------------------------------------------------------------------------
# Pre-stage conditioning nets: tensor inventories with the reference's
# state_dict key names, so real seg.pth / seg_model.pth / line_model2.pth and the synthetic
# weights go through the same loaders.
#   U2NETP   train_settings/models/geotr/geotr_core.py:24-36 (REBNCONV), :48-330 (RSU7..RSU4F), :745-845
#   UNet     train_settings/models/geotr/unet_model.py:4-37, unet_parts.py:8-77
# --------------------------------------------------------------------------------------

def _rebnconv(s, p, cin, cout):
    s[p + "conv_s1.weight"] = ((cout, cin, 3, 3), "w")
    s[p + "conv_s1.bias"] = ((cout,), "b")
    s[p + "bn_s1.weight"] = ((cout,), "bn_w")
    s[p + "bn_s1.bias"] = ((cout,), "bn_b")
    s[p + "bn_s1.running_mean"] = ((cout,), "bn_m")
    s[p + "bn_s1.running_var"] = ((cout,), "bn_v")
    s[p + "bn_s1.num_batches_tracked"] = ((), "bn_n")


def _rsu(s, p, depth, cin, mid, cout, flat=False):
    """RSU-`depth` (depth 7/6/5/4) or the dilated RSU-4F: registration order of the reference's __init__."""
    _rebnconv(s, p + "rebnconvin.", cin, cout)
    _rebnconv(s, p + "rebnconv1.", cout, mid)
    for k in range(2, depth + 1):
        _rebnconv(s, p + f"rebnconv{k}.", mid, mid)
    for k in range(depth - 1, 1, -1):
        _rebnconv(s, p + f"rebnconv{k}d.", 2 * mid, mid)
    _rebnconv(s, p + "rebnconv1d.", 2 * mid, cout)


U2NETP_STAGES = [("stage1", 7, 3, False), ("stage2", 6, 64, False), ("stage3", 5, 64, False), ("stage4", 4, 64, False),
                 ("stage5", 4, 64, True), ("stage6", 4, 64, True), ("stage5d", 4, 128, True), ("stage4d", 4, 128, False),
                 ("stage3d", 5, 128, False), ("stage2d", 6, 128, False), ("stage1d", 7, 128, False)]


def u2netp_spec(prefix: str = ""):
    """U2NETP(3, 1): mid 16, out 64 everywhere (geotr_core.py:745-777)."""
    s = OrderedDict()
    for name, depth, cin, flat in U2NETP_STAGES:
        _rsu(s, f"{prefix}{name}.", depth, cin, 16, 64, flat)
    for k in range(1, 7):
        s[f"{prefix}side{k}.weight"] = ((1, 64, 3, 3), "w")
        s[f"{prefix}side{k}.bias"] = ((1,), "b")
    s[prefix + "outconv.weight"] = ((1, 6, 1, 1), "w")
    s[prefix + "outconv.bias"] = ((1,), "b")
    return s


def _double_conv(s, p, cin, cout, mid=None):
    mid = mid or cout
    for idx, (a, b) in ((0, (cin, mid)), (3, (mid, cout))):
        s[p + f"double_conv.{idx}.weight"] = ((b, a, 3, 3), "w")
        s[p + f"double_conv.{idx}.bias"] = ((b,), "b")
        q = p + f"double_conv.{idx + 1}."
        s[q + "weight"] = ((b,), "bn_w")
        s[q + "bias"] = ((b,), "bn_b")
        s[q + "running_mean"] = ((b,), "bn_m")
        s[q + "running_var"] = ((b,), "bn_v")
        s[q + "num_batches_tracked"] = ((), "bn_n")


def unet_spec():
    """UNet(n_channels=3, n_classes=1, bilinear=True) (unet_model.py:4-23)."""
    s = OrderedDict()
    _double_conv(s, "inc.", 3, 64)
    for k, (a, b) in enumerate(((64, 128), (128, 256), (256, 512), (512, 512)), 1):
        _double_conv(s, f"down{k}.maxpool_conv.1.", a, b)
    for k, (a, b) in enumerate(((1024, 256), (512, 128), (256, 64), (128, 64)), 1):
        _double_conv(s, f"up{k}.conv.", a, b, a // 2)
    s["outc.conv.weight"] = ((1, 64, 1, 1), "w")
    s["outc.conv.bias"] = ((1,), "b")
    return s


def synth_convnet_state_dict(kind: str, seed: int = 0, prefix: str = ""):
    """Synthetic state dict of 'u2netp' (keys optionally prefixed, e.g. 'msk.') or 'unet'.  ReLU-BN stacks with
    std-1/sqrt(fan_in) uniform weights lose variance layer by layer, so conv weights carry a gain of 1.2 (activations
    stay O(1) through the ~40-conv-deep U2NETP; with Seg seed 22 the document mask covers ~80 % of the synthetic page)."""
    spec = u2netp_spec(prefix) if kind == "u2netp" else unet_spec()
    out = OrderedDict()
    for k, (shape, knd) in spec.items():
        t = synth_tensor(f"{kind}/{k}", shape, knd, seed)
        if knd == "w":
            t = (t * np.float32(1.2)).astype(np.float32)
        out[k] = t
    return out


def smooth_image(key: str, h: int, w: int, seed: int = 1234) -> np.ndarray:
    """[3,h,w] float32 in [0,1]: a page-like picture (a bright quadrilateral with dark 'text' stripes on a darker,
    textured background) from the counter-based generator - gives the pre-stage nets something with structure, so
    the document-mask probability is not hovering around its 0.5 threshold everywhere."""
    ys, xs = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    u = uniform01(key + "/par", 8, seed)
    cx, cy = 0.5 + 0.08 * (u[0] - 0.5), 0.5 + 0.08 * (u[1] - 0.5)
    shear = 0.25 * (u[2] - 0.5)
    xr, yr = xs - cx + shear * (ys - cy), ys - cy - 0.5 * shear * (xs - cx)
    page = ((np.abs(xr) < 0.30 + 0.05 * u[3]) & (np.abs(yr) < 0.38 + 0.05 * u[4])).astype(np.float32)
    lines = (np.sin(yr * (60.0 + 30.0 * u[5])) > 0.55).astype(np.float32) * (np.abs(xr) < 0.26)
    noise = uniform01(key + "/noise", 3 * h * w, seed).reshape(3, h, w)
    base = np.stack([0.25 + 0.1 * xs, 0.2 + 0.15 * ys, 0.3 - 0.1 * xs * ys]).astype(np.float32)
    img = base * (1 - page) + page * (0.92 - 0.75 * lines) + 0.06 * (noise - 0.5)
    return np.clip(img, 0.0, 1.0).astype(np.float32)
