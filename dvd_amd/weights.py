"""Repack a reference-named state_dict (the keys of `model1852000.pt`) into the
engine's tensor set (names / sizes enumerated by dvd_engine_tensor_info).

Host-side, once per model load:
  * patch-embed conv weights [384,C,2,2] -> GEMM operands with K order (p*2+q)*C + c (channels
    contiguous, so the patch-row builders write coalesced runs); r_embedder K padded 1032 -> 1088;
  * fused projections are split the way the kernels consume them (q / k / v^T, see attention.hip);
  * eval-mode BatchNorm of the decoder FFN (idf/cross_attn.py:24-50) is folded into the conv
    weights and a bias: y = conv(x) * g/sqrt(var+eps) + (beta - mean*g/sqrt(var+eps));
  * per-step GEMM weights are stored f16 (fp32 accumulate on MFMA); once-per-document weights
    (pyramid, c/m/l embedders, their K/V projections) stay f32 and run on the exact f32 MFMA path;
  * only blocks[-1] is packed: blocks 0..10 are dead compute (idf/cross_model.py:615-616).
"""
from __future__ import annotations

import numpy as np
import torch

HID, DEC, FFN, RK = 384, 1536, 2048, 1088
BN_EPS = 1e-5


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))


def _patch_w(w):
    """[O, C, 2, 2] conv weight -> [O, (p*2+q)*C + c]."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def _fold_bn(sd, prefix):
    g, b = _t(sd[prefix + "bn.weight"]).double(), _t(sd[prefix + "bn.bias"]).double()
    m, v = _t(sd[prefix + "bn.running_mean"]).double(), _t(sd[prefix + "bn.running_var"]).double()
    s = g / torch.sqrt(v + BN_EPS)
    return s, b - m * s


def live_block_index(sd) -> int:
    idx = [int(k.split(".")[1]) for k in sd if k.startswith("blocks.") and k.endswith("attn.qkv.weight")]
    if not idx:
        raise KeyError("state_dict has no DiT blocks")
    return max(idx)


def pack(sd, grid: int):
    """-> {engine tensor name: torch CPU tensor (f32 or f16, contiguous)}"""
    side = grid // 2
    T = side * side
    W = lambda k: _t(sd[k]).float()  # noqa: E731

    class _Out(dict):
        """records the fp32 source of every f16 tensor so it can be split into (hi, lo) at the end"""
        def __setitem__(self, k, v):
            if k.endswith("16"):
                full32[k] = v.float().contiguous()
                v = v.half()
            super().__setitem__(k, v)
    full32 = {}
    out = _Out()
    out["obs_w"] = W("obs_embedder.proj.weight").reshape(HID, 8)
    out["obs_b"] = W("obs_embedder.proj.bias")
    pos = W("noised_obs_pos_embed").reshape(-1, HID)
    if pos.shape[0] != T:
        raise ValueError(f"pos-embed has {pos.shape[0]} tokens, grid {grid} needs {T}")
    out["pos"] = pos
    rw = torch.zeros(HID, RK)
    rw[:, :1032] = _patch_w(W("r_embedder.proj.weight"))
    out["r_w16"] = rw
    out["r_b"] = W("r_embedder.proj.bias")
    for n, e in (("c", "c_embedder"), ("m", "m_embedder"), ("l", "l_embedder")):
        out[n + "_w"] = _patch_w(W(e + ".proj.weight"))
        out[n + "_b"] = W(e + ".proj.bias")
    out["t_w0"], out["t_b0"] = W("t_embedder.mlp.0.weight"), W("t_embedder.mlp.0.bias")
    out["t_w2"], out["t_b2"] = W("t_embedder.mlp.2.weight"), W("t_embedder.mlp.2.bias")
    b = f"blocks.{live_block_index(sd)}."
    out["ada_w"], out["ada_b"] = W(b + "adaLN_modulation.1.weight"), W(b + "adaLN_modulation.1.bias")
    ipw, ipb = W(b + "cross_attn.in_proj_weight"), W(b + "cross_attn.in_proj_bias")
    out["ca_wq16"], out["ca_bq"] = ipw[:HID], ipb[:HID]
    out["ca_wk16"], out["ca_bk"] = ipw[HID:2 * HID], ipb[HID:2 * HID]
    out["ca_wv16"], out["ca_bv"] = ipw[2 * HID:], ipb[2 * HID:]
    out["ca_wk32"], out["ca_wv32"] = ipw[HID:2 * HID], ipw[2 * HID:]
    out["ca_wo16"], out["ca_bo"] = W(b + "cross_attn.out_proj.weight"), W(b + "cross_attn.out_proj.bias")
    qkv_w, qkv_b = W(b + "attn.qkv.weight"), W(b + "attn.qkv.bias")
    out["sa_wqk16"], out["sa_bqk"] = qkv_w[:2 * HID], qkv_b[:2 * HID]
    out["sa_wv16"], out["sa_bv"] = qkv_w[2 * HID:], qkv_b[2 * HID:]
    out["sa_wp16"], out["sa_bp"] = W(b + "attn.proj.weight"), W(b + "attn.proj.bias")
    out["fc1_w16"], out["fc1_b"] = W(b + "mlp.fc1.weight"), W(b + "mlp.fc1.bias")
    out["fc2_w16"], out["fc2_b"] = W(b + "mlp.fc2.weight"), W(b + "mlp.fc2.bias")
    d = "decoder.position_dec."
    for hw in ("h", "w"):
        for j in (0, 2):
            out[f"pe_{hw}{j}_w"] = W(d + f"{hw}_scale.{j}.weight").reshape(DEC, DEC)
            out[f"pe_{hw}{j}_b"] = W(d + f"{hw}_scale.{j}.bias")
        tab = W(d + f"{hw}_position_encoder").reshape(DEC, -1)
        if tab.shape[1] < side:
            raise ValueError(f"decoder position table has {tab.shape[1]} positions, grid {grid} needs {side}")
        out[f"pe_{hw}tab"] = tab[:, :side].t().contiguous()
    for j in range(6):
        p, o = f"decoder.layer_stack.{j}.", f"d{j}_"
        out[o + "n1w"], out[o + "n1b"] = W(p + "norm1.weight"), W(p + "norm1.bias")
        out[o + "wqk16"] = torch.cat([W(p + "attn.linear_q.weight"), W(p + "attn.linear_k.weight")], 0)
        out[o + "wv16"] = W(p + "attn.linear_v.weight")
        out[o + "wfc16"] = W(p + "attn.fc.weight")
        out[o + "n2w"], out[o + "n2b"] = W(p + "norm2.weight"), W(p + "norm2.bias")
        f = p + "feed_forward."
        s1, b1 = _fold_bn(sd, f + "conv1.")
        out[o + "c1w16"] = (W(f + "conv1.conv.weight").reshape(FFN, DEC).double() * s1[:, None]).float()
        out[o + "c1b"] = b1.float()
        sd_, bd = _fold_bn(sd, f + "depthwise_conv.")
        dw = W(f + "depthwise_conv.conv.weight").reshape(FFN, 9).double() * sd_[:, None]
        out[o + "dww"] = dw.t().contiguous().float()          # [9, 2048] tap-major (ky*3+kx)
        out[o + "dwb"] = bd.float()
        s2, b2 = _fold_bn(sd, f + "conv2.")
        out[o + "c2w16"] = (W(f + "conv2.conv.weight").reshape(DEC, FFN).double() * s2[:, None]).float()
        out[o + "c2b"] = b2.float()
    out["dec_nw"], out["dec_nb"] = W("decoder.layer_norm.weight"), W("decoder.layer_norm.bias")
    out["fin_ada_w"] = W("final_layer2.adaLN_modulation.1.weight")
    out["fin_ada_b"] = W("final_layer2.adaLN_modulation.1.bias")
    out["fin_w"], out["fin_b"] = W("final_layer2.linear.weight"), W("final_layer2.linear.bias")
    names = ["level_0.0", "level_1.0", "level_2.0", "level_2.2", "level_3.0", "level_3.2", "level_3.4"]
    for i, n in enumerate(names):
        w = W(f"pyramid.{n}.weight")                               # [Cout, Cin, 3, 3]
        flat = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)       # (ky*3+kx)*Cin + c
        kp = ((flat.shape[1] + 63) // 64) * 64
        wp = torch.zeros(w.shape[0], kp)
        wp[:, :flat.shape[1]] = flat
        out[f"pyr{i}_w"] = wp
        out[f"pyr{i}_b"] = W(f"pyramid.{n}.bias")
    # split every f16 weight into hi + lo (both f16, lo UNSCALED - mostly f16 subnormals, absolute precision 2^-25,
    # which the gfx950 f16 MFMA honours): removes the systematic f16 weight-rounding error (which accumulates
    # linearly over the diffusion steps) at 2x the GEMM MFMAs, in one pass over K - see dvd_gemm_desc.B_lo
    for k in [k for k in out if k.endswith("16")]:
        hi = out[k]
        out[k + "_lo"] = (full32[k] - hi.float()).half()
    return {k: v.contiguous() for k, v in out.items()}
