"""GPU parity of the token-side HIP kernels, each against the torch-CPU op(s) it replaces (the same ops the
oracle / the reference call)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dvd_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dvd_amd import ops as _ops
    return _ops


def rnd(tag, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(synth.uniform(tag, shape, lo, hi, 21))


@pytest.mark.parametrize("c", [384, 1536])
def test_layernorm_variants(ops, c):
    x = rnd(f"ln{c}", (300, c), -3, 3)
    g, b = rnd("lng", (c,), 0.5, 1.5), rnd("lnb", (c,), -0.2, 0.2)
    sh, sc = rnd("lnsh", (1, c), -0.5, 0.5), rnd("lnsc", (1, c), -0.5, 0.5)
    plain = ops.layernorm_rows(x.cuda(), c).float().cpu()
    assert (plain - F.layer_norm(x, (c,), eps=1e-6)).abs().max() < 3e-3
    aff = ops.layernorm_rows(x.cuda(), c, gamma=g.cuda(), beta=b.cuda(), eps=1e-5).float().cpu()
    assert (aff - F.layer_norm(x, (c,), g, b, 1e-5)).abs().max() < 4e-3
    mod = ops.layernorm_rows(x.cuda(), c, shift=sh.cuda(), scale=sc.cuda(), mod_rows=300).float().cpu()
    assert (mod - (F.layer_norm(x, (c,), eps=1e-6) * (1 + sc) + sh)).abs().max() < 4e-3


def test_embed_obs_ln(ops):
    g, n = 16, 3
    x = rnd("eo/x", (n, 2, g, g))
    w, b, pos = rnd("eo/w", (384, 2, 2, 2), -0.3, 0.3), rnd("eo/b", (384,), -0.1, 0.1), rnd("eo/p", (64, 384))
    tok, ln = ops.embed_obs_ln(x.cuda(), w.reshape(384, 8).contiguous().cuda(), b.cuda(), pos.cuda())
    ref = F.conv2d(x, w, b, stride=2).flatten(2).transpose(1, 2) + pos[None]
    assert (tok.cpu().reshape(n, 64, 384) - ref).abs().max() < 1e-5
    assert (ln.float().cpu().reshape(n, 64, 384) - F.layer_norm(ref, (384,), eps=1e-6)).abs().max() < 3e-3


def test_timestep_mlp_and_adaln(ops):
    """small_linear: sinusoid -> Linear -> SiLU -> Linear (TimestepEmbedder), then SiLU -> Linear (adaLN) with the
    input tiled x4 (FinalLayer2's t.repeat(1,4))."""
    w0, b0 = rnd("t/w0", (384, 256), -0.1, 0.1), rnd("t/b0", (384,), -0.1, 0.1)
    w2, b2 = rnd("t/w2", (384, 384), -0.1, 0.1), rnd("t/b2", (384,), -0.1, 0.1)
    wa, ba = rnd("t/wa", (3072, 1536), -0.05, 0.05), rnd("t/ba", (3072,), -0.1, 0.1)
    for t in (2.0, 1.0, 0.0, 600.0, 333.3333):
        tt = torch.tensor([[t]], dtype=torch.float32)
        th = ops.small_linear(tt.cuda(), w0.cuda(), b0.cuda(), act_in=2, act_out=1, kmod=256)
        c = ops.small_linear(th, w2.cuda(), b2.cuda())
        fin = ops.small_linear(c, wa.cuda(), ba.cuda(), act_in=1, kmod=384)
        half = 128
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        e = torch.cat([torch.cos(tt * freqs), torch.sin(tt * freqs)], dim=-1)
        c_ref = F.linear(F.silu(F.linear(e, w0, b0)), w2, b2)
        fin_ref = F.linear(F.silu(c_ref.repeat(1, 4)), wa, ba)
        assert (c.cpu() - c_ref).abs().max() < 2e-5, t
        assert (fin.cpu() - fin_ref).abs().max() < 5e-5, t


@pytest.mark.parametrize("side", [8, 7, 1, 13])
def test_dwconv_bn_relu(ops, side):
    n, c = 2, 2048
    x = rnd("dw/x", (n, c, side, side), -1, 2)
    w = rnd("dw/w", (c, 1, 3, 3), -0.5, 0.5)
    s, b = rnd("dw/s", (c,), 0.5, 1.5), rnd("dw/b", (c,), -0.3, 0.3)
    ref = F.relu(F.conv2d(x.half().float(), w, None, padding=1, groups=c) * s[None, :, None, None] + b[None, :, None, None])
    x16 = x.permute(0, 2, 3, 1).reshape(n * side * side, c).half().contiguous()
    w9c = (w.reshape(c, 9) * s[:, None]).t().contiguous()
    out = ops.dwconv3x3(x16.cuda(), w9c.cuda(), b.cuda(), n, side).float().cpu()
    got = out.reshape(n, side, side, c).permute(0, 3, 1, 2)
    assert (got - ref).abs().max() < 6e-3


@pytest.mark.parametrize("side", [8, 7, 13])
def test_dwconv_tile_kernels_equal_v1(ops, lab, monkeypatch, side):
    """Lab build: the tile kernels (4 tokens along a grid row x TY rows per thread, sliding window over the input rows; the
    product runs 2 x 4) against the one-token-per-thread kernel (DVD_DWCONV_V1=1): same tap order -> bit-identical, incl.
    odd sides (ragged last row / column group)."""
    n, c = 2, 2048
    x16 = rnd("dw/x", (n * side * side, c), -1, 2).half().contiguous()
    w9c = rnd("dw/w9", (9, c), -0.5, 0.5)
    b = rnd("dw/b", (c,), -0.3, 0.3)
    outs = {}
    for tx, ty in (("4", "1"), ("4", "2"), ("4", "4"), ("2", "1"), ("2", "2"), ("2", "4"), ("2", "8"), ("1", "4"), ("1", "8")):
        monkeypatch.setenv("DVD_DWCONV_TY", ty)
        monkeypatch.setenv("DVD_DWCONV_TX", tx)
        outs[tx + "x" + ty] = ops.dwconv3x3(x16.cuda(), w9c.cuda(), b.cuda(), n, side).float().cpu()
    monkeypatch.delenv("DVD_DWCONV_TY")
    monkeypatch.delenv("DVD_DWCONV_TX")
    monkeypatch.setenv("DVD_DWCONV_V1", "1")
    out_v1 = ops.dwconv3x3(x16.cuda(), w9c.cuda(), b.cuda(), n, side).float().cpu()
    for ty, o in outs.items():
        assert torch.equal(o, out_v1), ty


@pytest.mark.parametrize("side", [8, 7, 13, 2, 1, 41])
def test_dwconv_strip_kernels_equal_v1(ops, lab, monkeypatch, side):
    """Lab build: the sliding-window strip kernels (round 6: TX columns x SY rows per wave, a 3-row register window, every
    variant of strip width, band height and streaming stores) against the one-token-per-thread kernel: the taps of an output
    are summed in the same order with the same FMAs -> bit-identical, incl. sides that leave a narrow last strip, a short last
    band, one-column / one-row images and bands shorter than the unroll."""
    n, c = 2, 1024
    x16 = rnd("dw/x", (n * side * side, c), -1, 2).half().contiguous().cuda()
    w9c = rnd("dw/w9", (9, c), -0.5, 0.5).cuda()
    b = rnd("dw/b", (c,), -0.3, 0.3).cuda()
    monkeypatch.setenv("DVD_DWCONV_V1", "1")
    out_v1 = ops.dwconv3x3(x16, w9c, b, n, side).cpu()
    monkeypatch.delenv("DVD_DWCONV_V1")
    for tx in ("2", "3", "4"):
        for sy in ("36", "5", "3", "1"):
            for nt in (False, True):
                monkeypatch.setenv("DVD_DWCONV_STRIP", tx)
                monkeypatch.setenv("DVD_DWCONV_SY", sy)
                monkeypatch.setenv("DVD_DWCONV_NT", "1" if nt else "0")
                got = ops.dwconv3x3(x16, w9c, b, n, side).cpu()
                assert torch.equal(got.view(torch.int16), out_v1.view(torch.int16)), (tx, sy, nt)


def test_dwconv_product_rule_on_a_large_map(ops):
    """The PRODUCT library on a map large enough for its sliding-window kernel (n side^2 >= 16384 tokens, c % 512 == 0):
    against torch's depthwise conv + folded BN + ReLU, on a side (130) that leaves a one-column last strip and a 10-row last
    band; and on the engine's own shape class (c = 2048)."""
    for n, side, c in ((1, 130, 512), (2, 96, 2048)):
        x = rnd("dwl/x", (n, c, side, side), -1, 2)
        w = rnd("dwl/w", (c, 1, 3, 3), -0.5, 0.5)
        s, b = rnd("dwl/s", (c,), 0.5, 1.5), rnd("dwl/b", (c,), -0.3, 0.3)
        ref = F.relu(F.conv2d(x.half().float(), w, None, padding=1, groups=c) * s[None, :, None, None] + b[None, :, None, None])
        x16 = x.permute(0, 2, 3, 1).reshape(n * side * side, c).half().contiguous()
        w9c = (w.reshape(c, 9) * s[:, None]).t().contiguous()
        out = ops.dwconv3x3(x16.cuda(), w9c.cuda(), b.cuda(), n, side).float().cpu()
        got = out.reshape(n, side, side, c).permute(0, 3, 1, 2)
        assert (got - ref).abs().max() < 6e-3, (n, side, c)


def test_dwconv_product_rule_equals_v1_on_a_large_map(ops, lab, monkeypatch):
    """... and its bits are the one-token-per-thread kernel's (lab build: the same rule picks the strip kernel there)."""
    n, side, c = 1, 130, 512
    x16 = rnd("dwl/x16", (n * side * side, c), -1, 2).half().contiguous().cuda()
    w9c, b = rnd("dwl/w9", (9, c), -0.5, 0.5).cuda(), rnd("dwl/b2", (c,), -0.3, 0.3).cuda()
    got = ops.dwconv3x3(x16, w9c, b, n, side).cpu()
    monkeypatch.setenv("DVD_DWCONV_V1", "1")
    want = ops.dwconv3x3(x16, w9c, b, n, side).cpu()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_adaptive_posenc(ops):
    n, side, c = 2, 4, 1536
    z = rnd("pe/z", (n * side * side, c))
    hs, ws = rnd("pe/hs", (n, c), 0, 1), rnd("pe/ws", (n, c), 0, 1)
    htab, wtab = rnd("pe/ht", (side, c)), rnd("pe/wt", (side, c))
    zz = z.clone().cuda()
    pooled = ops.posenc(zz, hs.cuda(), ws.cuda(), htab.cuda(), wtab.cuda(), n, side)
    assert (pooled.cpu() - z.reshape(n, side * side, c).mean(1)).abs().max() < 1e-6
    ref = z.reshape(n, side, side, c) + hs[:, None, None, :] * htab[None, :, None, :] + ws[:, None, None, :] * wtab[None, None, :, :]
    assert (zz.cpu().reshape(n, side, side, c) - ref).abs().max() < 1e-6


def test_final_tokens(ops):
    n, g = 2, 8
    T = (g // 2) ** 2
    z = rnd("fi/z", (n * T, 1536), -2, 2)
    gam, bet = rnd("fi/g", (1536,), 0.8, 1.2), rnd("fi/b", (1536,), -0.1, 0.1)
    sh, sc = rnd("fi/sh", (1536,), -0.3, 0.3), rnd("fi/sc", (1536,), -0.3, 0.3)
    w8, b8 = rnd("fi/w8", (8, 1536), -0.05, 0.05), rnd("fi/b8", (8,), -0.1, 0.1)
    flow = rnd("fi/flow", (n, 2, g, g), -0.3, 0.3)
    x0, tok8 = ops.final_tokens(z.cuda(), gam.cuda(), bet.cuda(), sh.cuda(), sc.cuda(), w8.cuda(), b8.cuda(), flow.cuda(), n, g)
    h = F.layer_norm(F.layer_norm(z, (1536,), gam, bet, 1e-5), (1536,), eps=1e-6) * (1 + sc) + sh
    o = F.linear(h, w8, b8)
    assert (tok8.cpu() - o).abs().max() < 2e-5
    side = g // 2
    ref = torch.einsum("nhwpqc->nchpwq", o.reshape(n, side, side, 2, 2, 2)).reshape(n, 2, g, g) + flow
    assert (x0.cpu() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_build_r_rows(ops, mode):
    from oracle import dvd_oracle as O
    g, docs, hyp = 8, 2, 2
    n = docs * hyp
    feat = rnd("rr/feat", (docs, 256, g, g), 0, 2)
    flow = rnd("rr/flow", (n, 2, g, g), -0.4, 0.4)
    explicit = rnd("rr/if", (n, 256, g, g), 0, 1)
    feat_n = feat.repeat_interleave(hyp, 0)
    init_feat = {0: torch.zeros(n, 256, g, g), 1: feat_n,
                 2: O.grid_sample_ref(feat_n, (flow + O.base_grid(g, g)) * 2 - 1), 3: explicit}[mode]
    rows = ops.build_r_rows(feat.permute(0, 2, 3, 1).contiguous().cuda(), flow.cuda(), hyp, mode,
                            init_feat=explicit.cuda() if mode == 3 else None).float().cpu()
    x = torch.cat([flow, init_feat], dim=1)                                       # [n,258,g,g]
    side = g // 2
    ref = x.reshape(n, 258, side, 2, side, 2).permute(0, 2, 4, 3, 5, 1).reshape(n * side * side, 4 * 258)   # (p,q,c)
    assert (rows[:, :1032] - ref).abs().max() < 2e-3
    assert float(rows[:, 1032:].abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 40, 24), (16, 32, 9, 13), (32, 48, 17, 5), (64, 128, 37, 29), (128, 256, 16, 16),
                                          (16, 72, 7, 11), (64, 64, 130, 127), (128, 40, 36, 36)])
def test_implicit_conv_has_the_bits_of_im2col_plus_gemm(ops, cin, cout, h, w):
    """dvd_conv3x3_nhwc (the pyramid's layers since round 5: narrow kernel up to 64 output channels, the 128 x 128 GEMM with an
    implicit A operand above) against the im2col + exact-f32 GEMM pair it replaces: same K order, same K-tiles, same MFMA
    sequence - equal bits, on maps whose row count is not a multiple of a wave's 32 rows or of a tile's 128, cout ragged, and on
    both sides of the row count below which a 33..64-channel layer is computed as two 32-column halves."""
    x = rnd(f"pyi/x{cin}", (h * w, cin)).cuda()
    wt, b = rnd(f"pyi/w{cin}", (cout, cin, 3, 3), -0.3, 0.3), rnd(f"pyi/b{cin}", (cout,), -0.1, 0.1).cuda()
    wp = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().cuda()
    a = ops.conv3x3_relu_nhwc(x, wp, b, cin, cout, h, w)
    i = ops.conv3x3_relu_nhwc_implicit(x, wp, b, cin, cout, h, w)
    assert torch.equal(a, i)
    ref = F.relu(F.conv2d(x.cpu().reshape(1, h, w, cin).permute(0, 3, 1, 2), wt, b.cpu(), padding=1))[0].permute(1, 2, 0).reshape(h * w, cout)
    assert (i.cpu() - ref).abs().max() < 5e-5


def test_pyramid_pieces(ops):
    h = w = 16
    cin, cout = 8, 64
    x = rnd("py/x", (1, cin, h, w))
    wt, b = rnd("py/w", (cout, cin, 3, 3), -0.3, 0.3), rnd("py/b", (cout,), -0.1, 0.1)
    flat = wt.permute(0, 2, 3, 1).reshape(cout, -1)
    wp = torch.zeros(cout, 128)
    wp[:, :flat.shape[1]] = flat
    out = ops.conv3x3_relu_nhwc(x[0].permute(1, 2, 0).contiguous().cuda(), wp.cuda(), b.cuda(), cin, cout, h, w)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))[0].permute(1, 2, 0).reshape(h * w, cout)
    assert (out.cpu() - ref).abs().max() < 2e-5
    pooled = ops.maxpool2_nhwc(out, cout, h, w).cpu()
    ref_p = F.max_pool2d(out.cpu().reshape(h, w, cout).permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0).reshape(-1, cout)
    assert torch.equal(pooled, ref_p)          # max-pool of the kernel's own conv output is exact
    up = ops.resize_bilinear_nhwc(pooled.cuda(), cout, h // 2, w // 2, 18, 18).cpu()
    ref_u = F.interpolate(ref_p.reshape(h // 2, w // 2, cout).permute(2, 0, 1)[None], size=(18, 18), mode="bilinear",
                          align_corners=True)[0].permute(1, 2, 0).reshape(-1, cout)
    assert (up - ref_u).abs().max() < 1e-5
