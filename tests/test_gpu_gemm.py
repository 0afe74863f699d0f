"""GPU parity of the MFMA GEMM (f16 and exact-f32 variants, every epilogue) vs a float64 CPU
reference of the same op."""
import os

import numpy as np
import pytest
import torch

from dvd_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dvd_amd import ops as _ops
    return _ops


def rnd(tag, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(synth.uniform(tag, shape, lo, hi, 11))


def gelu_tanh(x):
    return 0.5 * x * (1 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 384), (200, 130, 128), (1024, 1152, 384), (64, 2048, 1536),
                                   (4096, 2048, 128), (32768, 3072, 64)])
@pytest.mark.parametrize("dt", ["f16", "f32"])
def test_gemm_plain(ops, M, N, K, dt):
    a, b = rnd(f"ga{M}{K}", (M, K)), rnd(f"gb{N}{K}", (N, K))
    if dt == "f16":
        a, b = a.half(), b.half()
    ref = a.double() @ b.double().t()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out)
    err = (out.cpu().double() - ref).abs().max().item()
    tol = 1e-4 * K ** 0.5 if dt == "f16" else 2e-7 * K
    assert err < tol, err


def test_gemm_exact_integers(ops):
    rng = np.random.RandomState(3)
    a = torch.from_numpy(rng.randint(-4, 5, (256, 192)).astype(np.float32))
    b = torch.from_numpy(rng.randint(-4, 5, (384, 192)).astype(np.float32))   # asymmetric, non-square
    ref = a @ b.t()
    for cast in (torch.float16, torch.float32):
        out = torch.zeros(256, 384, device="cuda")
        ops.gemm_nt(a.to(cast).cuda(), b.to(cast).cuda(), out32=out)
        assert torch.equal(out.cpu(), ref)


def test_gemm_epilogue_all(ops):
    M, N, K, T = 512, 384, 256, 128          # 4 samples of T rows
    a, b = rnd("ea", (M, K)).half(), rnd("eb", (N, K)).half()
    bias, pos = rnd("ebias", (N,)), rnd("epos", (T, N))
    gate, res = rnd("egate", (M // T, N)), rnd("eres", (M, N))
    acc = a.double() @ b.double().t() + bias.double()
    # bias + GELU -> f16
    out16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out16=out16, bias=bias.cuda(), act=1)
    e1 = (out16.cpu().double() - gelu_tanh(acc)).abs().max().item()
    assert e1 < 1.5e-2, ("gelu f16", e1)      # f16 store of values up to ~16: half-ulp 4e-3 + f16 operand rounding
    # bias + ReLU + pos
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out, bias=bias.cuda(), act=2, pos=pos.cuda())
    ref = torch.relu(acc) + pos.double().repeat(M // T, 1)
    e2 = (out.cpu().double() - ref).abs().max().item()
    assert e2 < 2e-3, ("relu+pos", e2)
    # gate * (acc + bias) + residual, in place on the residual buffer, strided output (ld = 2N)
    big = torch.zeros(M, 2 * N, device="cuda")
    big[:, N:] = res.cuda()
    view = big[:, N:]
    ops.gemm_nt(a.cuda(), b.cuda(), out32=view, bias=bias.cuda(), gate=gate.cuda(), gate_rows=T, res=view)
    ref = gate.double().repeat_interleave(T, 0) * acc + res.double()
    e3 = (view.cpu().double() - ref).abs().max().item()
    assert e3 < 2e-3, ("gate+res", e3)
    assert float(big[:, :N].abs().max()) == 0.0


def test_gemm_batched_swapped_rowbias(ops):
    """V^T = W . X_z^T per sample z (weights as the A operand, per-row bias): the layout the
    attention kernel consumes."""
    Z, T, K, Nw = 3, 192, 128, 256
    x, w, bias = rnd("bx", (Z * T, K)).half(), rnd("bw", (Nw, K)).half(), rnd("bb", (Nw,))
    out = torch.zeros(Z, Nw, T, dtype=torch.float16, device="cuda")
    ops.gemm_nt(w.cuda(), x.cuda(), out16=out, bias=bias.cuda(), bias_row=True, batch=Z,
                strides={"B": T * K, "C16": Nw * T}, M=Nw, N=T, K=K, lda=K, ldb=K)
    for z in range(Z):
        ref = w.double() @ x[z * T:(z + 1) * T].double().t() + bias.double()[:, None]
        assert (out[z].cpu().double() - ref).abs().max() < 2e-2


def test_gemm_split_weights(ops):
    """W = hi + 2^-11 lo (both f16): result matches the fp32-weight product to activation-rounding level."""
    M, N, K = 256, 384, 1536
    a = rnd("sa", (M, K)).half()
    w = rnd("sw", (N, K)) * 0.05
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    ref = a.double() @ w.double().t()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=out)
    e_plain = (out.cpu().double() - ref).abs().max().item()
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=out, b_lo=lo.cuda())
    e_split = (out.cpu().double() - ref).abs().max().item()
    out_t = torch.zeros(N, M, device="cuda")
    ops.gemm_nt(hi.cuda(), a.cuda(), out32=out_t, a_lo=lo.cuda())
    e_split_a = (out_t.cpu().double().t() - ref).abs().max().item()
    print("split weights", e_plain, e_split, e_split_a)
    assert e_split < e_plain / 20 and e_split < 2e-5 and e_split_a < 2e-5, (e_plain, e_split, e_split_a)


@pytest.mark.parametrize("M,N,K", [(1024, 1536, 1536), (1100, 512, 256), (2048, 3072, 1536), (4096, 256, 64)])
def test_gemm_large_tile_kernel(ops, M, N, K):
    """Shapes routed to the 256x256 LDS-DMA kernel (N % 256 == 0, M >= 1024), incl. a ragged M."""
    a, b = rnd(f"la{M}{K}", (M, K)).half(), rnd(f"lb{N}{K}", (N, K)).half()
    ref = a.double() @ b.double().t()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * K ** 0.5, err


def test_gemm_large_tile_exact_and_epilogue(ops):
    rng = np.random.RandomState(5)
    M, N, K, T = 1536, 512, 192, 512
    a = torch.from_numpy(rng.randint(-4, 5, (M, K)).astype(np.float32)).half()
    b = torch.from_numpy(rng.randint(-4, 5, (N, K)).astype(np.float32)).half()
    ref = a.float() @ b.float().t()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out)
    assert torch.equal(out.cpu(), ref)
    bias, gate, res = rnd("lbias", (N,)), rnd("lgate", (M // T, N)), rnd("lres", (M, N))
    out2 = res.clone().cuda()
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out2, bias=bias.cuda(), gate=gate.cuda(), gate_rows=T, res=out2)
    ref2 = gate.double().repeat_interleave(T, 0) * (ref.double() + bias.double()) + res.double()
    assert (out2.cpu().double() - ref2).abs().max() < 1e-3
    out16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out16=out16, bias=bias.cuda(), act=2)
    assert (out16.cpu().double() - torch.relu(ref.double() + bias.double())).abs().max() < 0.3   # f16 store of |v| <= 400


def test_gemm_large_tile_split_weights(ops):
    M, N, K = 2048, 1536, 1536
    a = rnd("lsa", (M, K)).half()
    w = rnd("lsw", (N, K)) * 0.05
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    ref = a.double() @ w.double().t()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=out, b_lo=lo.cuda())
    e_b = (out.cpu().double() - ref).abs().max().item()
    out_t = torch.zeros(N, M, device="cuda")
    ops.gemm_nt(hi.cuda(), a.cuda(), out32=out_t, a_lo=lo.cuda())       # weight on the A side, N = M % 256 == 0
    e_a = (out_t.cpu().double().t() - ref).abs().max().item()
    assert e_b < 2e-5 and e_a < 2e-5, (e_b, e_a)


@pytest.mark.parametrize("M,N,K", [(2048, 1536, 1536), (1100, 512, 128), (4096, 256, 64), (1024, 2048, 1536),
                                   (1100, 384, 384), (2048, 1152, 384), (300, 128, 64),
                                   # ntm % 8 == 0 and ntn in {8, 12}: the blocked XCD tile walk (8 row panels x 4 N tiles),
                                   # incl. an XCD range with leftover row panels (M = 6144: 3 per XCD)
                                   (2048, 2048, 128), (4096, 3072, 64), (6144, 2048, 64), (32768, 3072, 64)])
def test_gemm_fused_split_kernel(ops, M, N, K):
    """lo_scale = 1 with an UNSCALED lo part (f16 subnormals) routes to the one-pass kernel (three LDS tiles, one
    accumulator set).  Result: fp32-weight grade, and equal to the two-pass kernel up to fp32 summation order;
    residual + bias + ReLU epilogue and a ragged M included."""
    a = rnd(f"fa{M}{K}", (M, K)).half()
    w = rnd(f"fw{N}{K}", (N, K)) * 0.05
    hi = w.half()
    lo = (w - hi.float()).half()                      # mostly subnormal
    assert (lo.float().abs() < 6.2e-5).float().mean() > 0.9
    ref = a.double() @ w.double().t()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=out, b_lo=lo.cuda(), lo_scale=1.0)
    e_fused = (out.cpu().double() - ref).abs().max().item()
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=out)
    e_plain = (out.cpu().double() - ref).abs().max().item()
    # the two-pass path of the product: a SCALED lo part (lo_scale != 1) sweeps K twice
    out2 = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=out2, b_lo=((w - hi.float()) * 2048.0).half().cuda(), lo_scale=2.0 ** -11)
    e_two = (out2.cpu().double() - ref).abs().max().item()
    print("fused split", M, N, K, e_plain, e_fused, e_two)
    assert e_fused < 3e-5 * (K / 1536) ** 0.5 + 2e-6 and e_fused < e_plain / 10, (e_plain, e_fused, e_two)
    assert e_two < 3e-5
    bias, res = rnd("fbias", (N,)), rnd("fres", (M, N))
    o3 = res.clone().cuda()
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=o3, b_lo=lo.cuda(), lo_scale=1.0, bias=bias.cuda(), act=2, res=o3)
    ref3 = torch.relu(ref + bias.double()) + res.double()
    assert (o3.cpu().double() - ref3).abs().max() < 5e-5
    # gate * (acc + bias) + residual and the positional add (the DiT block's epilogues)
    T = 100
    gate = rnd("fgate", ((M + T - 1) // T, N))
    pos = rnd("fpos", (T, N))
    o4 = res.clone().cuda()
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=o4, b_lo=lo.cuda(), lo_scale=1.0, bias=bias.cuda(), gate=gate.cuda(),
                gate_rows=T, res=o4)
    ref4 = gate.double().repeat_interleave(T, 0)[:M] * (ref + bias.double()) + res.double()
    assert (o4.cpu().double() - ref4).abs().max() < 5e-5
    o5 = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), hi.cuda(), out32=o5, b_lo=lo.cuda(), lo_scale=1.0, bias=bias.cuda(), pos=pos.cuda())
    ref5 = ref + bias.double() + pos.double().repeat((M + T - 1) // T, 1)[:M]
    assert (o5.cpu().double() - ref5).abs().max() < 5e-5


@pytest.mark.parametrize("env", [{}, {"DVD_GEMM_TWOPASS": "1"}, {"DVD_GEMM_V1": "1"}, {"DVD_GEMM_SPREAD": "1"},
                                 {"DVD_GEMM_SCALAR_EPILOGUE": "1"}, {"DVD_GEMM_NONPERSISTENT": "1"}],
                         ids=lambda e: next(iter(e), "default"))
def test_gemm_kernel_variants(ops, lab, monkeypatch, env):
    """LAB build: every GEMM path behind a DVD_GEMM_* switch: split weights (scaled and unscaled lo), residual + bias + ReLU, f16
    output with GELU, ragged M - against float64."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    M, N, K = 1100, 512, 256
    a = rnd("va", (M, K)).half()
    w = rnd("vw", (N, K)) * 0.05
    hi = w.half()
    ref = a.double() @ w.double().t()
    bias, res = rnd("vbias", (N,)), rnd("vres", (M, N))
    for lo, scale in (((w - hi.float()).half(), 1.0), (((w - hi.float()) * 2048.0).half(), 2.0 ** -11)):
        out = res.clone().cuda()
        ops.gemm_nt(a.cuda(), hi.cuda(), out32=out, b_lo=lo.cuda(), lo_scale=scale, bias=bias.cuda(), act=2, res=out)
        ref2 = torch.relu(ref + bias.double()) + res.double()
        assert (out.cpu().double() - ref2).abs().max() < 2e-5, (env, scale)
    out16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    ops.gemm_nt(a.cuda(), hi.cuda(), out16=out16, bias=bias.cuda(), act=1)
    refg = torch.nn.functional.gelu((a.double() @ hi.double().t()) + bias.double(), approximate="tanh")
    assert (out16.cpu().double() - refg).abs().max() < 2e-3, env


def test_gemm_weights_on_a_side_column_major_walk(ops):
    """The V^T projection's shape family - split weights on the A side (two sweeps over K), few row panels (ntm = 6) and many
    N tiles (ntn = 36 >= 32): the column-major XCD tile walk - batched, against float64."""
    D, T, K, B = 1536, 9216, 128, 2
    w = rnd("vt/w", (D, K)) * 0.05
    hi = w.half()
    lo = (w - hi.float()).half()
    h = rnd("vt/h", (B * T, K)).half()
    out = torch.zeros(B, D, T, dtype=torch.float16, device="cuda")
    ops.gemm_nt(hi.cuda(), h.cuda(), out16=out.view(B * D, T), a_lo=lo.cuda(), lo_scale=1.0, batch=B, M=D, N=T, K=K,
                lda=K, ldb=K, strides={"B": T * K, "C16": D * T})
    for b in range(B):
        ref = w.double() @ h[b * T:(b + 1) * T].double().t()
        err = (out[b].cpu().double() - ref).abs().max().item()
        assert err < 2e-3 * max(1.0, ref.abs().max().item()), (b, err)          # f16 output rounding dominates


@pytest.mark.parametrize("M,N,K", [(81, 16, 144), (82944, 16, 576), (1296, 1, 576), (5000, 32, 48), (300, 6, 16), (324, 16, 1152),
                                   (20736, 64, 288), (333, 40, 64)])
def test_gemm_f32_narrow_outputs(ops, M, N, K):
    """The pre-stage conv nets' shape family (N <= 64, exact f32): one wave per 32 rows x 32 / 64 columns, operands straight from global
    memory; bias + ReLU epilogue; ragged M, N = 1."""
    a, b = rnd(f"na{M}{K}", (M, K)), rnd(f"nb{N}{K}", (N, K))
    bias = rnd(f"nbias{N}", (N,))
    ref = a.double() @ b.double().t() + bias.double()
    out = torch.full((M, N), 7.0, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out, bias=bias.cuda(), act=2)
    err = (out.cpu().double() - torch.relu(ref)).abs().max().item()
    assert err < 2e-7 * K * 4, err
    out2 = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out2)
    assert (out2.cpu().double() - (ref - bias.double())).abs().max().item() < 2e-7 * K * 4


# ---------------------------------------------------------------------------------------------------------
# temporal dithering of the f16 weight rounding (dvd_dither_f16)
# ---------------------------------------------------------------------------------------------------------
from dither_ref import dither_ref  # noqa: E402  (tests/dither_ref.py: the numpy restatement, shared with the CPU property tests)


@pytest.mark.parametrize("family", ["weights", "tiny", "edge"])
def test_dither_bit_exact_vs_numpy(family):
    from dvd_amd import ops
    n = 1 << 16
    if family == "weights":
        w = synth.uniform("dith/w", (n,), -0.05, 0.05, 3)
    elif family == "tiny":                                 # subnormal f16 range and exact zeros
        w = synth.uniform("dith/t", (n,), -1e-6, 1e-6, 3)
        w[::7] = 0.0
    else:                                                  # powers of two (the gap halves below), exact f16 values, large
        w = synth.uniform("dith/e", (n,), -4.0, 4.0, 3)
        w[::5] = np.float32(0.5) - np.float32(1e-5)
        w[1::5] = np.float16(0.3).astype(np.float32)
        w[2::5] = -np.float32(2.0) + np.float32(3e-4)
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    for step, elem0 in ((0, 0), (1, 0), (17, 123456), (49, 4096)):
        got = ops.dither_f16(torch.from_numpy(hi).cuda(), torch.from_numpy(lo).cuda(), step, elem0).cpu().numpy()
        want = dither_ref(hi, lo, elem0, step)
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (family, step, elem0)


def test_dither_mean_converges_to_the_weight():
    """The point of the dithering: the mean of the re-rounded weight over S consecutive steps approaches hi + lo like
    log(S)/S, where the fixed rounding (hi alone) keeps its full error at every step."""
    from dvd_amd import ops
    n = 1 << 16
    w = synth.uniform("dith/m", (n,), -0.05, 0.05, 5)
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    target = hi.astype(np.float64) + lo.astype(np.float64)
    h, l = torch.from_numpy(hi).cuda(), torch.from_numpy(lo).cuda()
    fixed = float(np.sqrt(np.mean((hi.astype(np.float64) - target) ** 2)))
    for S, factor in ((10, 4.0), (50, 15.0), (250, 60.0)):
        acc = torch.zeros(n, dtype=torch.float64, device="cuda")
        for s in range(S):
            acc += ops.dither_f16(h, l, s).double()
        err = float(np.sqrt(np.mean((acc.cpu().numpy() / S - target) ** 2)))
        print(f"dither mean over {S} steps: rms error {err:.3e} vs fixed rounding {fixed:.3e} ({fixed / err:.1f}x)")
        assert err * factor < fixed, (S, err, fixed)


def test_gelu_epilogue_range(ops):
    """The one-exponential GELU of the GEMM epilogue (gemm.hip:gelu_tanh) over [-12, 12], through the region where
    exp(-2u) overflows (x below about -10.1) and at +-1e4 / +inf, against torch's gelu(approximate='tanh') on the same f32 inputs:
    the GEMM is x * 1 (exact in the f32-input kernel), so the epilogue is all that is measured (round-3 ADVICE)."""
    xs = torch.cat([torch.linspace(-12, 12, 4093), torch.tensor([-1e4, 1e4, float("inf"), -0.0, 0.0, -10.5, -10.4999, -5.5])])
    M, K, N = 4224, 16, 64
    pad = M - xs.numel()
    x = torch.cat([xs, torch.zeros(pad)])
    a = torch.zeros(M, K)
    a[:, 0] = x
    b = torch.zeros(N, K)
    b[:, 0] = 1.0
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out, act=1)
    ref = torch.nn.functional.gelu(x.double(), approximate="tanh")
    got = out.cpu()[:, 0].double()
    fin = torch.isfinite(ref)
    err = (got[fin] - ref[fin]).abs()
    tol = 4e-7 * ref[fin].abs().clamp_min(1.0)       # v_exp_f32 / v_rcp_f32 are ~1 ulp each
    assert bool((err <= tol).all()), float((err / tol).max())
    assert bool(torch.equal(got[~fin], ref[~fin]))   # +inf -> +inf
    assert bool((out.cpu()[:, 1:] == out.cpu()[:, :1]).all())   # every column saw the same value
    neg = got[(x < -10.5)]
    assert bool((neg == 0).all()) and bool(torch.signbit(neg).all())   # exactly -0, as the tanh form in fp32


@pytest.mark.parametrize("M,N,K", [(1024, 1536, 1536), (1100, 512, 256), (2048, 3072, 1536), (4096, 256, 64), (777, 256, 192)])
def test_gemm_big2_kernel(ops, lab, monkeypatch, M, N, K):
    """LAB switch DVD_GEMM_BIG2: the one-wave-per-SIMD 256 x 256 kernel (gemm_nt_big2_kernel: 4 waves x 128 x 128, half-slab
    ring of four, pinned asm stream; measured 4-9 % slower than the 8-wave kernel and kept in the lab build only) on the
    large-tile shapes incl. ragged M and K = 64 (a ring shorter than its depth): exact integers, then random data with the plain, bias + GELU -> f16 and residual epilogues."""
    monkeypatch.setenv("DVD_GEMM_BIG2", "1")
    rng = np.random.RandomState(M + N + K)
    ai = torch.from_numpy(rng.randint(-4, 5, (M, K)).astype(np.float32)).half()
    bi = torch.from_numpy(rng.randint(-4, 5, (N, K)).astype(np.float32)).half()      # asymmetric, non-square
    out = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(ai.cuda(), bi.cuda(), out32=out)
    assert torch.equal(out.cpu(), ai.float() @ bi.float().t())
    a, b = rnd(f"b2a{M}{K}", (M, K)).half(), rnd(f"b2b{N}{K}", (N, K)).half()
    ref = a.double() @ b.double().t()
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out)
    assert (out.cpu().double() - ref).abs().max().item() < 1e-4 * K ** 0.5
    bias, res = rnd("b2bias", (N,)), rnd("b2res", (M, N))
    out16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out16=out16, bias=bias.cuda(), act=1)
    refg = torch.nn.functional.gelu(ref + bias.double(), approximate="tanh")
    assert (out16.cpu().double() - refg).abs().max().item() < 2e-2
    out2 = res.clone().cuda()
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out2, bias=bias.cuda(), act=2, res=out2)
    ref2 = torch.relu(ref + bias.double()) + res.double()
    assert (out2.cpu().double() - ref2).abs().max().item() < 1e-4 * K ** 0.5 + 1e-5
    # and the same bits as the 8-wave kernel would give?  No: the K order inside a tile is the same (k ascending), so yes
    monkeypatch.delenv("DVD_GEMM_BIG2")
    out8 = torch.zeros(M, N, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out32=out8)
    assert torch.equal(out8, out), "big2 and the 8-wave kernel accumulate k in the same order: equal bits expected"


@pytest.mark.parametrize("side", ["B", "A"])
def test_gemm_small_family_many_rows_same_bits(ops, side):
    """dvd_gemm_desc.small_tiles = 2 (a LARGE batch of small grids: N % 256 == 0 shapes on the 256x256 kernel, two sweeps)
    must give the bits of small_tiles = 1 (128x128 tiles everywhere) - a document sampled alone and in a batch of 32 then
    agree bit for bit although the batch runs the faster kernel - for the (hi, lo) weight pair on either side and the
    epilogues the engine uses (bias + ReLU -> f16, residual -> f32, plain f16)."""
    M, N, K = 2304, 1536, 1536
    a = rnd("sm/a", (M, K)).half()
    w = rnd("sm/w", (N, K)) * 0.05
    hi = w.half()
    lo = (w - hi.float()).half()
    bias, res = rnd("sm/bias", (N,)), rnd("sm/res", (M, N))
    for st in (1, 2):
        if side == "B":
            o16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
            ops.gemm_nt(a.cuda(), hi.cuda(), out16=o16, b_lo=lo.cuda(), lo_scale=1.0, bias=bias.cuda(), act=2, small_tiles=st)
            o32 = res.clone().cuda()
            ops.gemm_nt(a.cuda(), hi.cuda(), out32=o32, b_lo=lo.cuda(), lo_scale=1.0, res=o32, small_tiles=st)
        else:          # the V^T projections: weights on the A side, the output transposed
            o16 = torch.zeros(N, M, dtype=torch.float16, device="cuda")
            ops.gemm_nt(hi.cuda(), a.cuda(), out16=o16, a_lo=lo.cuda(), lo_scale=1.0, small_tiles=st)
            o32 = torch.zeros(N, M, device="cuda")
            ops.gemm_nt(hi.cuda(), a.cuda(), out32=o32, a_lo=lo.cuda(), lo_scale=1.0, small_tiles=st)
        if st == 1:
            ref16, ref32 = o16.clone(), o32.clone()
    assert torch.equal(o16, ref16) and torch.equal(o32, ref32)
    want = (a.double() @ w.double().t())
    got = o32.cpu().double() - (res.double() if side == "B" else 0)
    assert ((got if side == "B" else got.t()) - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("M", [16384, 2000])
@pytest.mark.parametrize("N,K,epi", [(768, 384, "bias"), (1536, 384, "gelu"), (3072, 1536, "plain"), (1536, 1536, "res"),
                                     (2048, 1536, "relu"), (1536, 2048, "relu_res")])
def test_gemm_small_family_engine_shapes_same_bits(ops, M, N, K, epi):
    """from 16 384 token rows on, the small-grid engine sends its N % 256 == 0 GEMMs to the 256 x 256
    kernel (small_tiles = 2, two sweeps) while a single document runs 128 x 128 tiles (small_tiles = 1) - 'a document gives the
    same bits alone and in a batch' then rests on the two kernels accumulating identically.  Every (N, K) of the engine's
    small family with the epilogue it carries there (engine.hip:enqueue_step), at a batch-sized M and at a ragged one."""
    a = rnd(f"se/a{K}", (M, K)).half().cuda()
    w = rnd(f"se/w{N}{K}", (N, K)) * 0.05
    hi, lo = w.half().cuda(), (w - w.half().float()).half().cuda()
    bias, res = rnd("se/bias", (N,)).cuda(), rnd(f"se/res{N}", (M, N)).cuda()
    outs = []
    for st in (1, 2):
        kw = dict(b_lo=lo, lo_scale=1.0, small_tiles=st)
        if epi in ("res", "relu_res"):
            o = res.clone()
            ops.gemm_nt(a, hi, out32=o, res=o, **(dict(bias=bias, act=2) if epi == "relu_res" else {}), **kw)
        else:
            o = torch.zeros(M, N, dtype=torch.float16, device="cuda")
            extra = {"bias": dict(bias=bias), "gelu": dict(bias=bias, act=1), "relu": dict(bias=bias, act=2), "plain": {}}[epi]
            ops.gemm_nt(a, hi, out16=o, **extra, **kw)
        outs.append(o.cpu())
    assert torch.equal(outs[0], outs[1]), (N, K, epi, M)


@pytest.mark.parametrize("M,N,K", [(1536, 1536, 1536), (1100, 512, 256), (2304, 3072, 1536), (777, 256, 384), (4000, 2048, 1536),
                                   (1152, 1536, 2048), (384, 256, 256), (5, 768, 384)])
def test_gemm_t384_kernel(ops, lab, monkeypatch, M, N, K):
    """Round 5's 384 x 256 kernel (gemm_nt_t384_kernel: 4-slot half-slab ring, generated K loop; N % 256 == 0, K % 128 == 0,
    K >= 256) on full and ragged row tiles (M % 384 != 0, M < one tile), the ring's shortest K (first + final block only) and the
    steady loop: exact integers, random data vs float64, plain / bias + ReLU -> f16 / bias + GELU -> f16 / residual / bias +
    ReLU + residual (in place) epilogues.  Round 6: the product loop runs on v_mfma_f32_16x16x32_f16 (16 x 16 accumulator
    quads, turned into rows of 32 columns by v_permlane16_swap in every epilogue): exact on integers, float64-close on random
    data, and its five epilogues agree with each other bit for bit; round 5's 32x32x16 loop (lab switch DVD_GEMM_T384_M32) still
    has THE SAME BITS as gemm_nt_big_kernel (DVD_GEMM_NO_T384): same MFMA sequence per accumulator, same epilogue arithmetic."""
    rng = np.random.RandomState(M + N + K)
    ai = torch.from_numpy(rng.randint(-4, 5, (M, K)).astype(np.float32)).half()
    bi = torch.from_numpy(rng.randint(-4, 5, (N, K)).astype(np.float32)).half()
    out = torch.full((M, N), 7.0, device="cuda")
    ops.gemm_nt(ai.cuda(), bi.cuda(), out32=out)
    assert torch.equal(out.cpu(), ai.float() @ bi.float().t())
    a, b = rnd(f"t3a{M}{K}", (M, K)).half().cuda(), rnd(f"t3b{N}{K}", (N, K)).half().cuda()
    bias, res = rnd("t3bias", (N,)).cuda(), rnd("t3res", (M, N)).cuda()
    ref = a.cpu().double() @ b.cpu().double().t()

    def run_all():
        o32 = torch.zeros(M, N, device="cuda")
        ops.gemm_nt(a, b, out32=o32)
        o16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        ops.gemm_nt(a, b, out16=o16, bias=bias, act=2)
        g16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        ops.gemm_nt(a, b, out16=g16, bias=bias, act=1)
        r32 = res.clone()
        ops.gemm_nt(a, b, out32=r32, res=r32)
        br32 = res.clone()
        ops.gemm_nt(a, b, out32=br32, bias=bias, act=2, res=br32)
        both = torch.zeros(M, N, device="cuda")
        both16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        ops.gemm_nt(a, b, out32=both, out16=both16, bias=bias)
        torch.cuda.synchronize()
        return [t.cpu() for t in (o32, o16, g16, r32, br32, both, both16)]

    def check(outs, what):
        tol = 1e-4 * K ** 0.5
        assert (outs[0].double() - ref).abs().max().item() < tol, what
        assert (outs[1].double() - torch.relu(ref + bias.cpu().double())).abs().max().item() < 4e-2, what
        assert (outs[2].double() - torch.nn.functional.gelu(ref + bias.cpu().double(), approximate="tanh")).abs().max().item() < 4e-2, what
        assert (outs[3].double() - (ref + res.cpu().double())).abs().max().item() < tol + 1e-5, what
        assert (outs[4].double() - (torch.relu(ref + bias.cpu().double()) + res.cpu().double())).abs().max().item() < tol + 1e-5, what
        assert (outs[5].double() - (ref + bias.cpu().double())).abs().max().item() < tol + 1e-5, what
        # one accumulator, five epilogues (LDS-free f32 / f16, residual, the staged generic form): the same sums everywhere
        assert torch.equal(outs[3], outs[0] + res.cpu()), what                                   # residual flavour = plain + res
        assert torch.equal(outs[4], torch.relu(outs[0] + bias.cpu()) + res.cpu()), what
        assert torch.equal(outs[5], outs[0] + bias.cpu()), what                                  # staged generic form
        assert torch.equal(outs[6], outs[5].half()) and torch.equal(outs[1], torch.relu(outs[5]).half()), what

    new = run_all()                                    # the product loop: v_mfma_f32_16x16x32_f16 (round 6)
    check(new, "16x16x32 loop")
    monkeypatch.setenv("DVD_GEMM_T384_M32", "1")       # round 5's 32x32x16 loop (lab)
    m32 = run_all()
    check(m32, "32x32x16 loop")
    assert (new[0] - m32[0]).abs().max().item() < 1e-4 * K ** 0.5          # other MFMA shape, other summation grouping: close
    monkeypatch.setenv("DVD_GEMM_NO_T384", "1")
    old = run_all()
    for i, (x, y) in enumerate(zip(m32, old)):
        assert torch.equal(x, y), f"output {i}: the 32x32x16 t384 loop and the 256 x 256 kernel must give the same bits"


def test_gemm_t384_batched_weights_on_a_side(ops, lab, monkeypatch):
    """The decoder's V^T projection form: weights on the A side (M = 1536 = 4 row tiles of 384), one batch entry per sample,
    N = tokens (a multiple of 256), f16 output with ldc = tokens; and an odd steady-loop count (K = 640)."""
    Bn, M, N, K = 3, 1536, 1280, 640
    w = rnd("t3vw", (M, K)).half().cuda()
    x = rnd("t3vx", (Bn, N, K)).half().cuda()
    out = torch.zeros(Bn, M, N, dtype=torch.float16, device="cuda")
    ops.gemm_nt(w, x, out16=out, batch=Bn, N=N, strides={"B": N * K, "C16": M * N})
    ref = torch.einsum("mk,bnk->bmn", w.cpu().double(), x.cpu().double())
    assert (out.cpu().double() - ref).abs().max().item() < 4e-2
    monkeypatch.setenv("DVD_GEMM_T384_M32", "1")
    m32 = torch.zeros_like(out)
    ops.gemm_nt(w, x, out16=m32, batch=Bn, N=N, strides={"B": N * K, "C16": M * N})
    assert (m32.cpu().double() - ref).abs().max().item() < 4e-2
    monkeypatch.setenv("DVD_GEMM_NO_T384", "1")
    old = torch.zeros_like(out)
    ops.gemm_nt(w, x, out16=old, batch=Bn, N=N, strides={"B": N * K, "C16": M * N})
    assert torch.equal(m32.cpu(), old.cpu())


@pytest.mark.parametrize("M,N,K", [(2048, 1536, 1536), (2048, 384, 384), (300, 200, 128), (1024, 1088, 64), (77, 130, 192),
                                   (4096, 640, 1088)])
def test_gemm_eight_wave_kernel(ops, lab, monkeypatch, M, N, K):
    """gemm_nt_w8_kernel (round 5: the 128 x 128 tile by two waves per SIMD, taken by f16 problems with few tiles - the sampler at
    the reference's operating point) on the small family's calls: (hi, lo) weight pairs and single tensors, ragged M and N (scalar
    epilogue path when N % 8 != 0), one K-tile (K = 64) and odd tile counts, bias / GELU -> f16, residual in place, positional
    rows + gate, row bias (transposed outputs): against float64, and THE SAME BITS as the 4-wave kernel (lab switch
    DVD_GEMM_W8=0), which runs the same MFMA sequence per accumulator and the same epilogue arithmetic."""
    a = rnd(f"w8a{M}{K}", (M, K)).half().cuda()
    w = rnd(f"w8b{N}{K}", (N, K)) * 0.2
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    hi, lo = hi.cuda(), lo.cuda()
    bias, res = rnd("w8bias", (N,)).cuda(), rnd(f"w8res{M}{N}", (M, N)).cuda()
    brow = rnd("w8brow", (M,)).cuda()
    pos, gate = rnd(f"w8pos{N}", (64, N)).cuda(), rnd(f"w8gate{N}", (2, N)).cuda()
    ref = a.cpu().double() @ (hi.cpu().double() + lo.cpu().double() / 2048.0).t()

    def run_all():
        kw = dict(b_lo=lo, small_tiles=True)
        o32 = torch.zeros(M, N, device="cuda")
        ops.gemm_nt(a, hi, out32=o32, **kw)
        g16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        ops.gemm_nt(a, hi, out16=g16, bias=bias, act=1, **kw)
        r32 = res.clone()
        ops.gemm_nt(a, hi, out32=r32, res=r32, bias=bias, **kw)
        pg = torch.zeros(M, N, device="cuda")
        ops.gemm_nt(a, hi, out32=pg, bias=bias, pos=pos, gate=gate, gate_rows=(M + 1) // 2, res=res, **kw)
        t16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        ops.gemm_nt(a, hi, out16=t16, bias=brow, bias_row=True, small_tiles=True)          # single tensor, row bias
        torch.cuda.synchronize()
        return [t.cpu() for t in (o32, g16, r32, pg, t16)]

    monkeypatch.setenv("DVD_GEMM_RING128", "0")       # the register-staged 4-wave kernel as the reference of both variants
    monkeypatch.setenv("DVD_GEMM_RING256", "0")
    monkeypatch.setenv("DVD_GEMM_W8", "1")
    new = run_all()
    tol = 2e-4 * K ** 0.5
    assert (new[0].double() - ref).abs().max().item() < tol
    assert (new[1].double() - torch.nn.functional.gelu(ref + bias.cpu().double(), approximate="tanh")).abs().max().item() < 4e-2
    assert (new[2].double() - (ref + bias.cpu().double() + res.cpu().double())).abs().max().item() < tol + 1e-5
    rows = torch.arange(M)
    pgref = (ref + bias.cpu().double() + pos.cpu().double()[rows % 64]) * gate.cpu().double()[rows // ((M + 1) // 2)] + res.cpu().double()
    assert (new[3].double() - pgref).abs().max().item() < 2 * tol + 1e-5
    ref1 = a.cpu().double() @ hi.cpu().double().t() + brow.cpu().double()[:, None]
    assert (new[4].double() - ref1).abs().max().item() < 4e-2
    monkeypatch.setenv("DVD_GEMM_W8", "0")
    old = run_all()
    for i, (x, y) in enumerate(zip(new, old)):
        assert torch.equal(x, y), f"output {i}: the 8-wave and the 4-wave kernel must give the same bits"


@pytest.mark.parametrize("M,N,K", [(2048, 1536, 1536), (2048, 384, 384), (300, 200, 128), (1024, 1088, 192), (77, 130, 192),
                                   (4096, 640, 1088), (2048, 3072, 1536), (128, 128, 128), (2050, 1536, 2048), (256, 512, 128), (1000, 768, 192)])
def test_gemm_ring128_kernel(ops, lab, monkeypatch, M, N, K):
    """gemm_nt_ring128_kernel (round 5: the 128 x 128 tile fed by LDS-DMA through a five-slab ring, taken by f16 problems with
    few tiles - the sampler at the reference's operating point) on the small family's calls: (hi, lo) weight pairs and single
    tensors, ragged M and N (scalar epilogue path when N % 8 != 0), the shortest K (two slabs: prologue and clamped tail only),
    odd slab counts, bias / GELU -> f16, residual in place, positional rows + gate, row bias: against float64, and THE SAME
    BITS as the register-staged kernel (lab switch DVD_GEMM_RING128=0), which runs the same MFMA sequence per accumulator
    and the same epilogue arithmetic."""
    a = rnd(f"r1a{M}{K}", (M, K)).half().cuda()
    w = rnd(f"r1b{N}{K}", (N, K)) * 0.2
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    hi, lo = hi.cuda(), lo.cuda()
    bias, res = rnd("r1bias", (N,)).cuda(), rnd(f"r1res{M}{N}", (M, N)).cuda()
    brow = rnd("r1brow", (M,)).cuda()
    pos, gate = rnd(f"r1pos{N}", (64, N)).cuda(), rnd(f"r1gate{N}", (2, N)).cuda()
    ref = a.cpu().double() @ (hi.cpu().double() + lo.cpu().double() / 2048.0).t()

    def run_all():
        kw = dict(b_lo=lo, small_tiles=True)
        o32 = torch.zeros(M, N, device="cuda")
        ops.gemm_nt(a, hi, out32=o32, **kw)
        g16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        ops.gemm_nt(a, hi, out16=g16, bias=bias, act=1, **kw)
        r32 = res.clone()
        ops.gemm_nt(a, hi, out32=r32, res=r32, bias=bias, **kw)
        pg = torch.zeros(M, N, device="cuda")
        ops.gemm_nt(a, hi, out32=pg, bias=bias, pos=pos, gate=gate, gate_rows=(M + 1) // 2, res=res, **kw)
        t16 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        ops.gemm_nt(a, hi, out16=t16, bias=brow, bias_row=True, small_tiles=True)          # single tensor, row bias
        sw = torch.zeros(N, M, device="cuda")
        ops.gemm_nt(hi, a, out32=sw, a_lo=lo, small_tiles=True)                              # the weight pair on the A side
        torch.cuda.synchronize()
        return [t.cpu() for t in (o32, g16, r32, pg, t16, sw)]

    monkeypatch.setenv("DVD_GEMM_RING256", "0")
    monkeypatch.setenv("DVD_GEMM_RING128", "1")
    new = run_all()
    tol = 2e-4 * K ** 0.5
    assert (new[0].double() - ref).abs().max().item() < tol
    assert (new[1].double() - torch.nn.functional.gelu(ref + bias.cpu().double(), approximate="tanh")).abs().max().item() < 4e-2
    assert (new[2].double() - (ref + bias.cpu().double() + res.cpu().double())).abs().max().item() < tol + 1e-5
    rows = torch.arange(M)
    pgref = (ref + bias.cpu().double() + pos.cpu().double()[rows % 64]) * gate.cpu().double()[rows // ((M + 1) // 2)] + res.cpu().double()
    assert (new[3].double() - pgref).abs().max().item() < 2 * tol + 1e-5
    assert (new[5].double() - ref.t()).abs().max().item() < tol
    monkeypatch.setenv("DVD_GEMM_RING128", "0")
    old = run_all()
    for i, (x, y) in enumerate(zip(new, old)):
        assert torch.equal(x, y), f"output {i}: the ring kernel and the register-staged kernel must give the same bits"
    # gemm_nt_ring256_kernel (128 x 256 tiles, eight waves, six half slabs): taken where N % 256 == 0 (for the A-side call:
    # where the row count M of this test is), the other calls fall through to the kernels above
    monkeypatch.setenv("DVD_GEMM_RING256", "2")       # 2: also where it has fewer tiles than the dispatch rule asks for
    wide = run_all()
    for i, (x, y) in enumerate(zip(wide, old)):
        assert torch.equal(x, y), f"output {i}: the 128 x 256 ring kernel and the register-staged kernel must give the same bits"
