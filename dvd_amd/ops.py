"""Thin torch-tensor wrappers over the op-level entry points of libdvd_hip.so.

PyTorch supplies device memory and the current HIP stream; every computation happens in the
HIP library.  These wrappers mirror the reference callables they replace:
  grid_sample  <- register_model2(size,'bilinear')([img, grid])   datasets/utils/warping.py:14-23
  unwarp_*     <- evaluation.py:301-306 + visualization_utils.py:75-77
  sched_step   <- GaussianDiffusion.ddim_sample arithmetic         idf/gaussian_diffusion.py:470-489
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib
from .lib import ptr, stream_ptr


def _chk(t: torch.Tensor, dtype, name):
    if not t.is_cuda:
        raise lib.DvdError(f"{name}: expected a device tensor (the DvD engine has no CPU path)")
    if t.dtype != dtype or not t.is_contiguous():
        raise lib.DvdError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")


def grid_sample(src: torch.Tensor, grid_nchw: torch.Tensor, src_batch_div: int = 1) -> torch.Tensor:
    _chk(src, torch.float32, "src")
    _chk(grid_nchw, torch.float32, "grid")
    n, two, h, w = grid_nchw.shape
    ns, c, hin, win = src.shape
    assert two == 2 and ns * src_batch_div == n
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=src.device)
    lib.call("dvd_grid_sample_bilinear_zeros_ac", ptr(src), ptr(grid_nchw), ptr(out), n, c, hin, win, h, w,
             src_batch_div, stream_ptr())
    return out


def unwarp_grid(flow: torch.Tensor, h: int, w: int, scale: float = 0.987) -> torch.Tensor:
    _chk(flow, torch.float32, "flow")
    g = flow.shape[-1]
    out = torch.empty((1, 2, h, w), dtype=torch.float32, device=flow.device)
    lib.call("dvd_unwarp_grid", ptr(flow), g, ptr(out), h, w, C.c_float(scale), stream_ptr())
    return out


def unwarp_f32(flow: torch.Tensor, src_chw: torch.Tensor, scale: float = 0.987) -> torch.Tensor:
    """flow [1,2,G,G] (or [2,G,G]); src [1,3,H,W] f32 0..255 -> [H,W,3] f32."""
    _chk(flow, torch.float32, "flow")
    _chk(src_chw, torch.float32, "src")
    h, w = src_chw.shape[-2:]
    out = torch.empty((h, w, 3), dtype=torch.float32, device=src_chw.device)
    lib.call("dvd_unwarp_f32", ptr(flow), flow.shape[-1], ptr(src_chw), ptr(out), h, w, C.c_float(scale), stream_ptr())
    return out


def unwarp_u8(flow: torch.Tensor, src_hwc: torch.Tensor, scale: float = 0.987) -> torch.Tensor:
    """flow [1,2,G,G]; src [H,W,3] uint8 -> [H,W,3] uint8 (truncated like numpy astype)."""
    _chk(flow, torch.float32, "flow")
    _chk(src_hwc, torch.uint8, "src")
    h, w = src_hwc.shape[:2]
    out = torch.empty_like(src_hwc)
    lib.call("dvd_unwarp_u8", ptr(flow), flow.shape[-1], ptr(src_hwc), ptr(out), h, w, C.c_float(scale), stream_ptr())
    return out


def unwarp_u8_batch(flow: torch.Tensor, src_nhwc: torch.Tensor, scale: float = 0.987) -> torch.Tensor:
    """flow [B,2,G,G]; src [B,H,W,3] uint8 -> [B,H,W,3] uint8: the batch's documents in ONE launch."""
    _chk(flow, torch.float32, "flow")
    _chk(src_nhwc, torch.uint8, "src")
    b, h, w, three = src_nhwc.shape
    assert three == 3 and flow.shape[0] == b and flow.shape[1] == 2
    out = torch.empty_like(src_nhwc)
    lib.call("dvd_unwarp_u8_batch", ptr(flow), flow.shape[-1], ptr(src_nhwc), ptr(out), b, h, w, C.c_float(scale),
             stream_ptr())
    return out


def unwarp_f32_batch(flow: torch.Tensor, src_nchw: torch.Tensor, scale: float = 0.987) -> torch.Tensor:
    """flow [B,2,G,G]; src [B,3,H,W] f32 0..255 -> [B,H,W,3] f32."""
    _chk(flow, torch.float32, "flow")
    _chk(src_nchw, torch.float32, "src")
    b, three, h, w = src_nchw.shape
    assert three == 3 and flow.shape[0] == b and flow.shape[1] == 2
    out = torch.empty((b, h, w, 3), dtype=torch.float32, device=src_nchw.device)
    lib.call("dvd_unwarp_f32_batch", ptr(flow), flow.shape[-1], ptr(src_nchw), ptr(out), b, h, w, C.c_float(scale),
             stream_ptr())
    return out


def ingest_u8(img_hwc: torch.Tensor, swap_rb: bool = False, out_size: int = 512, want_rgb: bool = False):
    """Decoded image [H,W,3] uint8 on the device -> source_image [3,out,out] f32 in 0..1 (cv2.resize INTER_LINEAR / 255,
    doc_benchmark.py:84-88) and, if asked, the full-resolution RGB image (doc_benchmark.py:76)."""
    _chk(img_hwc, torch.uint8, "img")
    h, w, three = img_hwc.shape
    assert three == 3
    y = torch.empty((3, out_size, out_size), dtype=torch.float32, device=img_hwc.device)
    rgb_out = torch.empty_like(img_hwc) if (want_rgb and swap_rb) else None
    scratch = torch.empty(lib.raw().dvd_ingest_scratch_bytes(out_size), dtype=torch.uint8, device=img_hwc.device)
    lib.call("dvd_ingest_u8", ptr(img_hwc), h, w, int(swap_rb), ptr(y), out_size, ptr(rgb_out), ptr(scratch), stream_ptr())
    return (y, rgb_out if swap_rb else img_hwc) if want_rgb else y


def sched_step(coef: lib.SchedCoef, x_t, x0, noise=None, want_grid=False, out=None):
    _chk(x_t, torch.float32, "x_t")
    _chk(x0, torch.float32, "x0")
    n, _, g, _ = x_t.shape
    if out is None:
        out = torch.empty_like(x_t)
    else:
        _chk(out, torch.float32, "out")
        assert out.shape == x_t.shape and out.data_ptr() != x_t.data_ptr()
    ngrid = torch.empty_like(x_t) if want_grid else None
    if noise is not None:
        _chk(noise, torch.float32, "noise")
    lib.call("dvd_sched_step", C.byref(coef), ptr(x_t), ptr(x0), ptr(noise), ptr(out), ptr(ngrid), n, g, stream_ptr())
    return (out, ngrid) if want_grid else out


def hyp_mean_clamp(x0: torch.Tensor, n_hyp: int) -> torch.Tensor:
    _chk(x0, torch.float32, "x0")
    n, _, g, _ = x0.shape
    docs = n // n_hyp
    out = torch.empty((docs, 2, g, g), dtype=torch.float32, device=x0.device)
    lib.call("dvd_hyp_mean_clamp", ptr(x0), ptr(out), docs, n_hyp, g, stream_ptr())
    return out


def dither_f16(hi: torch.Tensor, lo: torch.Tensor, step: int, elem0: int = 0, out=None) -> torch.Tensor:
    """W = hi + lo (both f16) re-rounded to ONE f16 with the step-dependent sub-ulp offset of dvd_dither_f16."""
    _chk(hi, torch.float16, "hi")
    _chk(lo, torch.float16, "lo")
    if out is None:
        out = torch.empty_like(hi)
    lib.call("dvd_dither_f16", ptr(hi), ptr(lo), ptr(out), hi.numel(), C.c_uint(elem0 & 0xFFFFFFFF),
             C.c_uint(step & 0xFFFFFFFF), stream_ptr())
    return out


def selftest_mfma(a16, b16, vt16):
    out = torch.empty(3072, dtype=torch.float32, device=a16.device)
    lib.call("dvd_selftest_mfma", ptr(a16), ptr(b16), ptr(vt16), ptr(out), stream_ptr())
    return out


def _addr(t):
    return None if t is None else t.data_ptr()


def gemm_nt(a, b, *, out32=None, out16=None, bias=None, bias_row=False, act=0, pos=None, gate=None,
            gate_rows=0, res=None, batch=1, strides=None, M=None, N=None, K=None, lda=None, ldb=None, b_lo=None,
            a_lo=None, lo_scale=2.0 ** -11, small_tiles=False):
    """C = epi(A . B^T).  a [M,K] / b [N,K] f16 or f32 device tensors (2-D views may be strided in rows).
    strides: dict of batch strides in elements (A,B,C32,C16,bias,gate,res)."""
    assert a.dtype == b.dtype and a.dtype in (torch.float16, torch.float32)
    d = lib.GemmDesc()
    d.dtype = 1 if a.dtype == torch.float32 else 0
    d.M = M if M is not None else a.shape[-2]
    d.N = N if N is not None else b.shape[-2]
    d.K = K if K is not None else a.shape[-1]
    d.batch = batch
    st = strides or {}
    d.A, d.lda, d.strideA = _addr(a), (lda if lda is not None else a.stride(-2)), st.get("A", 0)
    d.B, d.ldb, d.strideB = _addr(b), (ldb if ldb is not None else b.stride(-2)), st.get("B", 0)
    if b_lo is not None:
        d.B_lo, d.lo_scale = _addr(b_lo), lo_scale
    if a_lo is not None:
        d.A_lo, d.lo_scale = _addr(a_lo), lo_scale
    if out32 is not None:
        d.C32, d.ldc, d.strideC32 = _addr(out32), out32.stride(-2), st.get("C32", 0)
    if out16 is not None:
        d.C16, d.ldc16, d.strideC16 = _addr(out16), out16.stride(-2), st.get("C16", 0)
    if bias is not None:
        d.bias, d.bias_row, d.strideBias = _addr(bias), int(bias_row), st.get("bias", 0)
    d.act = act
    if pos is not None:
        d.pos, d.ldpos, d.pos_rows = _addr(pos), pos.stride(-2), pos.shape[-2]
    if gate is not None:
        d.gate, d.ldgate, d.gate_rows, d.strideGate = _addr(gate), gate.stride(-2), gate_rows, st.get("gate", 0)
    if res is not None:
        d.res, d.ldres, d.strideRes = _addr(res), res.stride(-2), st.get("res", 0)
    d.small_tiles = int(small_tiles)
    lib.call("dvd_gemm_nt", C.byref(d), stream_ptr())


def flash_attn(q, k, vt, out, heads, head_dim, scale, kv_batch_div=1):
    """q [B,Tq,*] k [Bkv,Tk,*] (row-strided views ok), vt [Bkv, heads*hd, Tk], out [B,Tq,heads*hd]; all f16."""
    d = lib.AttnDesc()
    d.head_dim, d.heads, d.batch, d.tq, d.tk = head_dim, heads, q.shape[0], q.shape[1], k.shape[1]
    d.kv_batch_div = kv_batch_div
    d.Q, d.ldq, d.strideQ = q.data_ptr(), q.stride(1), q.stride(0)
    d.K, d.ldk, d.strideK = k.data_ptr(), k.stride(1), k.stride(0)
    d.Vt, d.ldvt, d.strideVt = vt.data_ptr(), vt.stride(1), vt.stride(0)
    d.O, d.ldo, d.strideO = out.data_ptr(), out.stride(1), out.stride(0)
    d.scale = scale
    lib.call("dvd_flash_attn", C.byref(d), stream_ptr())
    return out


# ---- token-side kernels (thin wrappers used by the parity tests; the engine calls them from C++) ----------------
def layernorm_rows(x, c, gamma=None, beta=None, shift=None, scale=None, mod_rows=1, eps=1e-6):
    rows = x.shape[0]
    out = torch.empty(rows, c, dtype=torch.float16, device=x.device)
    lib.call("dvd_layernorm_rows", ptr(x), x.stride(0), 0, ptr(out), c, 0, 1, rows, c, ptr(gamma), ptr(beta), ptr(shift),
             ptr(scale), (shift.stride(0) if shift is not None and shift.dim() > 1 else 0), mod_rows, C.c_float(eps),
             stream_ptr())
    return out


def dwconv3x3(x16, w9c, b, n, side):
    out = torch.empty_like(x16)
    lib.call("dvd_dwconv3x3", ptr(x16), ptr(out), ptr(w9c), ptr(b), n, side, x16.shape[-1], stream_ptr())
    return out


def embed_obs_ln(x, w, b, pos):
    n, _, g, _ = x.shape
    T = (g // 2) ** 2
    tok = torch.empty(n * T, 384, dtype=torch.float32, device=x.device)
    ln = torch.empty(n * T, 384, dtype=torch.float16, device=x.device)
    lib.call("dvd_embed_obs_ln", ptr(x), ptr(w), ptr(b), ptr(pos), ptr(tok), ptr(ln), n, g, stream_ptr())
    return tok, ln


def small_linear(x, w, b, act_in=0, act_out=0, kmod=None):
    m = x.shape[0]
    n, k = w.shape
    y = torch.empty(m, n, dtype=torch.float32, device=x.device)
    lib.call("dvd_small_linear", ptr(x), x.stride(0), ptr(w), ptr(b), ptr(y), n, m, k, n, kmod or k, act_in, act_out,
             stream_ptr())
    return y


def final_tokens(z, gamma, beta, shift, scale, w8, b8, init_flow, n, g):
    x0 = torch.empty(n, 2, g, g, dtype=torch.float32, device=z.device)
    tok8 = torch.empty(z.shape[0], 8, dtype=torch.float32, device=z.device)
    lib.call("dvd_final_tokens", ptr(z), ptr(gamma), ptr(beta), ptr(shift), ptr(scale), 0, z.shape[0], ptr(w8), ptr(b8),
             ptr(init_flow), ptr(x0), ptr(tok8), n, g, stream_ptr())
    return x0, tok8


def posenc(z, hs, ws, htab, wtab, n, side):
    c = z.shape[-1]
    chunks = 8
    part = torch.empty(n * chunks * c, dtype=torch.float32, device=z.device)
    pooled = torch.empty(n, c, dtype=torch.float32, device=z.device)
    lib.call("dvd_colmean", ptr(z), ptr(part), ptr(pooled), n, side * side, c, chunks, stream_ptr())
    lib.call("dvd_posenc_add", ptr(z), ptr(hs), ptr(ws), ptr(htab), ptr(wtab), n, side, c, stream_ptr())
    return pooled


def build_r_rows(feat_nhwc, flow, n_hyp, mode, init_feat=None):
    n, _, g, _ = flow.shape
    T = (g // 2) ** 2
    out = torch.empty(n * T, 1088, dtype=torch.float16, device=flow.device)
    lib.call("dvd_build_r_rows", ptr(feat_nhwc), ptr(init_feat), ptr(flow), ptr(out), 1088, n, g, n_hyp, mode, stream_ptr())
    return out


def conv3x3_relu_nhwc(x_nhwc, w_packed, bias, cin, cout, h, w):
    """One pyramid layer the way the engine runs it: im2col + exact-f32 GEMM (+bias, ReLU)."""
    kp = w_packed.shape[1]
    col = torch.empty(h * w, kp, dtype=torch.float32, device=x_nhwc.device)
    lib.call("dvd_im2col3x3", ptr(x_nhwc), 1, w * cin, cin, ptr(col), kp, cin, h, w, stream_ptr())
    out = torch.empty(h * w, cout, dtype=torch.float32, device=x_nhwc.device)
    gemm_nt(col, w_packed, out32=out, bias=bias, act=2)
    return out


def conv3x3_relu_nhwc_implicit(x_nhwc, w_packed, bias, cin, cout, h, w):
    """The same layer without the im2col matrix (cin % 16 == 0): what the engine runs for every pyramid layer but the first."""
    out = torch.empty(h * w, cout, dtype=torch.float32, device=x_nhwc.device)
    lib.call("dvd_conv3x3_nhwc", ptr(x_nhwc), cin, ptr(w_packed), w_packed.shape[1], ptr(bias), ptr(out), cout, h, w, 1,
             stream_ptr())
    return out


def maxpool2_nhwc(x, c, h, w):
    out = torch.empty((h // 2) * (w // 2), c, dtype=torch.float32, device=x.device)
    lib.call("dvd_maxpool2_nhwc", ptr(x), ptr(out), c, h, w, stream_ptr())
    return out


def resize_bilinear_nhwc(x, c, hin, win, hout, wout):
    out = torch.empty(hout * wout, c, dtype=torch.float32, device=x.device)
    lib.call("dvd_resize_bilinear_nhwc", ptr(x), ptr(out), c, hin, win, hout, wout, stream_ptr())
    return out
