"""Python handle of the HIP denoiser engine (dvd_engine_* in include/dvd_hip.h).

PyTorch provides device memory (one flat weight blob, one workspace) and the stream; the forward
pass itself is entirely inside libdvd_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib, weights
from .lib import ptr, stream_ptr


_SPECS = {}
GRAPH_MAX_GRID = 128


def _is_dev(t) -> bool:
    """Device-residency check of every tensor handed to the library (one place, so the CPU-only multi-rank test can
    stub it together with the compute entry points)."""
    return t.is_cuda


def aligned_empty(nbytes: int, device, align: int = 256) -> torch.Tensor:
    """uint8 buffer of `nbytes` whose address is a multiple of `align` (the engine's requirement for blobs and workspaces)."""
    store = torch.empty(nbytes + align, dtype=torch.uint8, device=device)
    off = (-store.data_ptr()) % align
    return store[off:off + nbytes]


def tensor_specs(grid: int):
    """[(name, dtype 0=f32/1=f16, nelem)] of the engine's weight set for a grid - host-only (no device memory, no
    workspace): what every rank needs to size the flat blob before the one-shot broadcast."""
    if grid not in _SPECS:
        h = C.c_void_p()
        lib.call("dvd_engine_create", grid, 1, 1, C.byref(h))
        try:
            specs = []
            name, dt, ne = C.c_char_p(), C.c_int(), C.c_long()
            for i in range(lib.raw().dvd_engine_tensor_count(h)):
                lib.call("dvd_engine_tensor_info", h, i, C.byref(name), C.byref(dt), C.byref(ne))
                specs.append((name.value.decode(), dt.value, ne.value))
        finally:
            lib.raw().dvd_engine_destroy(h)
        _SPECS[grid] = specs
    return _SPECS[grid]


def blob_layout(grid: int):
    """(name, dtype, nelem, byte offset) of every tensor inside the flat weight blob (256-B aligned), total bytes."""
    off, lay = 0, []
    for name, dt, ne in tensor_specs(grid):
        lay.append((name, dt, ne, off))
        off += (ne * (2 if dt == 1 else 4) + 255) // 256 * 256
    return lay, off


def pack_blob(state_dict, grid: int) -> torch.Tensor:
    """Host blob (uint8) from a reference-named state_dict."""
    packed = weights.pack(state_dict, grid)
    lay, total = blob_layout(grid)
    blob = torch.zeros(total, dtype=torch.uint8)
    for name, dt, ne, off in lay:
        t = packed[name]
        want = torch.float16 if dt == 1 else torch.float32
        if t.dtype != want or t.numel() != ne:
            raise lib.DvdError(f"packer produced {name}: {t.dtype} x{t.numel()}, engine expects {want} x{ne}")
        raw = t.contiguous().view(-1).view(torch.uint8)
        blob[off:off + raw.numel()] = raw
    return blob


class Engine:
    def __init__(self, grid: int, docs: int, n_hyp: int, device="cuda"):
        self.grid, self.docs, self.n_hyp, self.n = grid, docs, n_hyp, docs * n_hyp
        self.device = torch.device(device)
        h = C.c_void_p()
        lib.call("dvd_engine_create", grid, docs, n_hyp, C.byref(h))
        self._h = h
        nbytes = lib.raw().dvd_engine_workspace_bytes(h)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.workspace.data_ptr()) % 256
        self._ws_ptr = self.workspace.data_ptr() + off
        lib.call("dvd_engine_bind_workspace", h, C.c_void_p(self._ws_ptr), nbytes)
        self.specs = tensor_specs(grid)
        self.blob = None
        self._io = None
        # small grids are launch-bound (~110 launches of a few microseconds per evaluation): replay each evaluation as
        # one captured hipGraph.  At the BASELINE grid an evaluation is ~0.4 s of device time and nothing is gained.
        if grid <= GRAPH_MAX_GRID:
            self.set_option("graphs", 1)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.raw().dvd_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------
    def blob_layout(self):
        return blob_layout(self.grid)

    def pack_blob(self, state_dict) -> torch.Tensor:
        return pack_blob(state_dict, self.grid)

    def bind_blob(self, blob_dev: torch.Tensor):
        """Bind a device-resident blob (e.g. after the one-shot RCCL broadcast)."""
        assert blob_dev.device == self.workspace.device and blob_dev.dtype == torch.uint8
        lay, total = self.blob_layout()
        assert blob_dev.numel() >= total and blob_dev.data_ptr() % 256 == 0
        self.blob = blob_dev
        for name, dt, ne, off in lay:
            lib.call("dvd_engine_set_tensor", self._h, name.encode(), C.c_void_p(blob_dev.data_ptr() + off), ne)

    def load_state_dict(self, state_dict):
        self.bind_blob(self.pack_blob(state_dict).to(self.device))

    # ---- compute -------------------------------------------------------------------------
    def prepare(self, y512, mask_cat, mask_y512, line_msk):
        for t, shp in ((y512, (self.docs, 3, 512, 512)), (mask_cat, (self.docs, 1, 512, 512)),
                       (mask_y512, (self.docs, 384, self.grid, self.grid)),
                       (line_msk, (self.docs, 64, self.grid, self.grid))):
            if tuple(t.shape) != shp or t.dtype != torch.float32 or not _is_dev(t) or not t.is_contiguous():
                raise lib.DvdError(f"prepare: expected contiguous f32 device tensor of shape {shp}, got {tuple(t.shape)}")
        lib.call("dvd_engine_prepare_docs", self._h, ptr(y512), ptr(mask_cat), ptr(mask_y512), ptr(line_msk),
                 stream_ptr())

    def feat_nchw(self):
        out = torch.empty(self.docs, 256, self.grid, self.grid, dtype=torch.float32, device=self.device)
        lib.call("dvd_engine_feat_nchw", self._h, ptr(out), stream_ptr())
        return out

    def denoise(self, x_t, t_embed: float, feat_mode: int, init_flow, out=None, init_feat=None, dither_step: int = 0):
        """dither_step: the phase of the weight dithering for THIS evaluation (the sampler passes its loop counter,
        0 at the first step).  It is always set explicitly - the handle keeps no running counter - so an evaluation is a
        pure function of its arguments: equal inputs give equal bits whatever the engine ran before."""
        shp = (self.n, 2, self.grid, self.grid)
        for t in (x_t, init_flow):
            if tuple(t.shape) != shp or t.dtype != torch.float32 or not _is_dev(t) or not t.is_contiguous():
                raise lib.DvdError(f"denoise: expected contiguous f32 device tensor of shape {shp}")
        if out is None:
            out = torch.empty(shp, dtype=torch.float32, device=self.device)
        if feat_mode == 3:
            fs = (self.n, 256, self.grid, self.grid)
            if init_feat is None or tuple(init_feat.shape) != fs or init_feat.dtype != torch.float32 \
                    or not _is_dev(init_feat) or not init_feat.is_contiguous():
                raise lib.DvdError(f"denoise: feat_mode 3 needs a contiguous f32 device init_feat of shape {fs}")
        self.set_option("dither_step", int(dither_step))
        lib.call("dvd_engine_denoise_step", self._h, ptr(x_t), C.c_float(t_embed), feat_mode, ptr(init_flow),
                 ptr(init_feat if feat_mode == 3 else None), ptr(out), stream_ptr())
        return out

    def io_buffers(self):
        """Persistent I/O of the sampling loop (x_t and x0 ping-pong pairs, the first step's init_flow)."""
        if self._io is None:
            shp = (self.n, 2, self.grid, self.grid)
            mk = lambda: torch.empty(shp, dtype=torch.float32, device=self.device)  # noqa: E731
            self._io = {"img": [mk(), mk()], "x0": [mk(), mk()], "flow0": mk()}
        return self._io

    def set_option(self, name: str, value: int):
        lib.call("dvd_engine_set_option", self._h, name.encode(), int(value))

    def profile(self, enable: bool):
        lib.call("dvd_engine_profile", self._h, int(enable))

    def profile_read(self):
        n, ms = C.c_int(), C.c_double()
        lib.call("dvd_engine_profile_read", self._h, C.byref(n), C.byref(ms))
        return n.value, ms.value

    def debug_stop(self, stage: int):
        lib.call("dvd_engine_debug_stop", self._h, stage)

    def debug(self, name: str, dtype, shape):
        p, nb = C.c_void_p(), C.c_long()
        lib.call("dvd_engine_debug_buffer", self._h, name.encode(), C.byref(p), C.byref(nb))
        n = int(np.prod(shape))
        esz = torch.empty(0, dtype=dtype).element_size()
        assert n * esz <= nb.value, (name, n * esz, nb.value)
        off = p.value - self.workspace.data_ptr()
        return self.workspace[off:off + n * esz].view(dtype).view(*shape)
