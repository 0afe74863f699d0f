"""The denoiser as an nn.Module with the reference's parameter names (idf/cross_model.py:361-460), so that
`model.cpu().load_state_dict(ckpt, strict=False)`, `.to(dev)`, `.eval()`, `.parameters()` behave as in
val_TDiff.run (:79-85) - but whose forward pass is the HIP engine (no ATen compute, no CPU fallback)."""
from __future__ import annotations

import torch
from torch import nn

from . import schedule, synth
from .engine import Engine


def _require_gpu(dev):
    if dev.type != "cuda":
        raise RuntimeError("DvdDenoiser runs on the HIP engine only: move the model to a GPU (no CPU fallback)")


class DvdDenoiser(nn.Module):
    def __init__(self, input_size=64, in_channels=2, tv=True, depth=12):
        super().__init__()
        if in_channels != 2 or not tv:
            raise NotImplementedError("only the live configuration (in_channels=2, tv=True) is implemented")
        self.input_size, self.tv, self.depth = input_size, tv, depth
        self._names = {}
        spec = synth.state_dict_spec(input_size, range(depth))
        for key, (shape, kind) in spec.items():
            flat = key.replace(".", "__")
            self._names[key] = flat
            if kind in ("pos", "dec_h", "dec_w"):
                val = torch.from_numpy(synth.synth_state_dict(input_size, 0, keys={key})[key])
                if kind == "pos":
                    self.register_parameter(flat, nn.Parameter(val, requires_grad=False))
                else:
                    self.register_buffer(flat, val)
            elif kind.startswith("bn_") and kind in ("bn_m", "bn_v", "bn_n"):
                init = torch.ones(shape) if kind == "bn_v" else torch.zeros(shape, dtype=torch.long if kind == "bn_n" else torch.float32)
                self.register_buffer(flat, init)
            else:
                self.register_parameter(flat, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._engines = {}
        self._blob = None
        self._blob_version = -1
        self._version = 0

    # ---- checkpoint-compatible state dict (keys with dots, as in model1852000.pt) --------------------
    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        # checkpoint key names, in the reference module's own order (parameters and buffers interleaved)
        return type(sd)((key, sd[flat]) for key, flat in self._names.items() if flat in sd)

    def load_state_dict(self, state_dict, strict=True, **kw):
        mapped = {self._names[k]: v for k, v in state_dict.items() if k in self._names}
        unexpected = [k for k in state_dict if k not in self._names]
        res = super().load_state_dict(mapped, strict=False, **kw)
        missing = [k for k, f in self._names.items() if f in res.missing_keys]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        self._version += 1
        return res

    # ---- engine plumbing ------------------------------------------------------------------------------
    # ---- flat weight blob: dist_util.materialize_blobs([...]) packs on rank 0, ONE broadcast, every rank binds ----
    def blob_bytes(self) -> int:
        from .engine import blob_layout
        return blob_layout(self.input_size)[1]

    def pack_into(self, view_u8: torch.Tensor):
        from .engine import pack_blob
        view_u8.copy_(pack_blob({k: v.detach().cpu() for k, v in self.state_dict().items()}, self.input_size))

    def bind_blob(self, view_u8: torch.Tensor):
        self._blob, self._blob_version = view_u8, self._version

    def materialize_blob(self, src: int = 0):
        """Pack the current parameters into the engine's flat weight blob on rank `src` and hand it to every rank
        with ONE flat broadcast (RCCL over xGMI) - the sampling path's only collective.  COLLECTIVE: every rank of
        the process group must call it, unconditionally and at the same point; `engine()` itself never communicates,
        so a rank with an empty document shard cannot leave the others waiting in a broadcast.  (val_TDiff.run
        broadcasts this model together with the pre-stage nets: dist_util.materialize_blobs.)"""
        from . import dist_util
        _require_gpu(self.device)
        dist_util.materialize_blobs([self], src=src)
        return self._blob

    def engine(self, grid: int, docs: int, n_hyp: int) -> Engine:
        """The engine for (grid, docs, hypotheses) bound to the current weights.  Issues NO collective."""
        dev = self.device
        _require_gpu(dev)
        key = (grid, docs, n_hyp, dev.index)
        eng = self._engines.get(key)
        if eng is None:
            if grid != self.input_size:
                raise ValueError(f"model built for grid {self.input_size}, asked to sample grid {grid}")
            eng = self._engines[key] = Engine(grid, docs, n_hyp, device=dev)
            eng._bound_version = -1
        if eng._bound_version != self._version:
            if self._blob is None or self._blob_version != self._version:
                from . import dist_util
                if dist_util.world_size() > 1:
                    raise RuntimeError("the weights changed (or were never packed) in a multi-rank run: call "
                                       "model.materialize_blob() on EVERY rank before sampling (it is the one "
                                       "collective of the path; engine() must not issue it lazily)")
                self.materialize_blob()
            eng.bind_blob(self._blob)
            eng._bound_version = self._version
        return eng

    def forward(self, x, t, y=None, y512=None, mask_y512=None, init_flow=None, local_corr=None, trg_feat=None,
                src_feat=None, src_64=None, mask_x=None, tv=None, source_0=None, tmode=None, line_msk=None,
                mask_cat=None, init_feat=None, iter=False, mode=None, dither_step=None):
        """One denoiser evaluation with the reference's keyword surface (idf/cross_model.py:568-570):
        returns (x0_pred [N,2,G,G], feat [N,256,G,G]).  Each sample is treated as its own document.
        Like the reference's forward it is a PURE function of its arguments: on grids whose GEMM weights are dithered the
        phase is `dither_step` when given (diffusion.ddim_sample / p_mean_variance pass the loop's S-1-i, so a hand-rolled
        chain reproduces ddim_sample_loop bit for bit), else schedule.dither_phase(t) - never the engine's history."""
        if src_feat is not None or not (tv is True) or not iter:
            raise NotImplementedError("only the src_feat=None, tv=True, iter=True path is live (admin/local.py:27-29)")
        n, _, g, _ = x.shape
        t0 = float(t[0])
        if not bool((t.float() == t0).all()):
            raise ValueError("the timestep must be identical across the batch (the reference's rule is batch-global)")
        t_embed = schedule.embedded_time(t0) if mode is None else t0
        eng = self.engine(g, n, 1)
        f32 = lambda a: a.to(x.device, torch.float32).contiguous()  # noqa: E731
        eng.prepare(f32(y512), f32(mask_cat), f32(mask_y512), f32(line_msk))
        flow = f32(init_flow) if init_flow is not None else torch.zeros_like(x)
        if t0 > 600 or (n > 1 and t0 == 2.0):
            fmode, feat_in = 1, None
        elif init_feat is None:
            fmode, feat_in = 0, None
        else:
            fmode, feat_in = 3, f32(init_feat)
        phase = schedule.dither_phase(t0) if dither_step is None else int(dither_step)
        x0 = eng.denoise(f32(x), t_embed, fmode, flow, init_feat=feat_in, dither_step=phase)
        if init_flow is None:
            pass  # the reference only adds init_flow when it is given (:645-646); zeros were used above
        return x0, eng.feat_nchw()

    @property
    def device(self):
        return next(self.parameters()).device


def DiT_S_2(**kwargs):
    return DvdDenoiser(depth=12, **kwargs)


DiT_models2 = {"DiT-S/2": DiT_S_2}
