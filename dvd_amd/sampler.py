"""The sampling loop on the HIP engine - host-side mirror of
GaussianDiffusion.ddim_sample_loop_progressive_only_mean (idf/gaussian_diffusion.py:537-644) for
B documents x H hypotheses at once, plus the DDPM ancestral variant (BASELINE config 4)."""
from __future__ import annotations

import torch

from . import ops, schedule
from .engine import Engine


def feat_mode_for(t_model: float, n: int, first_step: bool) -> int:
    """Which init_feat the denoiser sees (idf/cross_model.py:596-601; idf/gaussian_diffusion.py:618-624):
    1 = the pyramid features themselves, 2 = features warped by the previous x0, 0 = zeros."""
    if t_model > 600 or (n > 1 and t_model == 2.0):
        return 1
    return 0 if first_step else 2


def sample(engine: Engine, tables: schedule.Tables, x_T: torch.Tensor, sampler: str = "ddim", eta: float = 0.0,
           noise_fn=None, mean_hyp: bool = True, trace=None, last_step: int = 0, init_flow=None, init_feat=None,
           t_override: bool = True):
    """x_T [docs*H, 2, G, G] on the engine's device (sample index = doc*H + h).  The engine must have been
    prepared for its documents.  noise_fn(step) -> [N,2,G,G] supplies the per-step noise (DDPM, or DDIM with
    eta > 0).  Returns [docs,2,G,G] (hypothesis mean + clamp, :639-640) or the clamped per-sample maps.
    last_step > 0 stops the roll-out early (the training-time variant starts at S-1 and ends at `timestep + 1`,
    idf/gaussian_diffusion.py:720).  init_flow [N,2,G,G] / init_feat [N,256,G,G] are the caller's model_kwargs of
    the FIRST step (:578,:729: later steps use the previous x0 and the features warped by it; the model itself
    replaces init_feat by the pyramid features while t_model > 600, idf/cross_model.py:597-598).  t_override=False is
    the denoiser's `mode != None` (training) call: the raw model time is embedded, no 2/1 override (:575-580)."""
    n = engine.n
    S = tables.num_timesteps
    # The loop's I/O lives in buffers owned by the engine object and reused by every roll-out: x_t and x0 ping-pong
    # between two buffers each, so a denoiser evaluation sees one of a handful of fixed (x_t, init_flow, x0) address
    # triples - which is what lets the engine replay it as a captured hipGraph (dvd_engine_set_option "graphs").
    io = engine.io_buffers()
    if tuple(x_T.shape) != tuple(io["img"][0].shape):
        raise ValueError(f"x_T must be {tuple(io['img'][0].shape)}, got {tuple(x_T.shape)}")
    img_bufs, x0_bufs = io["img"], io["x0"]
    img = img_bufs[0]
    img.copy_(x_T)
    first_flow = io["flow0"]
    if init_flow is None:
        first_flow.zero_()
    else:
        if tuple(init_flow.shape) != tuple(img.shape):
            raise ValueError(f"init_flow must be {tuple(img.shape)}, got {tuple(init_flow.shape)}")
        first_flow.copy_(init_flow)
    x0 = None
    if not 0 <= last_step < S:
        raise ValueError(f"last_step {last_step} outside [0, {S})")
    for k, i in enumerate(range(S - 1, last_step - 1, -1)):
        t_model = tables.model_time(i)
        first = i == S - 1
        flow = first_flow if first else x0
        out = x0_bufs[k & 1]
        mode = feat_mode_for(t_model, n, first)
        feat_in = None
        if first and mode == 0 and init_feat is not None:      # a first step at t_model <= 600 sees the caller's init_feat
            mode, feat_in = 3, init_feat.to(img.device, torch.float32).contiguous()
        t_embed = schedule.embedded_time(t_model) if t_override else float(t_model)
        x0 = engine.denoise(img, t_embed, mode, flow, out=out, init_feat=feat_in, dither_step=k)
        if trace is not None:
            trace.append(x0.clone())
        if sampler == "ddim":
            coef = tables.ddim_coef(i, eta)
        elif sampler == "ddpm":
            coef = tables.ddpm_coef(i)
        else:
            raise ValueError(f"unknown sampler {sampler!r}")
        noise = noise_fn(i) if (coef.sigma != 0.0 and noise_fn is not None) else None
        if coef.sigma != 0.0 and noise is None:
            raise ValueError("this step needs noise: pass noise_fn")
        img = ops.sched_step(coef, img, x0, noise, out=img_bufs[(k + 1) & 1])
    if mean_hyp:
        return ops.hyp_mean_clamp(x0, engine.n_hyp)
    return torch.clamp(x0, -1, 1)
