"""Timestep respacing with the reference's names (idf/respace.py)."""
from __future__ import annotations

import torch as th

from .gaussian_diffusion import GaussianDiffusion
from .schedule import space_timesteps  # noqa: F401  (re-export, same contract as idf/respace.py:7-60)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that keeps a subset of the base timesteps (idf/respace.py:63-108).  The sampling loop takes the
    model time of step i from `Tables.model_time(i)`; the single-step entry points (`p_mean_variance`, `ddim_sample`) go
    through `_wrap_model` like the reference's (:95-108), so a model the caller already wrapped is called once, with the
    step INDEX, and maps it itself."""

    def __init__(self, use_timesteps, **kwargs):
        super().__init__(use_timesteps=set(use_timesteps), **kwargs)
        self.use_timesteps = set(use_timesteps)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t          # scaling is done by the wrapped model (:106-108)


class _WrappedModel:
    """idf/respace.py:111-123: `wrapped(x, ts, **kwargs)` takes step indices `ts` [N] (integer tensor), looks up the base
    timestep of each and, with `rescale_timesteps`, scales it to the 0..1000 range in float32 - then calls the model."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps

    def __call__(self, x, ts, **kwargs):
        map_tensor = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = map_tensor[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
