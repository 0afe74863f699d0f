"""Timestep respacing with the reference's names (idf/respace.py)."""
from __future__ import annotations

from .gaussian_diffusion import GaussianDiffusion
from .schedule import space_timesteps  # noqa: F401  (re-export, same contract as idf/respace.py:7-60)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that keeps a subset of the base timesteps (idf/respace.py:63-108).  The
    `_WrappedModel` time mapping (:111-123) is `Tables.model_time`."""

    def __init__(self, use_timesteps, **kwargs):
        super().__init__(use_timesteps=set(use_timesteps), **kwargs)
        self.use_timesteps = set(use_timesteps)
