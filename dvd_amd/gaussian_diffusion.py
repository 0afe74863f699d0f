"""Host-side mirror of the reference's diffusion object for the sampling path
(idf/gaussian_diffusion.py): same constructor keywords, attribute names and
`ddim_sample_loop(...) -> (sample, final)` contract, with the loop body running on the HIP engine.

Only sampling is mirrored (training losses, VLB terms and the UNet paths are out of scope)."""
from __future__ import annotations

import enum

import numpy as np
import torch as th

from . import ops, sampler, schedule


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    return schedule.named_betas(schedule_name, num_diffusion_timesteps)


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False,
                 use_timesteps=None):
        if model_mean_type != ModelMeanType.START_X:
            raise NotImplementedError("the DvD sampling path predicts x_0 (predict_xstart=True, admin/local.py)")
        if model_var_type not in (ModelVarType.FIXED_LARGE, ModelVarType.FIXED_SMALL):
            raise NotImplementedError("learned variances are not on the DvD sampling path")
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        self.rescale_timesteps = rescale_timesteps
        self.tables = schedule.Tables(np.asarray(betas, dtype=np.float64), use_timesteps, rescale_timesteps)
        for name in ("betas", "num_timesteps", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                     "posterior_mean_coef1", "posterior_mean_coef2", "timestep_map"):
            setattr(self, name, getattr(self.tables, name))
        self.original_num_steps = self.tables.original_num_steps

    # ------------------------------------------------------------------------------------------
    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                         device=None, progress=False, eta=0.0, sampling_kwargs=None, logger=None, n_batch=1,
                         time_variant=False, pyramid=None, sampler_kind="ddim"):
        """Same call as idf/gaussian_diffusion.py:494-534.  `shape` is (B, 2, G, G): B = 1 in the reference
        (one document per call); B > 1 samples B documents x n_batch hypotheses in one engine batch
        (model_kwargs tensors then carry B rows).  Returns (sample [B,2,G,G], final dict)."""
        if clip_denoised or denoised_fn is not None:
            raise NotImplementedError("clip_denoised / denoised_fn are off on the DvD path (admin/local.py:65)")
        if not time_variant or not model_kwargs.get("iter", True):
            raise NotImplementedError("only the live time_variant=True / iter=True configuration is mirrored")
        B, C, G, G2 = shape
        assert C == 2 and G == G2
        dev = device or next(model.parameters()).device
        eng = model.engine(G, B, n_batch)
        kw = model_kwargs
        eng.prepare(*[kw[k].to(dev, th.float32).contiguous() for k in ("y512", "mask_cat", "mask_y512", "line_msk")])
        if noise is not None:
            x_T = noise.to(dev, th.float32).contiguous()
            assert tuple(x_T.shape) == (B * n_batch, 2, G, G)
        else:
            _ = th.randn(*shape, device=dev)                     # the reference draws and discards this (:562)
            x_T = th.randn((B * n_batch, 2, G, G), device=dev)    # (:569)
        noise_fn = None
        if sampler_kind == "ddpm" or eta != 0.0:
            noise_fn = lambda i: th.randn((B * n_batch, 2, G, G), device=dev)  # noqa: E731
        sample = sampler.sample(eng, self.tables, x_T, sampler=sampler_kind, eta=eta, noise_fn=noise_fn)
        final = {"sample": sample, "pred_xstart": sample, "feat_dict": eng.feat_nchw()}
        return sample, final

    def p_sample_loop(self, model, shape, **kw):
        """DDPM ancestral sampling (absent from the reference): BASELINE config 4."""
        return self.ddim_sample_loop(model, shape, sampler_kind="ddpm", **kw)

    # reference-compatible single step (idf/gaussian_diffusion.py:445-491) on an explicit x0 prediction
    def ddim_step(self, x_t, x0, i, eta=0.0, noise=None):
        return ops.sched_step(self.tables.ddim_coef(i, eta), x_t, x0, noise)
