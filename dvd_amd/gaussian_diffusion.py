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


_DITHER_KW_CACHE = None          # WeakKeyDictionary, built on first use (callables that cannot be weakly referenced are not cached)


def _accepts_dither_step(model) -> bool:
    """True for this package's denoiser behind any chain of pass-through wrappers (each hop must forward **kwargs or name
    the keyword); False for a callable with the reference's keyword surface.  Cached per model object."""
    global _DITHER_KW_CACHE
    import inspect
    import weakref
    if _DITHER_KW_CACHE is None:
        _DITHER_KW_CACHE = weakref.WeakKeyDictionary()
    try:
        hit = _DITHER_KW_CACHE.get(model)
    except TypeError:
        hit = None
    if hit is not None:
        return hit

    def takes(obj):
        fwd = getattr(obj, "forward", obj)
        try:
            params = inspect.signature(fwd).parameters
        except (TypeError, ValueError):
            return False, False
        return "dither_step" in params, any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())

    obj, ok = model, False
    for _ in range(8):
        named, var_kw = takes(obj)
        if named:
            ok = True
            break
        nxt = next((getattr(obj, a) for a in ("model", "module", "__wrapped__")
                    if getattr(obj, a, None) is not None and callable(getattr(obj, a))), None)
        if not var_kw or nxt is None or nxt is obj:      # a hop that neither names the keyword nor forwards **kwargs
            break
        obj = nxt
    try:
        _DITHER_KW_CACHE[model] = ok
    except TypeError:
        pass
    return ok


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
        sample = sampler.sample(eng, self.tables, x_T, sampler=sampler_kind, eta=eta, noise_fn=noise_fn,
                                **self._first_step_kwargs(kw, n_batch))
        final = {"sample": sample, "pred_xstart": sample, "feat_dict": eng.feat_nchw()}
        return sample, final

    def _first_step_kwargs(self, kw, n_batch):
        """model_kwargs['init_flow'] / ['init_feat'] reach the denoiser at the first step only, tiled over the
        hypotheses like every model_kwarg (idf/gaussian_diffusion.py:574,578; sample index = doc * n_batch + h).
        init_feat is dead while the first step's model time is > 600 (the model overwrites it, idf/cross_model.py:597-598:
        every schedule with >= 3 steps), so it is only tiled when it can be read."""
        out = {}
        if kw.get("init_flow") is not None:
            out["init_flow"] = kw["init_flow"].repeat_interleave(n_batch, dim=0)
        if kw.get("init_feat") is not None and self.tables.model_time(self.num_timesteps - 1) <= 600:
            out["init_feat"] = kw["init_feat"].repeat_interleave(n_batch, dim=0)
        return out

    def p_sample_loop(self, model, shape, **kw):
        """DDPM ancestral sampling (absent from the reference): BASELINE config 4."""
        return self.ddim_sample_loop(model, shape, sampler_kind="ddpm", **kw)

    def ddim_sample_loop_progressive_only_mean(self, model, shape, **kw):
        """Generator form (idf/gaussian_diffusion.py:537-644): like the reference it yields ONCE, after the last step,
        a dict whose 'sample' and 'pred_xstart' are the hypothesis mean clamped to [-1, 1]."""
        sample, final = self.ddim_sample_loop(model, shape, **kw)
        yield final

    def ddim_sample_loop_for_training(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                      model_kwargs=None, device=None, progress=False, eta=0.0, sampling_kwargs=None,
                                      logger=None, n_batch=1, time_variant=False, iter=True, mode="train", timestep=None,
                                      pyramid=None):
        """Training-time roll-out (idf/gaussian_diffusion.py:647-782): steps S-1 ... timestep+1, NO hypothesis mean,
        clamp only; returns (sample [B*n_batch,2,G,G], feat [B,256,G,G]).  mode=None applies the denoiser's timestep
        override like sampling does; any other mode - 'train' is what training_losses_time_variant passes (:921-946,
        one document, n_batch=1, per-sample start `timestep`) - feeds the raw model time (idf/cross_model.py:574-580)."""
        final = None
        for final in self.ddim_sample_for_training(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                   denoised_fn=denoised_fn, model_kwargs=model_kwargs, device=device,
                                                   eta=eta, n_batch=n_batch, time_variant=time_variant, iter=iter,
                                                   mode=mode, timestep=timestep):
            pass
        return final["sample"], final["feat_dict"]

    def ddim_sample_for_training(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                 model_kwargs=None, device=None, progress=False, eta=0.0, logger=None, n_batch=1,
                                 time_variant=False, iter=False, mode="train", timestep=None, pyramid=None):
        if clip_denoised or denoised_fn is not None:
            raise NotImplementedError("clip_denoised / denoised_fn are off on the DvD path")
        if not time_variant or not iter:
            raise NotImplementedError("only the live time_variant=True / iter=True configuration is mirrored")
        if timestep is None or not -1 <= int(timestep) < self.num_timesteps - 1:
            raise ValueError(f"timestep must be in [-1, {self.num_timesteps - 2}]")
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
            _ = th.randn(*shape, device=dev)
            x_T = th.randn((B * n_batch, 2, G, G), device=dev)
        noise_fn = (lambda i: th.randn((B * n_batch, 2, G, G), device=dev)) if eta != 0.0 else None   # noqa: E731
        sample = sampler.sample(eng, self.tables, x_T, eta=eta, noise_fn=noise_fn, mean_hyp=False,
                                last_step=int(timestep) + 1, t_override=mode is None,
                                **self._first_step_kwargs(kw, n_batch))
        yield {"sample": sample, "pred_xstart": sample, "feat_dict": eng.feat_nchw()}

    # ---- single-step pieces with the reference's signatures -------------------------------------
    def _extract(self, arr, t, shape):
        """_extract_into_tensor (idf/gaussian_diffusion.py:1185-1195): float64 table -> gather -> float32 -> broadcast."""
        res = th.from_numpy(np.asarray(arr, dtype=np.float64)).to(t.device)[t.long()].float()
        while res.dim() < len(shape):
            res = res[..., None]
        return res.expand(shape)

    def q_sample(self, x_start, t, noise=None):
        """q(x_t | x_0) (idf/gaussian_diffusion.py:250-268)."""
        if noise is None:
            noise = th.randn_like(x_start)
        ac = np.asarray(self.alphas_cumprod, dtype=np.float64)
        return (self._extract(np.sqrt(ac), t, x_start.shape) * x_start +
                self._extract(np.sqrt(1.0 - ac), t, x_start.shape) * noise)

    def _step_index(self, t):
        i = int(t[0])
        if not bool((t == i).all()):
            raise ValueError("the timestep must be identical across the batch (the denoiser's override rule is batch-global)")
        return i

    def _single_call_kwargs(self, model, model_kwargs, i):
        """model_kwargs of a single-step call at timestep index i: the weight-dithering phase is the one the sampling
        loop uses at that index (sampler.sample: k = S-1-i), so chaining ddim_sample by hand gives the loop's bits.
        `dither_step` is an extension of THIS package's denoiser; the reference calls model(x, t, **model_kwargs) with the
        caller's keywords only (idf/gaussian_diffusion.py:327), so a callable with the reference's keyword surface - a wrapper,
        a lambda, a test double - gets exactly those and nothing else."""
        kw = dict(model_kwargs or {})
        if "dither_step" not in kw and _accepts_dither_step(model):
            kw["dither_step"] = self.num_timesteps - 1 - i
        return kw

    def _wrap_model(self, model):
        return model

    def _scale_timesteps(self, t):
        """idf/gaussian_diffusion.py:1170-1173 (the base class scales by 1000 / num_timesteps itself; SpacedDiffusion leaves
        it to the wrapped model).  `Tables.model_time` is the same float32 product, tested against both forms."""
        i = self._step_index(t)
        return th.full((t.shape[0],), self.tables.model_time(i), device=t.device)

    def _call_model(self, model, x, t, i, model_kwargs):
        """`model(x, self._scale_timesteps(t), **model_kwargs)` on `self._wrap_model(model)` (idf/gaussian_diffusion.py:327,
        idf/respace.py:80-85): a model the caller wrapped in respace._WrappedModel is not wrapped twice and receives the
        step indices."""
        kw = self._single_call_kwargs(model, model_kwargs, i)
        from .respace import _WrappedModel
        wrapped = self._wrap_model(model)
        if isinstance(wrapped, _WrappedModel):
            return wrapped(x, t.long(), **kw)
        return wrapped(x, self._scale_timesteps(t), **kw)

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """idf/gaussian_diffusion.py:294-415 for the live configuration (START_X, FIXED_LARGE/SMALL, no clipping):
        {'mean','variance','log_variance','pred_xstart','feat_dict'}; the denoiser runs on the HIP engine, the posterior
        mean on the fused scheduler kernel."""
        if clip_denoised or denoised_fn is not None:
            raise NotImplementedError("clip_denoised / denoised_fn are off on the DvD path")
        i = self._step_index(t)
        x0, feat = self._call_model(model, x, t, i, model_kwargs)
        c = self.tables.ddpm_coef(i)
        c.sigma = 0.0                                                  # mean only
        mean = ops.sched_step(c, x.float().contiguous(), x0)
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            logvar = float(self.tables.fixed_large_log_variance[i])
        else:
            logvar = float(self.posterior_log_variance_clipped[i])
        lv = th.full_like(x0, np.float32(logvar))
        return {"mean": mean, "variance": th.exp(lv), "log_variance": lv, "pred_xstart": x0, "feat_dict": feat}

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, eta=0.0):
        """One DDIM step with the reference's signature (idf/gaussian_diffusion.py:445-491):
        {'sample','pred_xstart','feat_dict'}."""
        if clip_denoised or denoised_fn is not None:
            raise NotImplementedError("clip_denoised / denoised_fn are off on the DvD path")
        i = self._step_index(t)
        x0, feat = self._call_model(model, x, t, i, model_kwargs)
        coef = self.tables.ddim_coef(i, eta)
        noise = th.randn_like(x0) if coef.sigma != 0.0 else None
        return {"sample": ops.sched_step(coef, x.float().contiguous(), x0, noise), "pred_xstart": x0, "feat_dict": feat}

    # reference-compatible single step (idf/gaussian_diffusion.py:445-491) on an explicit x0 prediction
    def ddim_step(self, x_t, x0, i, eta=0.0, noise=None):
        return ops.sched_step(self.tables.ddim_coef(i, eta), x_t, x0, noise)
