"""Factory with the reference's names and keyword surface (idf/script_util.py:11-90,206-258)."""
from __future__ import annotations

from . import gaussian_diffusion as gd
from .cross_model import DiT_models2
from .respace import SpacedDiffusion, space_timesteps


def model_and_diffusion_defaults():
    return dict(image_size=256, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
                attention_resolutions="16,8", dropout=0.0, learn_sigma=False, sigma_small=False, class_cond=False,
                diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False,
                predict_xstart=True, rescale_timesteps=True, rescale_learned_sigmas=True, use_checkpoint=False,
                use_scale_shift_norm=True)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, sigma_small, num_channels, num_res_blocks,
                               num_heads, num_heads_upsample, attention_resolutions, dropout, diffusion_steps,
                               noise_schedule, timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, device, train_mode, tv,
                               grid_size=64):
    model = create_model(image_size, num_channels, num_res_blocks, learn_sigma=learn_sigma, class_cond=class_cond,
                         use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions,
                         num_heads=num_heads, num_heads_upsample=num_heads_upsample,
                         use_scale_shift_norm=use_scale_shift_norm, dropout=dropout, device=device,
                         train_mode=train_mode, tv=tv, grid_size=grid_size)
    diffusion = create_gaussian_diffusion(steps=diffusion_steps, learn_sigma=learn_sigma, sigma_small=sigma_small,
                                          noise_schedule=noise_schedule, use_kl=use_kl, predict_xstart=predict_xstart,
                                          rescale_timesteps=rescale_timesteps,
                                          rescale_learned_sigmas=rescale_learned_sigmas,
                                          timestep_respacing=timestep_respacing)
    return model, diffusion


def create_model(image_size, num_channels, num_res_blocks, learn_sigma, class_cond, use_checkpoint,
                 attention_resolutions, num_heads, num_heads_upsample, use_scale_shift_norm, dropout, device,
                 train_mode, tv, grid_size=64):
    """Only the live train_mode is implemented: 'stage_1_dit_cross' ignores every UNet hyper-parameter and
    builds DiT-S/2 on the latent grid (idf/script_util.py:155-162; the reference hard-codes 512//8 = 64,
    `grid_size` generalises it)."""
    if train_mode != "stage_1_dit_cross":
        raise ValueError(f"unsupported train mode: {train_mode} (the HIP engine implements 'stage_1_dit_cross')")
    return DiT_models2["DiT-S/2"](input_size=grid_size, in_channels=2, tv=tv)


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing=""):
    betas = gd.get_named_beta_schedule(noise_schedule, steps)
    loss_type = gd.LossType.RESCALED_KL if use_kl else (gd.LossType.RESCALED_MSE if rescale_learned_sigmas
                                                        else gd.LossType.MSE)
    if not timestep_respacing:
        timestep_respacing = [steps]
    var_type = (gd.ModelVarType.LEARNED_RANGE if learn_sigma else
                (gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE))
    return SpacedDiffusion(use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas,
                           model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
                           model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps)


def args_to_dict(args, keys):
    return {k: getattr(args.env, k) for k in keys}
