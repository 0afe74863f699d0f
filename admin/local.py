"""Per-installation settings of the sampling run: same attribute surface as the reference's
admin/local.py (every name the sampling path or `args_to_dict` reads), expressed as a table.
Engine-side additions are at the end."""

_DEFAULTS = {
    # directories
    "workspace_dir": "checkpoints",
    # what to evaluate on: 'synthetic' generates documents on the fly; any other name expects
    # per-document conditioning .npz files in `conditioning_dir`
    "eval_dataset_name": "synthetic",
    "eval_dataset": "",
    "dataset_name": "doc3d",
    "time_variant": True,
    # live model / loop configuration (reference admin/local.py:27-35,55-69,81-84)
    "train_mode": "stage_1_dit_cross",
    "iter": True,
    "train_VGG": True,
    "use_gt_mask": False,
    "use_line_mask": True,
    "use_init_flow": False,
    "diffusion_steps": 3,
    "image_size": 64,
    "flow_size": (64, 64),
    "num_channels": 128,
    "num_res_blocks": 3,
    "num_heads": 4,
    "num_heads_upsample": -1,
    "attention_resolutions": "16,8",
    "dropout": 0.0,
    "learn_sigma": False,
    "sigma_small": False,
    "class_cond": False,
    "noise_schedule": "cosine",
    "use_kl": False,
    "predict_xstart": True,
    "rescale_timesteps": True,
    "rescale_learned_sigmas": True,
    "use_checkpoint": False,
    "use_scale_shift_norm": True,
    "clip_denoised": False,
    "timestep_respacing": "",
    "n_batch": 2,            # hypotheses per document
    "visualize": True,
    "use_sr_net": False,
    "val_batch_size": 1,
    "model_path": "checkpoints/model1852000.pt",
    "seg_model_path": "checkpoints/seg.pth",
    "line_seg_model_path": "checkpoints/line_model2.pth",
    "new_seg_model_path": "checkpoints/seg_model.pth",
    # ---- engine-side additions (defaults reproduce the reference's behaviour) ----
    "grid_size": 64,         # coordinate grid G (reference: fixed 64)
    "batch_docs": 1,         # documents sampled together per GPU (reference: 1)
    "sampler": "ddim",       # 'ddim' | 'ddpm'
    "num_synthetic_docs": 4,
    "full_res": (1024, 768), # synthetic full-resolution source size (H, W)
    "conditioning_dir": "",   # directory of per-document conditioning .npz files (skips ingest + pre-stage nets)
    "num_workers": 0,         # DataLoader workers of the image-directory path (they only decode; the reference: 8)
    # run the pre-stage conditioning nets (GeoTr_Seg_Inf.msk, Seg, line UNet; reference evaluation.py:162-216) on the
    # document images, as the reference does; False = synthetic documents carry random conditioning tensors
    "use_prestage_nets": True,
    # None = only when eval_dataset_name == 'synthetic' (a missing checkpoint on a real dataset raises)
    "synthetic_weights_if_missing": None,
}


class EnvironmentSettings:
    def __init__(self):
        for key, value in _DEFAULTS.items():
            setattr(self, key, value)
        self.tensorboard_dir = self.workspace_dir
        self.pretrained_networks = self.workspace_dir
        self.pre_trained_models_dir = self.workspace_dir + "/backup"
