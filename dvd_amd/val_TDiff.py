"""`run(settings)`: the sampling plug-in with the reference's structure (train_settings/dvd/val_TDiff.py:40-116):
set up the process group, build model + diffusion from `settings.env`, load the checkpoint (strict=False), shard the
documents over the ranks, sample, barrier."""
from __future__ import annotations

import glob
import os

import numpy as np
import torch
import torch.distributed as dist

from . import dist_util, logger, synth
from .evaluation import npz_documents, run_evaluation_docunet, synthetic_documents
from .script_util import args_to_dict, create_model_and_diffusion, model_and_diffusion_defaults


def get_parameter_number(net):
    total = sum(p.numel() for p in net.parameters())
    return {"Total": total, "Trainable": sum(p.numel() for p in net.parameters() if p.requires_grad)}


def _want_synthetic_weights(env):
    """Synthetic stand-in weights are for the synthetic benchmark only: a mistyped checkpoint path on a real dataset
    must fail, not write plausible-looking garbage where the reference writes its results."""
    flag = getattr(env, "synthetic_weights_if_missing", None)
    return env.eval_dataset_name == "synthetic" if flag is None else bool(flag)


def run(settings):
    own_group = dist_util.setup_dist()
    env = settings.env
    logger.configure(dir=f"SAMPLING_{env.eval_dataset_name}_{settings.name}")
    logger.log("Loading model and diffusion...")
    model, diffusion = create_model_and_diffusion(
        device=dist_util.dev(), train_mode=env.train_mode, tv=env.time_variant, grid_size=env.grid_size,
        **args_to_dict(settings, model_and_diffusion_defaults().keys()))
    setattr(diffusion, "settings", settings)
    # rank 0 alone reads the checkpoint (val_TDiff.py:79); the other ranks receive the packed blob below
    if dist_util.rank() == 0:
        if os.path.exists(env.model_path):
            model.cpu().load_state_dict(dist_util.load_state_dict(env.model_path, map_location="cpu"), strict=False)
            logger.log(f"Model loaded with {env.model_path}")
        elif _want_synthetic_weights(env):
            sd = synth.synth_state_dict(env.grid_size, seed=7)
            model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            logger.log(f"{env.model_path} not found: using deterministic synthetic weights (seed 7)")
        else:
            raise FileNotFoundError(f"{env.model_path} (set env.synthetic_weights_if_missing=True to sample with "
                                    "synthetic stand-in weights)")
    _require_gpu()
    model.to(dist_util.dev())
    print(get_parameter_number(model))
    model.eval()
    # the path's ONE collective, issued eagerly and unconditionally by every rank BEFORE the documents are sharded:
    # rank 0 packs, one flat broadcast (a rank whose shard is empty still takes part, then goes to the barrier)
    model.materialize_blob()

    if env.eval_dataset_name == "synthetic" or not env.conditioning_dir:
        n_docs = env.num_synthetic_docs
        mine = dist_util.shard_documents(n_docs)
        documents = synthetic_documents(settings, mine)
    else:
        files = sorted(glob.glob(os.path.join(env.conditioning_dir, "*.npz")))
        mine = dist_util.shard_documents(len(files))
        documents = npz_documents(settings, mine, files)
    logger.info(f"rank {dist_util.rank()}/{dist_util.world_size()}: {len(mine)} documents")
    logger.info("Starting sampling")
    results = run_evaluation_docunet(settings, logger, documents, diffusion, model, dist_util.dev())
    if dist.is_initialized():
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    logger.log("sampling complete")
    return results


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("the DvD engine needs an MI355X (no CPU fallback)")
