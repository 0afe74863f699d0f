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
    if getattr(env, "use_init_flow", False):
        raise NotImplementedError("env.use_init_flow needs GeoTr's `ref_bm`, whose weights the reference itself never "
                                  "loads (val_TDiff.py:57-58 only reloads `.msk`): not a live configuration")
    # rank 0 alone reads the checkpoint (val_TDiff.py:79); the other ranks receive the packed blob below.  A failure on
    # rank 0 (missing / corrupt file) is agreed on by every rank BEFORE the broadcast, so all ranks raise together.
    failure = None
    if dist_util.rank() == 0:
        try:
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
        except Exception as e:  # noqa: BLE001 - re-raised on every rank by raise_together
            failure = e
    dist_util.raise_together(failure, "loading the denoiser checkpoint")
    _require_gpu()
    model.to(dist_util.dev())
    print(get_parameter_number(model))
    model.eval()
    pre, failure = None, None
    if getattr(env, "use_prestage_nets", False):
        try:
            pre = load_prestage_models(env)
        except Exception as e:  # noqa: BLE001
            failure = e
        dist_util.raise_together(failure, "loading the pre-stage checkpoints")
    # the path's ONE collective, issued eagerly and unconditionally by every rank BEFORE the documents are sharded:
    # rank 0 packs every model (denoiser + the three pre-stage nets) into one flat buffer, one broadcast (a rank whose
    # shard is empty still takes part, then goes to the barrier)
    dist_util.materialize_blobs([model] + list(pre or ()))

    if env.eval_dataset_name == "synthetic":
        n_docs = env.num_synthetic_docs
        mine = dist_util.shard_documents(n_docs)
        documents = synthetic_documents(settings, mine)
    elif env.conditioning_dir:
        files = sorted(glob.glob(os.path.join(env.conditioning_dir, "*.npz")))
        mine = dist_util.shard_documents(len(files))
        documents = npz_documents(settings, mine, files)
    else:      # a benchmark directory of images: the reference's Doc_benchmark behind a DataLoader (val_TDiff.py:93-104)
        import datasets
        from torch.utils.data import DataLoader, Subset
        from utils_data.image_transforms import ArrayToTensor
        # the workers only decode (PIL); uint8 CHW crosses PCIe (the reference ships float32: get_float=True), the
        # resize to 512 x 512 and everything after it runs on the GPU.  Unshuffled: rank r owns files r, r+world, ...
        test_set = datasets.Doc_benchmark(env.eval_dataset, ArrayToTensor(get_float=False))
        mine = dist_util.shard_documents(len(test_set))
        documents = DataLoader(Subset(test_set, list(mine)), batch_size=1, shuffle=False, drop_last=False,
                               num_workers=int(getattr(env, "num_workers", 0)))
    logger.info(f"rank {dist_util.rank()}/{dist_util.world_size()}: {len(mine)} documents")
    logger.info("Starting sampling")
    dewarp, seg, line = pre if pre is not None else (None, None, None)
    results = run_evaluation_docunet(settings, logger, documents, diffusion, model, dewarp, line, seg)   # val_TDiff.py:103-104
    if dist.is_initialized():
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    logger.log("sampling complete")
    return results


def load_prestage_models(env):
    """The three pre-stage nets exactly as val_TDiff.py:57-75 builds and loads them: GeoTr_Seg_Inf with
    reload_segmodel(model.msk, seg_model_path), UNet and Seg from {'model': state_dict} checkpoints.  Rank 0 alone reads
    the files; the packed weights reach the other ranks in the one flat broadcast."""
    from .prestage import GeoTr_Seg_Inf, Seg, UNet, reload_segmodel
    dewarp, line, seg = GeoTr_Seg_Inf(), UNet(n_channels=3, n_classes=1), Seg()
    if dist_util.rank() == 0:
        def tt(sd):
            return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        jobs = ((env.seg_model_path, lambda p: reload_segmodel(dewarp.msk, p),
                 lambda: dewarp.msk.load_state_dict(tt(synth.synth_convnet_state_dict("u2netp", 11)), strict=True)),
                (env.line_seg_model_path,
                 lambda p: line.load_state_dict(dist_util.load_state_dict(p, map_location="cpu")["model"], strict=True),
                 lambda: line.load_state_dict(tt(synth.synth_convnet_state_dict("unet", 13)), strict=True)),
                (env.new_seg_model_path,
                 lambda p: seg.load_state_dict(dist_util.load_state_dict(p, map_location="cpu")["model"], strict=True),
                 lambda: seg.load_state_dict(tt(synth.synth_convnet_state_dict("u2netp", 22, prefix="msk.")), strict=True)))
        for path, from_file, synthetic in jobs:
            if os.path.exists(path):
                from_file(path)
                logger.log(f"loaded {path}")
            elif _want_synthetic_weights(env):
                synthetic()
                logger.log(f"{path} not found: using deterministic synthetic weights")
            else:
                raise FileNotFoundError(path)
    for m in (dewarp, seg, line):
        m.to(dist_util.dev())
        m.eval()
    return dewarp, seg, line


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("the DvD engine needs an MI355X (no CPU fallback)")
