"""Per-document driver of the sampling path - mirror of train_settings/dvd/evaluation.py
(`run_sample_lr_dewarping` :80-138, the tail of `run_evaluation_docunet` :245-306) on the HIP engine.

Documents arrive either as decoded images (dict key "image_u8": [H,W,3] uint8 RGB) - then the ingest kernel
(doc_benchmark.py:75-97) and the pre-stage conditioning nets (evaluation.py:162-216, dvd_amd/prestage.py) run first -
or as dicts of ready conditioning tensors (synthetic or loaded from .npz), i.e. exactly the tensors those nets produce."""
from __future__ import annotations

import os
import time

import numpy as np
import torch as th

from . import ops, synth


def run_sample_lr_dewarping(settings, logger, diffusion, model, source, feature_size, init_flow, doc_mask,
                            seg_map_all=None, textline_map=None, init_feat=None):
    """One batch of documents through the sampler (evaluation.py:80-138): returns the clamped flow [B,2,G,G]."""
    kw = {"init_flow": init_flow, "src_feat": None, "src_64": None, "y512": source, "tmode": settings.env.train_mode,
          "mask_cat": doc_mask, "init_feat": init_feat, "iter": settings.env.iter}
    if not settings.env.use_gt_mask:
        kw["mask_y512"] = seg_map_all
    if settings.env.use_line_mask:
        kw["line_msk"] = textline_map
    logger.info("\nStarting sampling")
    B = source.shape[0]
    sample, _ = diffusion.ddim_sample_loop(
        model, (B, 2, feature_size, feature_size), noise=None, clip_denoised=settings.env.clip_denoised,
        model_kwargs=kw, eta=0.0, progress=True, denoised_fn=None, sampling_kwargs={"src_img": source}, logger=logger,
        n_batch=settings.env.n_batch, time_variant=settings.env.time_variant, pyramid=None,
        sampler_kind=getattr(settings.env, "sampler", "ddim"))
    return th.clamp(sample, min=-1, max=1)


def synthetic_documents(settings, indices):
    """Synthetic documents: with env.use_prestage_nets a page-like IMAGE (the whole pipeline runs: ingest -> pre-stage
    nets -> sampler -> unwarp), otherwise random conditioning tensors in the value ranges of SURVEY 8(d)."""
    G = settings.env.grid_size
    for i in indices:
        if getattr(settings.env, "use_prestage_nets", False):
            h, w = settings.env.full_res
            img = synth.smooth_image(f"doc{i}/image", h, w, seed=1234)
            d = {"image_u8": np.ascontiguousarray((img.transpose(1, 2, 0) * 255.0).astype(np.uint8))}
        else:
            d = synth.synth_document(i, G, seed=1234, full_res=tuple(settings.env.full_res))
        d["path"] = f"synthetic_{i:05d}"
        yield d


def image_documents(settings, indices, files):
    """Image files of a benchmark directory (doc_benchmark.py:60-62,80-83): decoded on the CPU (PIL -> RGB uint8);
    everything after the decode runs on the GPU."""
    from PIL import Image, ImageOps
    for i in indices:
        # cv2.imread applies the EXIF orientation (IMREAD_COLOR without IMREAD_IGNORE_ORIENTATION); PIL does not by itself
        im = ImageOps.exif_transpose(Image.open(files[i]))
        d = {"image_u8": np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))}
        d["path"] = os.path.splitext(os.path.basename(files[i]))[0]
        yield d


def npz_documents(settings, indices, files):
    for i in indices:
        z = np.load(files[i])
        d = {k: z[k] for k in ("y512", "mask_cat", "mask_y512", "line_msk", "src_u8")}
        d["path"] = os.path.splitext(os.path.basename(files[i]))[0]
        yield d


def prepare_conditioning(batch, device, grid, prestage_models):
    """Documents given as decoded images: ingest (cv2.resize to 512^2, / 255; doc_benchmark.py:84-88) and the pre-stage
    nets (evaluation.py:162-216) fill in y512 / mask_cat / mask_y512 / line_msk / src_u8 as DEVICE tensors.  Documents
    that already carry their conditioning tensors (synthetic ones with env.use_prestage_nets=False, .npz files of
    env.conditioning_dir) pass through untouched and need no pre-stage nets."""
    from . import prestage
    need_ingest = [d for d in batch if "image_u8" in d and "y512" not in d]
    for d in need_ingest:
        img = th.from_numpy(d["image_u8"]).to(device)
        d["y512"], d["src_u8"] = ops.ingest_u8(img, swap_rb=False, out_size=512, want_rgb=True)
    todo = [d for d in batch if any(k not in d for k in ("mask_cat", "mask_y512", "line_msk"))]
    if not todo:
        return
    if prestage_models is None:
        raise RuntimeError(f"{len(todo)} document(s) lack mask_cat / mask_y512 / line_msk and the pre-stage nets are not "
                           "loaded (env.use_prestage_nets=False): give ready conditioning tensors or load the nets")
    src = th.stack([d["y512"] if th.is_tensor(d["y512"]) else th.from_numpy(d["y512"]).to(device) for d in todo])
    cond = prestage.conditioning(*prestage_models, src, grid)
    for j, d in enumerate(todo):
        for k in ("mask_cat", "mask_y512", "line_msk"):
            d[k] = cond[k][j]


def run_evaluation_docunet(settings, logger, documents, diffusion, model, device, prestage_models=None):
    """Document loop (evaluation.py:142-327): batches `batch_docs` documents, runs ingest + the pre-stage nets for
    documents given as images, samples, unwarps the full-resolution u8 source with the fused HIP kernel and (if
    env.visualize) writes PNGs where the reference writes them."""
    env = settings.env
    out_dir = f"vis_hp/{env.eval_dataset_name}/{settings.name}/dewarped_pred"
    if env.visualize:
        os.makedirs(out_dir, exist_ok=True)
    G, B = env.grid_size, env.batch_docs
    times, results, batch = [], [], []

    def flush():
        if not batch:
            return
        nb = len(batch)
        dev_t = lambda v: v.to(device) if th.is_tensor(v) else th.from_numpy(v).to(device)  # noqa: E731
        stack = lambda k: th.stack([dev_t(d[k]) for d in batch]).contiguous()  # noqa: E731
        prepare_conditioning(batch, device, G, prestage_models)
        t0 = time.time()
        model_docs = nb
        src, msk, seg, line = stack("y512"), stack("mask_cat"), stack("mask_y512"), stack("line_msk")
        flow = run_sample_lr_dewarping(settings, logger, diffusion, model, src, G,
                                       th.zeros(nb, 2, G, G, device=device), msk, seg, line,
                                       th.zeros(nb, 256, G, G, device=device))
        th.cuda.synchronize()
        times.append((time.time() - t0) / model_docs)
        # :301-306 + viz :75-77 - one launch for the batch when its documents share a full-resolution size
        if len({tuple(d["src_u8"].shape) for d in batch}) == 1:
            outs = ops.unwarp_u8_batch(flow.contiguous(), stack("src_u8"))
        else:
            outs = [ops.unwarp_u8(flow[j:j + 1].contiguous(), dev_t(d["src_u8"]).contiguous()) for j, d in enumerate(batch)]
        for j, d in enumerate(batch):
            out = outs[j]
            results.append((d["path"], out))
            if env.visualize:
                from PIL import Image
                Image.fromarray(out.cpu().numpy()).save(os.path.join(out_dir, f"warped_{d['path']}.png"))
        batch.clear()

    for d in documents:
        batch.append(d)
        if len(batch) == B:
            flush()
    flush()
    if times:
        print(len(times))
        print("Elapsed time:{:.2f} avg_second ".format(sum(times) / len(times)))
    return results
