"""Per-document driver of the sampling path - mirror of train_settings/dvd/evaluation.py
(`run_sample_lr_dewarping` :80-138, `run_evaluation_docunet` :142-327, both with the reference's positional signatures) on
the HIP engine.

Documents arrive either as decoded images (dict key "image_u8": [H,W,3] uint8 RGB) - then the ingest kernel
(doc_benchmark.py:75-97) and the pre-stage conditioning nets (evaluation.py:162-216, dvd_amd/prestage.py) run first -
or as dicts of ready conditioning tensors (synthetic or loaded from .npz), i.e. exactly the tensors those nets produce."""
from __future__ import annotations

import os
import time

import numpy as np
import torch as th

from . import ops, synth


def run_sample_lr_dewarping(settings, logger, diffusion, model, radius, source, feature_size, raw_corr, init_flow, c20,
                            source_64, pyramid, doc_mask, seg_map_all=None, textline_map=None, init_feat=None):
    """One batch of documents through the sampler, with the reference's positional signature (evaluation.py:80-138; call
    site :247-265): returns the clamped flow [B,2,G,G].  `radius`, `raw_corr`, `source_64` and `pyramid` are dead in the
    reference's body as well (the correlation code is commented out, the conv pyramid belongs to the denoiser); `c20` is
    the reference's `src_feat`, None on the live configuration (train_VGG=True, :219-221).  B = source.shape[0] documents
    are sampled in one engine batch (the reference: B = 1)."""
    kw = {"init_flow": init_flow, "src_feat": c20, "src_64": None, "y512": source, "tmode": settings.env.train_mode,
          "mask_cat": doc_mask, "init_feat": init_feat, "iter": settings.env.iter}
    if not settings.env.use_gt_mask:
        kw["mask_y512"] = seg_map_all
    if settings.env.use_line_mask:
        kw["line_msk"] = textline_map
    logger.info("\nStarting sampling")
    B = source.shape[0]
    extra = {}
    if getattr(settings.env, "sampler", "ddim") != "ddim":
        extra["sampler_kind"] = settings.env.sampler          # engine-side extension (BASELINE configs[3]: 'ddpm')
    sample, _ = diffusion.ddim_sample_loop(
        model, (B, 2, feature_size, feature_size), noise=None, clip_denoised=settings.env.clip_denoised,
        model_kwargs=kw, eta=0.0, progress=True, denoised_fn=None, sampling_kwargs={"src_img": source}, logger=logger,
        n_batch=settings.env.n_batch, time_variant=settings.env.time_variant, pyramid=pyramid, **extra)
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
        img = d["image_u8"]
        img = (img if th.is_tensor(img) else th.from_numpy(img)).to(device).contiguous()
        d["y512"], d["src_u8"] = ops.ingest_u8(img, swap_rb=False, out_size=512, want_rgb=True)
    todo = [d for d in batch if any(k not in d for k in ("mask_cat", "mask_y512", "line_msk"))]
    if not todo:
        return
    if prestage_models is None:
        raise RuntimeError(f"{len(todo)} document(s) lack mask_cat / mask_y512 / line_msk and the pre-stage nets are not "
                           "loaded (env.use_prestage_nets=False): give ready conditioning tensors or load the nets")
    src = th.stack([(d["y512"] if th.is_tensor(d["y512"]) else th.from_numpy(d["y512"])).to(device) for d in todo])
    cond = prestage.conditioning(*prestage_models, src, grid)
    for j, d in enumerate(todo):
        for k in ("mask_cat", "mask_y512", "line_msk"):
            d[k] = cond[k][j]


def documents_of(item):
    """One loader item -> the documents it holds.  Two item shapes are accepted:
      * the reference's (doc_benchmark.py:91-97 through `DataLoader(batch_size=b)`): `source_image` [b,3,512,512] f32 in 0..1
        (may be absent: computed on the GPU, see datasets/doc_dataset/doc_benchmark.py), `source_image_ori` [b,3,H,W]
        (0..255, float or uint8), `path` list of b file names;
      * this package's document dicts (`image_u8` [H,W,3] or ready conditioning tensors + `src_u8`, `path` = a stem)."""
    if "source_image_ori" not in item and "source_image" not in item:
        return [item]
    ori = item.get("source_image_ori", item.get("source_image"))
    paths = item["path"]
    if isinstance(paths, str):
        paths, ori = [paths], ori[None]
    docs = []
    for j, path in enumerate(paths):
        d = {"path": path, "source_vis": ori[j]}
        if "source_image" in item:
            src = item["source_image"]
            d["y512"] = src[j] if src.dim() == 4 else src
        docs.append(d)
    return docs


def _source_u8(d, device):
    """The full-resolution source of one document as [H,W,3] uint8 on the device, or None when it cannot be had exactly
    (a float `source_image_ori` that is not integer-valued in 0..255: the f32 tail kernel is used then)."""
    if "src_u8" in d:
        v = d["src_u8"]
        return (v if th.is_tensor(v) else th.from_numpy(v)).to(device).contiguous()
    vis = d["source_vis"].to(device)
    if vis.dtype == th.uint8:
        return vis.permute(1, 2, 0).contiguous()
    as_u8 = vis.clamp(0, 255).to(th.uint8)
    if not th.equal(as_u8.to(vis.dtype), vis):
        return None
    return as_u8.permute(1, 2, 0).contiguous()


def run_evaluation_docunet(settings, logger, val_loader, diffusion, model, pretrained_dewarp_model,
                           pretrained_line_seg_model=None, pretrained_seg_model=None):
    """Document loop with the reference's positional signature (evaluation.py:142-327; call site val_TDiff.py:103-104).
    `val_loader` yields the reference's dicts or this package's documents (`documents_of`); `env.batch_docs` documents are
    batched per pass (the reference: 1): ingest + the three pre-stage nets for documents that arrive as images
    (:162-216), the sampler (:247-265), then the tail (:301-306 + visualization_utils.py:75-77) as ONE fused u8 launch per
    batch, and - if env.visualize - `visualize_dewarping` writes the PNG where the reference writes it.
    The pre-stage models may all be None when every document carries ready conditioning tensors.
    Returns [(path, uint8 [H,W,3] device tensor)] (the reference returns None)."""
    from utils_flow.visualization_utils import visualize_dewarping
    env = settings.env
    device = next(model.parameters()).device
    nets = (pretrained_dewarp_model, pretrained_seg_model, pretrained_line_seg_model)
    prestage_models = None if all(m is None for m in nets) else nets
    os.makedirs(f"vis_hp/{env.eval_dataset_name}/{settings.name}", exist_ok=True)
    G, B = env.grid_size, env.batch_docs
    times, results, batch = [], [], []

    def flush():
        if not batch:
            return
        nb = len(batch)
        dev_t = lambda v: v.to(device) if th.is_tensor(v) else th.from_numpy(v).to(device)  # noqa: E731
        stack = lambda k: th.stack([dev_t(d[k]) for d in batch]).contiguous()  # noqa: E731
        for d in batch:                                     # reference-shaped items: the source as u8 for ingest / tail
            if "source_vis" in d and "src_u8" not in d:
                u8 = _source_u8(d, device)
                if u8 is not None:
                    d["src_u8"] = u8
                    if "y512" not in d:
                        d["image_u8"] = u8
                elif "y512" not in d:
                    raise ValueError(f"{d['path']}: no 'source_image' and a non-integer 'source_image_ori' to make it from")
        prepare_conditioning(batch, device, G, prestage_models)
        t0 = time.time()
        src, msk, seg, line = stack("y512"), stack("mask_cat"), stack("mask_y512"), stack("line_msk")
        flow = run_sample_lr_dewarping(settings, logger, diffusion, model, 4, src, G, None,
                                       th.zeros(nb, 2, G, G, device=device), None, None, None, msk, seg, line,
                                       th.zeros(nb, 256, G, G, device=device))
        th.cuda.synchronize()
        times.append((time.time() - t0) / nb)
        # :301-306 + viz :75-77 - one launch for the batch when its documents are byte images of one full-resolution size
        if all("src_u8" in d for d in batch) and len({tuple(d["src_u8"].shape) for d in batch}) == 1:
            outs = ops.unwarp_u8_batch(flow.contiguous(), stack("src_u8"))
        else:
            outs = []
            for j, d in enumerate(batch):
                fj = flow[j:j + 1].contiguous()
                if "src_u8" in d:
                    outs.append(ops.unwarp_u8(fj, dev_t(d["src_u8"]).contiguous()))
                else:   # a float source that is not a byte image: the fused f32 tail, truncated like numpy's astype(uint8)
                    outs.append(ops.unwarp_f32(fj, d["source_vis"].to(device).float()[None].contiguous()).to(th.uint8))
        for j, d in enumerate(batch):
            out = outs[j]
            results.append((d["path"], out))
            if env.visualize:
                name = d["path"] if os.path.splitext(d["path"])[1] else d["path"] + ".png"
                visualize_dewarping(settings, None, d, len(results) - 1, None, [name], warped_u8=out)
        batch.clear()

    for item in val_loader:
        for d in documents_of(item):
            batch.append(d)
            if len(batch) == B:
                flush()
    flush()
    if times:
        print(len(times))
        print("Elapsed time:{:.2f} avg_second ".format(sum(times) / len(times)))
    return results
