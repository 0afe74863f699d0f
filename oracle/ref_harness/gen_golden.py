#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

TEST INFRASTRUCTURE.  Runs only in the build container (it needs /root/reference, which
does not exist on the GPU box); its outputs - small .npz files of inputs (as seeds or
arrays) and the reference's outputs - are committed, this script is how they were made:

    python oracle/ref_harness/gen_golden.py [--only G1,G2,...]

The reference is imported unmodified from /root/reference through the import shims in
oracle/ref_harness/shims (third-party packages absent from this image).  Weights are the
synthetic, counter-based tensors of dvd_amd.synth loaded with load_state_dict(strict=True),
so the oracle and the HIP engine can regenerate exactly the same model anywhere.

Fixtures:
  G1 schedule.npz      cosine schedule tables for S in {3,10,50,250}, t_model sequences
  G2 forward_g{G}.npz  one DiT.forward per t-class at G in {16,32,64}, N=2, + checkpoints
  G3 loop_g{G}_s{S}.npz full ddim_sample_loop (G=64: S=3,10) / training-variant loop (G=16,32)
  G9 loop_g{G}_s{S}_tame.npz the same loops on the TAME weight family (out_gain = 1.6 / S), G=64 S=10,50 and G=32 S=50
  G4 ddim_step.npz     ddim_sample on random (x_t,x0) for every t of S=50; p_mean_variance S=250
  G5 unwarp.npz        upsample+affine+grid_sample+uint8 tail on a small ragged image
  G6 grid_sample.npz   register_model2 on the per-step feature-warp shape
  G8 prestage_g16.npz  the pre-stage conditioning nets (evaluation.py:162-216: GeoTr_Seg_Inf.msk, Seg, line UNet + the six
                       align_corners=False resizes) on synthetic weights: mask_cat / mask_y512 / line_msk for G=16
  G7 rollout_train_g{G}_s{S}.npz  ddim_sample_loop_for_training(mode='train', n_batch=1, timestep=k) with a
                       non-zero init_flow: the call training_losses_time_variant makes (gaussian_diffusion.py:921-946)
"""
import argparse
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from dvd_amd import synth  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
SEED_W = 7          # weight seed used by every fixture
SEED_IN = 1234      # input seed


def _workdir():
    d = tempfile.mkdtemp(prefix="dvd_ref_work_")
    os.makedirs(os.path.join(d, "vis_hp", "debug_vis"))
    os.chdir(d)
    return d


def import_reference():
    from train_settings.dvd.improved_diffusion import (cross_model, gaussian_diffusion,  # noqa
                                                       respace, script_util, dist_util)
    from datasets.utils import warping
    return cross_model, gaussian_diffusion, respace, script_util, dist_util, warping


def build_model(cross_model, grid, script_util=None, dist_util=None, out_gain=1.0):
    """Reference model at coordinate-grid size `grid` with the synthetic state dict (out_gain: synth.tame_gain(S) for the
    tame family whose S-step roll-out stays inside (-1, 1))."""
    if grid == 64 and script_util is not None:
        # through the reference's own factory (script_util.py:155-162)
        from admin.local import EnvironmentSettings

        class S:
            pass
        s = S()
        s.env = EnvironmentSettings()
        model, _ = script_util.create_model_and_diffusion(
            device=dist_util.dev(), train_mode=s.env.train_mode, tv=s.env.time_variant,
            **script_util.args_to_dict(s, script_util.model_and_diffusion_defaults().keys()))
    else:
        model = cross_model.DiT_models2["DiT-S/2"](input_size=grid, in_channels=2, tv=True)
    spec = synth.state_dict_spec(grid)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(spec.keys()), (
        "state_dict key order differs from dvd_amd.synth.state_dict_spec:\n"
        + "\n".join(f"{a} | {b}" for a, b in zip(ref_sd.keys(), spec.keys()) if a != b))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]), (k, v.shape, spec[k][0])
    sd = synth.synth_state_dict(grid, SEED_W, out_gain=out_gain)
    # the computed tables must agree with what the reference computes itself
    for k in ("noised_obs_pos_embed", "decoder.position_dec.h_position_encoder",
              "decoder.position_dec.w_position_encoder"):
        err = float(np.abs(ref_sd[k].numpy() - sd[k]).max())
        assert err < 2e-6, (k, err)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    return model


def make_diffusion(script_util, steps):
    return script_util.create_gaussian_diffusion(
        steps=steps, learn_sigma=False, sigma_small=False, noise_schedule="cosine", use_kl=False,
        predict_xstart=True, rescale_timesteps=True, rescale_learned_sigmas=True,
        timestep_respacing="")


def doc_inputs(grid, doc=0):
    d = synth.synth_document(doc, grid, SEED_IN)
    return {k: torch.from_numpy(v)[None] for k, v in d.items()}


def summ(t):
    t = t.detach().float().reshape(-1)
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item()], dtype=np.float64)


# ------------------------------------------------------------------------------- G1
def gen_schedule(mods):
    _, gd, respace, script_util, _, _ = mods
    out = {}
    for S in (3, 10, 50, 250):
        diff = make_diffusion(script_util, S)
        for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                     "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            out[f"s{S}/{name}"] = np.asarray(getattr(diff, name), dtype=np.float64)
        out[f"s{S}/timestep_map"] = np.asarray(diff.timestep_map, dtype=np.int64)
        # FIXED_LARGE variance tables as p_mean_variance builds them (gaussian_diffusion.py:365-378)
        seen = {}

        def fake(x, t, **kw):
            seen["t"] = t.clone()
            return x
        tm = []
        var, logvar = [], []
        x = torch.zeros(2, 2, 4, 4)
        for i in range(S):
            t = torch.tensor([i, i])
            o = diff.p_mean_variance(fake, x, t, clip_denoised=False, model_kwargs={})
            tm.append(float(seen["t"][0]))
            var.append(float(o["variance"][0, 0, 0, 0]))
            logvar.append(float(o["log_variance"][0, 0, 0, 0]))
        out[f"s{S}/t_model_raw"] = np.asarray(tm, dtype=np.float32)       # respace.py:118-123
        out[f"s{S}/fixed_large_var_f32"] = np.asarray(var, dtype=np.float32)
        out[f"s{S}/fixed_large_logvar_f32"] = np.asarray(logvar, dtype=np.float32)
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **out)
    print("G1 schedule.npz", len(out), "arrays")


# ------------------------------------------------------------------------------- G2
def gen_forward(mods, grid):
    cross_model, gd, respace, script_util, dist_util, warping = mods
    model = build_model(cross_model, grid, script_util, dist_util)
    inp = doc_inputs(grid)
    N = 2
    rep = lambda v: v.repeat(N, 1, 1, 1)  # noqa: E731
    x = torch.from_numpy(synth.synth_noise(0, N, grid, SEED_IN))
    init_flow = torch.from_numpy(synth.uniform("g2/init_flow", (N, 2, grid, grid), -0.3, 0.3, SEED_IN))
    init_feat = torch.from_numpy(synth.uniform("g2/init_feat", (N, 256, grid, grid), 0.0, 1.5, SEED_IN))
    out = {"grid": np.int64(grid), "seed_w": np.int64(SEED_W), "seed_in": np.int64(SEED_IN)}
    out["x"] = x.numpy()
    out["init_flow"] = init_flow.numpy()
    # t classes (cross_model.py:575-580): >600 -> [2,2]; (300,600) -> [1,1]; else raw
    for tag, tval in (("t2", 666.6667), ("t1", 400.0), ("raw0", 0.0), ("raw600", 600.0), ("raw200", 200.0)):
        marks = {}
        hooks = []

        def grab(name):
            def h(mod, args, res):
                marks[name] = res
            return h
        hooks.append(model.obs_embedder.register_forward_hook(grab("obs_tok")))
        hooks.append(model.t_embedder.register_forward_hook(grab("t_emb")))
        hooks.append(model.c_embedder.register_forward_hook(grab("c_tok")))
        hooks.append(model.m_embedder.register_forward_hook(grab("m_tok")))
        hooks.append(model.l_embedder.register_forward_hook(grab("l_tok")))
        hooks.append(model.r_embedder.register_forward_hook(grab("r_tok")))
        hooks.append(model.blocks[-1].register_forward_hook(grab("block")))
        hooks.append(model.decoder.position_dec.register_forward_hook(grab("dec_pos")))
        for j, lyr in enumerate(model.decoder.layer_stack):
            hooks.append(lyr.register_forward_hook(grab(f"dec{j}")))
        hooks.append(model.decoder.register_forward_hook(grab("dec_out")))
        hooks.append(model.final_layer2.register_forward_hook(grab("final")))
        t = torch.tensor([tval] * N, dtype=torch.float32)
        with torch.no_grad():
            x0, feat = model(x.clone(), t, y512=rep(inp["y512"]), mask_y512=rep(inp["mask_y512"]),
                             init_flow=init_flow.clone(), tv=True, tmode="stage_1_dit_cross",
                             line_msk=rep(inp["line_msk"]), mask_cat=rep(inp["mask_cat"]),
                             init_feat=init_feat.clone(), iter=True, mode=None)
        for h in hooks:
            h.remove()
        out[f"{tag}/t_in"] = np.float32(tval)
        out[f"{tag}/x0"] = x0.numpy()
        blk = marks.pop("block")          # (x4, x3, x2, x1)
        marks["blk_x1"], marks["blk_x2"], marks["blk_x3"], marks["blk_x4"] = blk[3], blk[2], blk[1], blk[0]
        marks["dec_pos"] = marks["dec_pos"].flatten(2).transpose(1, 2)   # -> [N,T,C] token-major
        for name, v in marks.items():
            v = v.detach()
            out[f"{tag}/ck/{name}/stats"] = summ(v)
            out[f"{tag}/ck/{name}/head"] = v.reshape(-1)[:64].numpy().copy()
            out[f"{tag}/ck/{name}/tail"] = v.reshape(-1)[-64:].numpy().copy()
        if tag == "t2":
            f0 = feat[0]
            out["feat/stats"] = summ(f0)
            out["feat/sub"] = f0[::16, :: max(1, grid // 16), :: max(1, grid // 16)].numpy().copy()
            if grid == 16:
                out["feat/full"] = f0.numpy().copy()
        print(f"  G2 grid {grid} {tag}: x0 mean {x0.mean():+.5f} std {x0.std():.5f}")
    np.savez_compressed(os.path.join(GOLD, f"forward_g{grid}.npz"), **out)
    print("G2", f"forward_g{grid}.npz")


# ------------------------------------------------------------------------------- G3
def gen_loop(mods, grid, steps, full_blocks, tame=False):
    """tame=True (G9): the tame weight family, every 7th step's x0 + the last kept (a 50-step file stays ~1 MB)."""
    cross_model, gd, respace, script_util, dist_util, warping = mods
    model = build_model(cross_model, grid, script_util, dist_util, out_gain=synth.tame_gain(steps) if tame else 1.0)
    if not full_blocks:
        # F2: only blocks[-1] is live; bit-identical and 10x cheaper
        model.blocks = torch.nn.ModuleList([model.blocks[-1]])
    diff = make_diffusion(script_util, steps)
    inp = doc_inputs(grid)
    H = 2
    rec = {"t": [], "x0": [], "x_in": []}

    def pre(mod, args, kwargs):
        rec["x_in"].append(args[0].detach().clone().numpy())
        rec["t"].append(float(args[1][0]))

    def post(mod, args, kwargs, res):
        rec["x0"].append(res[0].detach().clone().numpy())
    h1 = model.register_forward_pre_hook(pre, with_kwargs=True)
    h2 = model.register_forward_hook(post, with_kwargs=True)
    kw = {"init_flow": torch.zeros(1, 2, grid, grid), "src_feat": None, "src_64": None,
          "y512": inp["y512"], "tmode": "stage_1_dit_cross", "mask_cat": inp["mask_cat"],
          "init_feat": torch.zeros(1, 256, grid, grid), "iter": True,
          "mask_y512": inp["mask_y512"], "line_msk": inp["line_msk"]}
    seed = 4321 + grid + steps
    # replicate the reference's draws to learn x_T (gaussian_diffusion.py:562,569 / :721,728)
    torch.manual_seed(seed)
    _ = torch.randn(1, 2, grid, grid)
    x_T = torch.randn(H, 2, grid, grid)
    torch.manual_seed(seed)
    t0 = time.time()
    if grid == 64:
        sample, final = diff.ddim_sample_loop(
            model, (1, 2, grid, grid), noise=None, clip_denoised=False, model_kwargs=kw, eta=0.0,
            progress=False, denoised_fn=None, sampling_kwargs={"src_img": inp["y512"]}, logger=None,
            n_batch=H, time_variant=True, pyramid=None)
        kind = "ddim_sample_loop"
    else:
        # the only loop variant that picks `base` by grid size (gaussian_diffusion.py:744-752);
        # it does NOT average the hypotheses (:776), only clamps.
        kw2 = {k: v for k, v in kw.items() if k not in ("tmode", "iter")}
        sample, _ = diff.ddim_sample_loop_for_training(
            model, (1, 2, grid, grid), noise=None, clip_denoised=False, model_kwargs=kw2, eta=0.0,
            n_batch=H, time_variant=True, iter=True, mode=None, timestep=-1)
        kind = "ddim_sample_loop_for_training"
    dt = time.time() - t0
    h1.remove()
    h2.remove()
    assert np.array_equal(rec["x_in"][0], x_T.numpy()), "x_T replication failed"
    out = {"grid": np.int64(grid), "steps": np.int64(steps), "n_hyp": np.int64(H), "kind": kind,
           "seed_w": np.int64(SEED_W), "seed_in": np.int64(SEED_IN),
           "x_T": x_T.numpy(), "t_model": np.asarray(rec["t"], dtype=np.float32),
           "x0_steps": np.stack(rec["x0"]), "x_in_steps": np.stack(rec["x_in"]),
           "sample": sample.numpy(), "ref_seconds": np.float64(dt)}
    name = f"loop_g{grid}_s{steps}.npz"
    if tame:
        keep = sorted(set(range(0, steps, 7)) | {steps - 1})
        last = rec["x0"][-1]
        out.update(out_gain=np.float64(synth.tame_gain(steps)), kept_steps=np.asarray(keep, dtype=np.int64),
                   x0_steps=np.stack([rec["x0"][i] for i in keep]),
                   last_x0_std=np.float64(last.std()), last_x0_saturated=np.float64((np.abs(last) >= 1).mean()))
        del out["x_in_steps"]
        name = f"loop_g{grid}_s{steps}_tame.npz"
    np.savez_compressed(os.path.join(GOLD, name), **out)
    print(f"{'G9' if tame else 'G3'} {name}  {kind}  {dt:.1f}s  t_model={rec['t'][:12]}  "
          f"sample mean {sample.mean():+.5f} std {sample.std():.5f}  last x0 std {rec['x0'][-1].std():.3f}")


# ------------------------------------------------------------------------------- G7
def gen_train_rollout(mods, grid, steps, timestep):
    """The training-time roll-out exactly as training_losses_time_variant calls it (gaussian_diffusion.py:921-946):
    ONE document, n_batch=1, mode='train' (the denoiser embeds the RAW model time: no 2/1 override,
    cross_model.py:575-580), roll-out from S-1 down to timestep+1, the caller's init_flow seen by the first step."""
    cross_model, gd, respace, script_util, dist_util, warping = mods
    model = build_model(cross_model, grid, script_util, dist_util)
    model.blocks = torch.nn.ModuleList([model.blocks[-1]])       # F2: bit-identical, 10x cheaper
    diff = make_diffusion(script_util, steps)
    inp = doc_inputs(grid)
    rec = {"t": [], "x0": [], "x_in": []}

    def pre(mod, args, kwargs):
        rec["x_in"].append(args[0].detach().clone().numpy())
        rec["t"].append(float(args[1][0]))

    def post(mod, args, kwargs, res):
        rec["x0"].append(res[0].detach().clone().numpy())
    h1 = model.register_forward_pre_hook(pre, with_kwargs=True)
    h2 = model.register_forward_hook(post, with_kwargs=True)
    init_flow = torch.from_numpy(synth.uniform("g7/init_flow", (1, 2, grid, grid), -0.3, 0.3, SEED_IN))
    kw = {"init_flow": init_flow.clone(), "y512": inp["y512"], "mask_cat": inp["mask_cat"],
          "init_feat": torch.zeros(1, 256, grid, grid), "mask_y512": inp["mask_y512"], "line_msk": inp["line_msk"]}
    seed = 977 + grid + steps
    torch.manual_seed(seed)
    _ = torch.randn(1, 2, grid, grid)
    x_T = torch.randn(1, 2, grid, grid)
    torch.manual_seed(seed)
    sample, feat = diff.ddim_sample_loop_for_training(
        model, (1, 2, grid, grid), noise=None, clip_denoised=False, model_kwargs=kw, eta=0.0,
        n_batch=1, time_variant=True, iter=True, mode="train", timestep=timestep)
    h1.remove()
    h2.remove()
    assert np.array_equal(rec["x_in"][0], x_T.numpy()), "x_T replication failed"
    assert len(rec["t"]) == steps - 1 - timestep
    out = {"grid": np.int64(grid), "steps": np.int64(steps), "timestep": np.int64(timestep),
           "x_T": x_T.numpy(), "init_flow": init_flow.numpy(), "t_model": np.asarray(rec["t"], dtype=np.float32),
           "x0_steps": np.stack(rec["x0"]), "x_in_steps": np.stack(rec["x_in"]), "sample": sample.numpy(),
           "feat_stats": summ(feat)}
    np.savez_compressed(os.path.join(GOLD, f"rollout_train_g{grid}_s{steps}.npz"), **out)
    print(f"G7 rollout_train_g{grid}_s{steps}.npz  timestep={timestep}  t_model={rec['t']}  "
          f"sample mean {sample.mean():+.5f} std {sample.std():.5f}")


# ------------------------------------------------------------------------------- G8
def gen_prestage(mods, grid=16):
    """evaluation.py:162-216 with the reference's own modules (val_TDiff.py:57-75 builds them): GeoTr_Seg_Inf (only its
    U2NETP `.msk` is loaded and only its mask output is read on the live configuration), Seg, UNet."""
    from train_settings.models.geotr import geotr_core
    from train_settings.models.geotr.unet_model import UNet
    sd_a = synth.synth_convnet_state_dict("u2netp", 11)                       # -> GeoTr_Seg_Inf.msk
    sd_b = synth.synth_convnet_state_dict("u2netp", 22, prefix="msk.")        # -> Seg
    sd_l = synth.synth_convnet_state_dict("unet", 13)
    tt = lambda sd: {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}  # noqa: E731
    msk_net = geotr_core.U2NETP(3, 1)
    assert list(msk_net.state_dict().keys()) == list(sd_a.keys())
    msk_net.load_state_dict(tt(sd_a), strict=True)
    seg = geotr_core.Seg()
    assert list(seg.state_dict().keys()) == list(sd_b.keys())
    seg.load_state_dict(tt(sd_b), strict=True)
    line = UNet(n_channels=3, n_classes=1)
    assert list(line.state_dict().keys()) == list(sd_l.keys())
    line.load_state_dict(tt(sd_l), strict=True)
    for m in (msk_net, seg, line):
        m.eval()
    src = torch.from_numpy(synth.smooth_image("g8/src", 512, 512, SEED_IN))[None]       # [1,3,512,512] 0..1
    with torch.no_grad():
        src288 = F.interpolate(src, size=(288), mode="bilinear", align_corners=True)                    # :162
        # GeoTr_Seg_Inf.forward (geotr_core.py:1003-1019) without its dead, never-loaded GeoTr branch
        msk, *_ = msk_net(src288)
        mask_x = F.interpolate(msk, size=(512), mode="bilinear", align_corners=True)
        mskx, d0, hx6, hx5d, hx4d, hx3d, hx2d, hx1d = seg(src288)                                        # :198
        feats = [F.interpolate(t, size=grid, mode="bilinear", align_corners=False)
                 for t in (hx6, hx5d, hx4d, hx3d, hx2d, hx1d)]                                           # :199-204
        seg_map_all = torch.cat(feats, dim=1)                                                           # :206
        textline_map, textline_mask = line(mskx)                                                        # :209
        line_small = F.interpolate(textline_map, size=grid, mode="bilinear", align_corners=False)        # :210
    near = float(((seg.msk(src288)[0] - 0.5).abs() < 1e-4).float().mean())
    out = {"grid": np.int64(grid), "mask_cat_sub": mask_x[0, 0, ::8, ::8].numpy().copy(), "mask_cat_stats": summ(mask_x),
           "d0_seg_sub": F.interpolate(d0, size=288, mode="bilinear", align_corners=True)[0, 0, ::4, ::4].numpy().copy(),
           "d0_512_stats": summ(d0), "mask_fraction": np.float64((mskx.abs().sum(1) > 0).float().mean()),
           "mask_near_threshold_fraction": np.float64(near),
           "mask_bits": np.packbits(((seg.msk(src288)[0] > 0.5)[0, 0]).numpy()),
           "mask_y512": seg_map_all[0].numpy().copy(), "line_msk": line_small[0].numpy().copy(),
           "hx1d_sub": hx1d[0, ::8, ::16, ::16].numpy().copy(), "hx6": hx6[0].numpy().copy(),
           "line_map_sub": textline_map[0, ::8, ::16, ::16].numpy().copy(),
           "line_logits_sub": textline_mask[0, 0, ::8, ::8].numpy().copy()}
    np.savez_compressed(os.path.join(GOLD, f"prestage_g{grid}.npz"), **out)
    print(f"G8 prestage_g{grid}.npz  mask fraction {out['mask_fraction']:.3f}  near-threshold {near:.2e}  "
          f"mask_y512 std {seg_map_all.std():.4f}  line std {line_small.std():.4f}")


# ------------------------------------------------------------------------------- G4
def gen_ddim_step(mods):
    _, gd, respace, script_util, _, _ = mods
    out = {}
    S = 50
    diff = make_diffusion(script_util, S)
    x_t = torch.from_numpy(synth.normalish("g4/x_t", (2, 2, 8, 8), SEED_IN))
    x0 = torch.from_numpy(synth.uniform("g4/x0", (2, 2, 8, 8), -1.0, 1.0, SEED_IN))
    out["x_t"], out["x0"] = x_t.numpy(), x0.numpy()
    fake = lambda x, t, **kw: x0  # noqa: E731
    samples = []
    for i in range(S):
        t = torch.tensor([i, i])
        torch.manual_seed(0)
        o = diff.ddim_sample(fake, x_t, t, clip_denoised=False, model_kwargs={}, eta=0.0)
        samples.append(o["sample"].numpy())
    out["ddim50/sample"] = np.stack(samples)
    # eta > 0 exercises sigma (the reference always calls with eta=0.0, evaluation.py:127)
    noise = torch.from_numpy(synth.normalish("g4/noise", (2, 2, 8, 8), SEED_IN))
    out["noise"] = noise.numpy()
    S2 = 250
    diff2 = make_diffusion(script_util, S2)
    means, logv = [], []
    for i in range(S2):
        t = torch.tensor([i, i])
        o = diff2.p_mean_variance(fake, x_t, t, clip_denoised=False, model_kwargs={})
        means.append(o["mean"].numpy())
        logv.append(float(o["log_variance"][0, 0, 0, 0]))
    out["ddpm250/mean"] = np.stack(means)
    out["ddpm250/log_variance"] = np.asarray(logv, dtype=np.float32)
    np.savez_compressed(os.path.join(GOLD, "ddim_step.npz"), **out)
    print("G4 ddim_step.npz")


# ------------------------------------------------------------------------------- G5
def gen_unwarp(mods):
    _, gd, _, _, _, warping = mods
    out = {}
    reg = warping.register_model2((512, 512), "bilinear")
    for tag, (H, W, G) in {"a": (97, 131, 16), "b": (64, 48, 8), "c": (33, 250, 32)}.items():
        flow = torch.from_numpy(synth.uniform(f"g5/{tag}/flow", (1, 2, G, G), -0.25, 0.25, SEED_IN))
        if tag == "a":      # push some samples well outside [-1,1] to hit zero padding
            flow[0, 0, :3, :] += 0.6
            flow[0, 1, -3:, :] -= 0.7
        if tag == "b":      # zero flow: identity grid scaled by 0.987 (exact-border behaviour)
            flow.zero_()
        src_u8 = synth.synth_document(0, 8, SEED_IN, full_res=(H, W))["src_u8"]
        src = torch.from_numpy(src_u8).permute(2, 0, 1)[None].float()          # [1,3,H,W] 0..255
        # evaluation.py:301-306
        s = F.interpolate(flow, size=(H, W), mode="bilinear", align_corners=True)
        base = F.interpolate(gd.coords_grid_tensor((512, 512)) / 511., size=(H, W), mode="bilinear",
                             align_corners=True)
        grid = (((s + base) * 1) * 2 - 1) * 0.987
        # visualization_utils.py:75-77
        warped = reg([src, grid])
        img = warped[0].permute(1, 2, 0).numpy()
        out[f"{tag}/flow"] = flow.numpy()
        out[f"{tag}/src_u8"] = src_u8
        out[f"{tag}/grid"] = grid.numpy()
        out[f"{tag}/out_f32"] = img.copy()
        out[f"{tag}/out_u8"] = img.astype(np.uint8)
    np.savez_compressed(os.path.join(GOLD, "unwarp.npz"), **out)
    print("G5 unwarp.npz")


# ------------------------------------------------------------------------------- G6
def gen_grid_sample(mods):
    _, gd, _, _, _, warping = mods
    reg = warping.register_model2((512, 512), "bilinear")
    G = 16
    feat = torch.from_numpy(synth.uniform("g6/feat", (2, 256, G, G), 0.0, 2.0, SEED_IN))
    x0 = torch.from_numpy(synth.uniform("g6/x0", (2, 2, G, G), -0.4, 0.4, SEED_IN))
    base = gd.coords_grid_tensor((G, G)) / float(G - 1)
    grid = (x0 + base) * 2 - 1                              # gaussian_diffusion.py:622
    out = reg([feat, grid])
    np.savez_compressed(os.path.join(GOLD, "grid_sample.npz"), x0=x0.numpy(), grid=grid.numpy(),
                        out=out.numpy(), base=base.numpy())
    print("G6 grid_sample.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="G1,G2,G3,G4,G5,G6,G7,G8,G9")
    args = ap.parse_args()
    want = set(args.only.split(","))
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    _workdir()
    mods = import_reference()
    if "G1" in want:
        gen_schedule(mods)
    if "G4" in want:
        gen_ddim_step(mods)
    if "G5" in want:
        gen_unwarp(mods)
    if "G6" in want:
        gen_grid_sample(mods)
    if "G2" in want:
        for g in (16, 32, 64):
            gen_forward(mods, g)
    if "G3" in want:
        gen_loop(mods, 16, 3, True)
        gen_loop(mods, 32, 3, True)
        gen_loop(mods, 64, 3, True)
        gen_loop(mods, 64, 10, False)
    if "G9" in want:                            # tame family (x0 inside (-1, 1)): un-flattered long loops from the REAL reference
        gen_loop(mods, 64, 10, False, tame=True)
        gen_loop(mods, 64, 50, False, tame=True)
        gen_loop(mods, 32, 50, False, tame=True)
    if "G8" in want:
        gen_prestage(mods, 16)
    if "G7" in want:
        gen_train_rollout(mods, 16, 10, 2)      # t_model 900 .. 300 raw: crosses both override thresholds
        gen_train_rollout(mods, 32, 3, -1)


if __name__ == "__main__":
    main()
