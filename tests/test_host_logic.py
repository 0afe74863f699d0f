"""CPU tests of the host-side logic: schedule tables vs the reference's golden tables, the timestep rules,
the weight packer <-> engine tensor contract (the engine's tensor table is host code and needs no GPU),
BatchNorm folding, f16 hi/lo weight splitting, the checkpoint-compatible module, and the plug-in surface."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from dvd_amd import lib, schedule, synth, weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("S", [3, 10, 50, 250])
def test_tables_bit_identical_to_reference(S):
    g = np.load(os.path.join(GOLD, "schedule.npz"))
    t = schedule.Tables(schedule.named_betas("cosine", S))
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                 "posterior_mean_coef1", "posterior_mean_coef2"):
        assert np.array_equal(getattr(t, name), g[f"s{S}/{name}"]), name
    assert list(t.timestep_map) == list(g[f"s{S}/timestep_map"])
    assert np.array_equal(np.float32([t.model_time(i) for i in range(S)]), g[f"s{S}/t_model_raw"])
    assert np.array_equal(t.fixed_large_log_variance.astype(np.float32), g[f"s{S}/fixed_large_logvar_f32"])


def test_timestep_rules():
    assert schedule.embedded_time(666.7) == 2.0 and schedule.embedded_time(600.0) == 600.0
    assert schedule.embedded_time(400.0) == 1.0 and schedule.embedded_time(300.0) == 300.0
    assert schedule.embedded_time(0.0) == 0.0
    from dvd_amd.sampler import feat_mode_for
    assert feat_mode_for(666.7, 2, True) == 1 and feat_mode_for(333.3, 2, False) == 2
    assert feat_mode_for(500.0, 2, True) == 0 and feat_mode_for(2.0, 2, False) == 1 and feat_mode_for(2.0, 1, False) == 2
    t = schedule.Tables(schedule.named_betas("cosine", 10))
    assert [schedule.embedded_time(t.model_time(i)) for i in range(9, -1, -1)] == \
        [2.0, 2.0, 2.0, 600.0, 1.0, 1.0, 300.0, 200.0, 100.0, 0.0]


def test_space_timesteps_contract():
    assert schedule.space_timesteps(10, [10]) == set(range(10))
    assert schedule.space_timesteps(300, "10,15,20") == schedule.space_timesteps(300, [10, 15, 20])
    assert len(schedule.space_timesteps(1000, "ddim50")) == 50
    with pytest.raises(ValueError):
        schedule.space_timesteps(10, [11])
    t = schedule.Tables(schedule.named_betas("linear", 100), schedule.space_timesteps(100, [10]))
    assert t.num_timesteps == 10 and t.original_num_steps == 100 and t.model_time(9) == 990.0


def test_ddim_coefficients_match_torch_reference_math():
    t = schedule.Tables(schedule.named_betas("cosine", 50))
    for i in (0, 1, 17, 49):
        c = t.ddim_coef(i)
        abp = np.float32(t.alphas_cumprod_prev[i])
        assert c.sqrt_abar_prev == np.sqrt(abp) and c.dir_coef == np.sqrt(np.float32(1) - abp)
        assert c.sigma == 0.0 and c.kind == 0
    assert t.ddpm_coef(0).sigma == 0.0 and t.ddpm_coef(5).sigma > 0


def engine_specs(grid):
    h = C.c_void_p()
    lib.call("dvd_engine_create", grid, 1, 2, C.byref(h))
    out = []
    name, dt, ne = C.c_char_p(), C.c_int(), C.c_long()
    for i in range(lib.raw().dvd_engine_tensor_count(h)):
        lib.call("dvd_engine_tensor_info", h, i, C.byref(name), C.byref(dt), C.byref(ne))
        out.append((name.value.decode(), dt.value, ne.value))
    ws = lib.raw().dvd_engine_workspace_bytes(h)
    lib.raw().dvd_engine_destroy(h)
    return out, ws


@pytest.mark.parametrize("grid", [16, 32])
def test_packer_matches_engine_tensor_table(grid):
    specs, ws = engine_specs(grid)
    assert ws > 0
    packed = weights.pack(synth.synth_state_dict(grid, 7, blocks=[11]), grid)
    assert set(packed) == {n for n, _, _ in specs}
    for name, dt, ne in specs:
        t = packed[name]
        assert t.dtype == (torch.float16 if dt == 1 else torch.float32), name
        assert t.numel() == ne and t.is_contiguous(), (name, t.numel(), ne)


def test_engine_state_errors_without_gpu():
    h = C.c_void_p()
    lib.call("dvd_engine_create", 16, 1, 2, C.byref(h))
    with pytest.raises(lib.DvdError, match="no workspace"):
        lib.call("dvd_engine_denoise_step", h, C.c_void_p(8), C.c_float(2.0), 1, C.c_void_p(8), None, C.c_void_p(8), None)
    with pytest.raises(lib.DvdError, match="unknown tensor"):
        lib.call("dvd_engine_set_tensor", h, b"nope", C.c_void_p(16), 4)
    with pytest.raises(lib.DvdError, match="expects"):
        lib.call("dvd_engine_set_tensor", h, b"obs_b", C.c_void_p(16), 5)
    with pytest.raises(lib.DvdError):
        lib.call("dvd_engine_create", 15, 1, 2, C.byref(C.c_void_p()))
    lib.raw().dvd_engine_destroy(h)


def test_bn_fold_and_split_weights():
    grid = 16
    sd = synth.synth_state_dict(grid, 7, blocks=[11])
    p = weights.pack(sd, grid)
    pre = "decoder.layer_stack.2.feed_forward.conv1."
    x = torch.from_numpy(synth.uniform("bn/x", (2, 1536, 4, 4), -1, 1, 3))
    conv = torch.nn.functional.conv2d(x, torch.from_numpy(sd[pre + "conv.weight"]))
    ref = torch.nn.functional.batch_norm(conv, torch.from_numpy(sd[pre + "bn.running_mean"]),
                                         torch.from_numpy(sd[pre + "bn.running_var"]),
                                         torch.from_numpy(sd[pre + "bn.weight"]), torch.from_numpy(sd[pre + "bn.bias"]),
                                         False, 0.0, 1e-5)
    w = p["d2_c1w16"].float() + p["d2_c1w16_lo"].float()
    got = torch.einsum("oc,nchw->nohw", w, x) + p["d2_c1b"][None, :, None, None]
    assert (got - ref).abs().max() < 2e-5
    # the split reconstructs the fp32 weight ~2^11 times better than a single f16
    hi_only = torch.einsum("oc,nchw->nohw", p["d2_c1w16"].float(), x) + p["d2_c1b"][None, :, None, None]
    assert (hi_only - ref).abs().max() > 20 * (got - ref).abs().max()


def test_patch_weight_layout():
    """K order of the patch-embed GEMM operand is (p*2+q)*C + c."""
    grid = 16
    sd = synth.synth_state_dict(grid, 7, blocks=[11])
    p = weights.pack(sd, grid)
    w = torch.from_numpy(sd["c_embedder.proj.weight"])       # [384,256,2,2]
    x = torch.from_numpy(synth.uniform("pw/x", (1, 256, 2, 2), -1, 1, 3))
    ref = torch.nn.functional.conv2d(x, w, stride=2).reshape(384)
    row = x[0].permute(1, 2, 0).reshape(-1)                   # (p, q, c)
    assert (p["c_w"] @ row - ref).abs().max() < 1e-5
    assert p["r_w16"].shape == (384, 1088) and float(p["r_w16"][:, 1032:].abs().max()) == 0.0


def test_denoiser_module_is_checkpoint_compatible():
    from dvd_amd.script_util import args_to_dict, create_model_and_diffusion, model_and_diffusion_defaults
    import admin.settings as ws
    s = ws.Settings()
    s.env.grid_size = 16
    model, diffusion = create_model_and_diffusion(device="cpu", train_mode=s.env.train_mode, tv=s.env.time_variant,
                                                  grid_size=16, **args_to_dict(s, model_and_diffusion_defaults().keys()))
    sd = model.state_dict()
    spec = synth.state_dict_spec(16)
    assert list(sd.keys()) == list(spec.keys()) and len(sd) == 369
    synth_sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state_dict(16, 7).items()}
    model.load_state_dict(synth_sd, strict=True)
    assert torch.equal(model.state_dict()["blocks.11.mlp.fc1.weight"], synth_sd["blocks.11.mlp.fc1.weight"])
    missing = dict(synth_sd)
    missing.pop("final_layer2.linear.bias")
    with pytest.raises(RuntimeError):
        model.load_state_dict(missing, strict=True)
    model.load_state_dict(missing, strict=False)             # what val_TDiff.run does
    assert diffusion.num_timesteps == 3 and diffusion.rescale_timesteps
    with pytest.raises(RuntimeError, match="HIP engine"):     # no CPU fallback
        model.engine(16, 1, 2)
    with pytest.raises(ValueError):
        create_model_and_diffusion(device="cpu", train_mode="stage_1", tv=True,
                                   **args_to_dict(s, model_and_diffusion_defaults().keys()))


def test_plugin_surface_importable():
    import importlib
    mod = importlib.import_module("train_settings.dvd.val_TDiff")
    assert callable(mod.run)
    from train_settings.dvd.improved_diffusion import dist_util, logger, script_util  # noqa: F401
    from train_settings.dvd.improved_diffusion.script_util import create_model_and_diffusion  # noqa: F401
    assert dist_util.shard_documents(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((dist_util.shard_documents(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_product_never_imports_oracle():
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    bad = []
    for d in ("dvd_amd", "admin", "train_settings"):
        for base, _, files in os.walk(os.path.join(root, d)):
            for f in files:
                if f.endswith(".py"):
                    txt = open(os.path.join(base, f)).read()
                    if "import oracle" in txt or "from oracle" in txt:
                        bad.append(os.path.join(base, f))
    assert not bad, bad
    assert "oracle" not in open(os.path.join(root, "run_sampling.py")).read()


def test_q_sample_matches_reference_formula():
    """q(x_t | x_0) = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps with float64 tables gathered to float32
    (idf/gaussian_diffusion.py:250-268, :1185-1195)."""
    import torch
    from dvd_amd import script_util
    diffusion = script_util.create_gaussian_diffusion(steps=10, noise_schedule="cosine", predict_xstart=True,
                                                      rescale_timesteps=True, timestep_respacing="")
    x0 = torch.from_numpy(synth.uniform("q/x0", (3, 2, 4, 4), -1, 1, 1))
    eps = torch.from_numpy(synth.uniform("q/eps", (3, 2, 4, 4), -2, 2, 1))
    t = torch.tensor([0, 4, 9])
    got = diffusion.q_sample(x0, t, noise=eps)
    ac = np.asarray(diffusion.alphas_cumprod, dtype=np.float64)
    a = torch.from_numpy(np.sqrt(ac))[t].float()[:, None, None, None]
    b = torch.from_numpy(np.sqrt(1 - ac))[t].float()[:, None, None, None]
    assert torch.equal(got, a * x0 + b * eps)


def test_reference_import_paths_of_the_prestage_nets():
    """val_TDiff.py:9-10 of the reference imports these names from these modules."""
    from train_settings.models.geotr.geotr_core import GeoTr_Seg_Inf, Seg, reload_segmodel  # noqa: F401
    from train_settings.models.geotr.unet_model import UNet
    assert UNet(n_channels=3, n_classes=1).kind == "unet" and Seg().msk.kind == "u2netp"


def test_tame_family_only_rescales_the_final_linear_layer():
    """synth.tame_gain(S) = 1.6 / S; out_gain touches final_layer2.linear.{weight,bias} and nothing else (so the golden
    vectors of the plain family and every other tensor of the tame family come from the same generator)."""
    assert synth.tame_gain(50) == 1.6 / 50 and synth.tame_gain(250) == 1.6 / 250
    a = synth.synth_state_dict(16, 7, blocks=[11])
    b = synth.synth_state_dict(16, 7, blocks=[11], out_gain=synth.tame_gain(50))
    changed = [k for k in a if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]))]
    assert sorted(changed) == ["final_layer2.linear.bias", "final_layer2.linear.weight"]
    np.testing.assert_allclose(b["final_layer2.linear.weight"], a["final_layer2.linear.weight"] * np.float32(0.032), rtol=1e-6)


def test_prepare_conditioning_needs_prestage_nets_only_for_documents_that_lack_tensors():
    """Round-2 ADVICE (medium): documents that already carry y512 / mask_cat / mask_y512 / line_msk (synthetic ones with
    env.use_prestage_nets=False, .npz documents) must pass through with NO pre-stage nets loaded; a document that lacks
    them must raise - host logic only, nothing here touches the GPU."""
    from dvd_amd import evaluation
    ready = {k: np.zeros((1, 2, 2), np.float32) for k in ("y512", "mask_cat", "mask_y512", "line_msk")}
    batch = [dict(ready), dict(ready)]
    evaluation.prepare_conditioning(batch, "cpu", 16, None)            # no nets, nothing to do: must not raise
    assert all(set(ready) <= set(d) for d in batch)
    lacking = [dict(ready), {"y512": np.zeros((3, 512, 512), np.float32)}]
    with pytest.raises(RuntimeError, match="pre-stage nets are not"):
        evaluation.prepare_conditioning(lacking, "cpu", 16, None)


def test_raise_together_single_rank():
    """dist_util.raise_together: with one rank it re-raises the rank's own exception and is a no-op otherwise (the 2-rank
    behaviour - every rank raises when rank 0 failed - is tests/test_distributed_cpu.py)."""
    from dvd_amd import dist_util
    dist_util.raise_together(None)
    with pytest.raises(FileNotFoundError):
        dist_util.raise_together(FileNotFoundError("x"))


def test_library_path_is_not_taken_from_the_environment(monkeypatch):
    """Round-2 VERDICT: DVD_HIP_LIB could put any .so under the product.  The binding now loads the in-tree build only;
    another build is an explicit lib.use_library() call in the caller's own code."""
    import importlib
    monkeypatch.setenv("DVD_HIP_LIB", "/nonexistent/libevil.so")
    mod = importlib.reload(lib)
    try:
        assert mod.LIB_PATH.endswith(os.path.join("dvd_amd", "libdvd_hip.so")) and mod.version() >= 1000
    finally:
        monkeypatch.delenv("DVD_HIP_LIB")
        importlib.reload(lib)


def test_single_step_calls_keep_the_reference_keyword_surface():
    """p_mean_variance / ddim_sample call model(x, t, **model_kwargs) like the reference
    (idf/gaussian_diffusion.py:327); the package's own `dither_step` keyword is injected only for a model that declares it -
    a lambda, a wrapper or a test double with the reference's keywords must not see it."""
    from dvd_amd import gaussian_diffusion as gd
    from dvd_amd.script_util import create_gaussian_diffusion

    diff = create_gaussian_diffusion(steps=10, noise_schedule="cosine", timestep_respacing="", predict_xstart=True)
    seen = {}

    def ref_like(x, t, init_flow=None, y512=None, **kw):       # the reference's surface: unknown keywords are the caller's
        seen.update(kw)
        return x, None

    class Own:
        def forward(self, x, t, init_flow=None, dither_step=None):
            return x, None

        __call__ = forward

    assert not gd._accepts_dither_step(lambda x, t, init_flow=None: (x, None))
    assert gd._accepts_dither_step(Own())
    assert diff._single_call_kwargs(lambda x, t, init_flow=None: (x, None), {"init_flow": 1}, 3) == {"init_flow": 1}
    assert diff._single_call_kwargs(Own(), {"init_flow": 1}, 3) == {"init_flow": 1, "dither_step": 6}
    assert diff._single_call_kwargs(Own(), {"dither_step": 2}, 3) == {"dither_step": 2}      # the caller's value wins
    assert "dither_step" not in diff._single_call_kwargs(ref_like, None, 0)
    from dvd_amd.cross_model import DvdDenoiser
    assert "dither_step" in __import__("inspect").signature(DvdDenoiser.forward).parameters


# ------------------------------------------------------------------------------------------------
# Round 6: the drop-in surface in the reference's call shapes: signatures, the time map of
# respace._WrappedModel, the loader item shapes, the dataset.  Host logic only - nothing here touches the GPU.
# ------------------------------------------------------------------------------------------------
def test_evaluation_entry_points_have_the_reference_positional_signatures():
    """evaluation.py:80-84 and :142-145 of the reference, name by name and in order: a reference-side caller passes these
    POSITIONALLY (evaluation.py:247-265, val_TDiff.py:103-104)."""
    import inspect
    from train_settings.dvd.evaluation import run_evaluation_docunet, run_sample_lr_dewarping
    names = lambda f: [(p.name, p.default is not inspect.Parameter.empty)  # noqa: E731
                       for p in inspect.signature(f).parameters.values() if p.kind is p.POSITIONAL_OR_KEYWORD]
    assert names(run_sample_lr_dewarping) == [(n, False) for n in (
        "settings", "logger", "diffusion", "model", "radius", "source", "feature_size", "raw_corr", "init_flow", "c20",
        "source_64", "pyramid", "doc_mask")] + [("seg_map_all", True), ("textline_map", True), ("init_feat", True)]
    assert names(run_evaluation_docunet) == [(n, False) for n in (
        "settings", "logger", "val_loader", "diffusion", "model", "pretrained_dewarp_model")] + \
        [("pretrained_line_seg_model", True), ("pretrained_seg_model", True)]
    from utils_flow.visualization_utils import visualize_dewarping
    assert [n for n, _ in names(visualize_dewarping)] == ["settings", "sample", "data", "i", "source_vis", "data_path", "ref_flow"]


def test_run_sample_lr_dewarping_builds_the_reference_kwargs_and_call():
    """The model_kwargs dict and the ddim_sample_loop call of evaluation.py:106-135, observed on a recording diffusion."""
    import torch
    import admin.settings as ws
    from dvd_amd import evaluation
    s = ws.Settings()
    seen = {}

    class Diff:
        def ddim_sample_loop(self, model, shape, **kw):
            seen.update(kw, model=model, shape=shape)
            return torch.full(shape, 3.0), {}

    class Log:
        def info(self, *a):
            pass
    src, flow0, msk = torch.zeros(1, 3, 512, 512), torch.zeros(1, 2, 64, 64), torch.ones(1, 1, 512, 512)
    seg, line, feat0 = torch.zeros(1, 384, 64, 64), torch.zeros(1, 64, 64, 64), torch.zeros(1, 256, 64, 64)
    out = evaluation.run_sample_lr_dewarping(s, Log(), Diff(), "MODEL", 4, src, 64, None, flow0, None, None, "PYR", msk,
                                             seg, line, feat0)
    assert float(out.max()) == 1.0                                        # th.clamp(sample, -1, 1) (:137)
    assert seen["model"] == "MODEL" and seen["shape"] == (1, 2, 64, 64) and seen["pyramid"] == "PYR"
    assert seen["noise"] is None and seen["clip_denoised"] is False and seen["eta"] == 0.0 and seen["n_batch"] == 2
    assert seen["time_variant"] is True and seen["sampling_kwargs"]["src_img"] is src and "sampler_kind" not in seen
    kw = seen["model_kwargs"]
    assert set(kw) == {"init_flow", "src_feat", "src_64", "y512", "tmode", "mask_cat", "init_feat", "iter", "mask_y512", "line_msk"}
    assert kw["y512"] is src and kw["mask_cat"] is msk and kw["mask_y512"] is seg and kw["line_msk"] is line
    assert kw["init_flow"] is flow0 and kw["init_feat"] is feat0 and kw["src_feat"] is None and kw["iter"] is True


@pytest.mark.parametrize("steps,respacing", [(3, ""), (10, ""), (50, ""), (250, ""), (1000, "ddim50"), (1000, "10,15,25")])
def test_wrapped_model_time_map_equals_the_tables(steps, respacing):
    """respace._WrappedModel (idf/respace.py:111-123): index tensor -> timestep_map -> float32 * (1000 / original steps),
    equal BITWISE to Tables.model_time(i) (what the sampling loop feeds the engine) for every step; the golden G1
    t_model sequence of S = 10 pins the form against the real reference."""
    import torch
    from dvd_amd import respace, script_util
    d = script_util.create_gaussian_diffusion(steps=steps, noise_schedule="cosine", predict_xstart=True,
                                              rescale_timesteps=True, timestep_respacing=respacing)
    assert isinstance(d, respace.SpacedDiffusion)
    got = []
    inner = lambda x, ts, **kw: got.append((ts.clone(), kw)) or "OUT"  # noqa: E731
    w = d._wrap_model(inner)
    assert isinstance(w, respace._WrappedModel) and d._wrap_model(w) is w and w.model is inner
    assert w.original_num_steps == steps and w.rescale_timesteps and list(w.timestep_map) == list(d.timestep_map)
    for i in range(d.num_timesteps):
        assert w(None, torch.tensor([i, i]), a=1) == "OUT"
        ts, kw = got[-1]
        assert ts.dtype == torch.float32 and kw == {"a": 1}
        assert ts.tolist() == [d.tables.model_time(i)] * 2, (i, ts, d.tables.model_time(i))
    if (steps, respacing) == (10, ""):           # golden G1 (the real reference's wrapped-model times, index order)
        sch = np.load(os.path.join(os.path.dirname(__file__), "golden", "schedule.npz"))
        assert np.array_equal(np.array([float(t[0][0]) for t in got], np.float32), sch["s10/t_model_raw"])


def test_dither_keyword_reaches_the_denoiser_through_pass_through_wrappers():
    """Round-5 ADVICE: the loop's dithering phase must reach a DvdDenoiser behind ANY pass-through wrapper (DDP's .module,
    a user wrapper forwarding **kwargs, respace._WrappedModel), and a callable with the reference's keyword surface must
    not be handed an unknown keyword."""
    from dvd_amd import respace
    from dvd_amd.gaussian_diffusion import _accepts_dither_step

    class Den:                                   # stands for DvdDenoiser: names the keyword
        def forward(self, x, t, init_flow=None, dither_step=None):
            return x

        __call__ = forward

    class DDPLike:
        def __init__(self, m):
            self.module = m

        def forward(self, *a, **kw):
            return self.module(*a, **kw)

        __call__ = forward

    class UserWrap:
        def __init__(self, m):
            self.model = m

        def __call__(self, x, t, **kw):
            return self.model(x, t, **kw)

    class Blocking:                              # forwards a FIXED keyword set: the phase cannot pass
        def __init__(self, m):
            self.model = m

        def __call__(self, x, t, init_flow=None):
            return self.model(x, t, init_flow=init_flow)
    den = Den()
    assert _accepts_dither_step(den)
    assert _accepts_dither_step(DDPLike(den)) and _accepts_dither_step(UserWrap(DDPLike(den)))
    assert _accepts_dither_step(respace._WrappedModel(den, [0, 1, 2], True, 3))
    assert not _accepts_dither_step(Blocking(den))
    assert not _accepts_dither_step(lambda x, t, init_flow=None: x)
    assert not _accepts_dither_step(lambda x, t, **kw: x)        # forwards to nothing that names it
    w = UserWrap(den)
    assert _accepts_dither_step(w) and _accepts_dither_step(w)   # second call: the per-object cache


def test_doc_benchmark_dataset_and_loader_item_shapes(tmp_path):
    """datasets.Doc_benchmark (doc_benchmark.py:49-97) behind the reference's DataLoader(batch_size=1): the reference's
    keys and layouts, files in sorted order, EXIF-free PNG decode exact; evaluation.documents_of splits both item shapes."""
    import torch
    from PIL import Image
    from torch.utils.data import DataLoader
    import datasets
    from dvd_amd import evaluation
    from utils_data.image_transforms import ArrayToTensor
    rng = np.random.RandomState(0)
    imgs = {}
    for name, hw in (("b_page.png", (40, 30)), ("a_page.PNG", (24, 56)), ("notes.txt", None)):
        if hw is None:
            (tmp_path / name).write_text("not an image")
            continue
        imgs[name] = rng.randint(0, 256, size=hw + (3,)).astype(np.uint8)
        Image.fromarray(imgs[name]).save(tmp_path / name, format="PNG")
    ds = datasets.Doc_benchmark(str(tmp_path), ArrayToTensor(get_float=True))
    assert len(ds) == 2 and ds.data_paths == ["a_page.PNG", "b_page.png"]
    items = list(DataLoader(ds, batch_size=1, shuffle=False, num_workers=0))
    for item, name in zip(items, ds.data_paths):
        assert set(item) == {"source_image_ori", "path"} and item["path"] == [os.path.join(str(tmp_path), name)]
        ori = item["source_image_ori"]
        assert ori.dtype == torch.float32 and tuple(ori.shape) == (1, 3) + imgs[name].shape[:2]
        assert np.array_equal(ori[0].permute(1, 2, 0).numpy().astype(np.uint8), imgs[name])
        (d,) = evaluation.documents_of(item)
        assert d["path"] == item["path"][0] and d["source_vis"].shape == ori.shape[1:] and "y512" not in d
        u8 = evaluation._source_u8(d, "cpu")
        assert u8.dtype == torch.uint8 and np.array_equal(u8.numpy(), imgs[name])
    # a reference loader item (cv2 dataset: source_image present), b = 2; and one of this package's documents
    item = {"source_image": torch.rand(2, 3, 512, 512), "source_image_ori": torch.zeros(2, 3, 8, 6), "path": ["x/p.jpg", "x/q.jpg"]}
    docs = evaluation.documents_of(item)
    assert [d["path"] for d in docs] == ["x/p.jpg", "x/q.jpg"] and torch.equal(docs[1]["y512"], item["source_image"][1])
    own = {"image_u8": np.zeros((4, 4, 3), np.uint8), "path": "stem"}
    assert evaluation.documents_of(own) == [own]
    # a float source that is not a byte image cannot become u8 exactly
    assert evaluation._source_u8({"source_vis": torch.full((3, 2, 2), 7.5)}, "cpu") is None
    assert ArrayToTensor(get_float=False)(imgs["b_page.png"]).dtype == torch.uint8
