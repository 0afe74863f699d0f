"""Pins the CPU oracle (oracle/dvd_oracle.py) to the golden vectors produced by the REAL
reference (oracle/ref_harness/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from dvd_amd import synth
from oracle import dvd_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


# ----------------------------------------------------------------------------- G1
@pytest.mark.parametrize("S", [3, 10, 50, 250])
def test_schedule_tables(S):
    g = load("schedule.npz")
    sch = O.Schedule(S)
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                 "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_allclose(getattr(sch, name), g[f"s{S}/{name}"], rtol=1e-13, atol=0, err_msg=name)
    assert np.array_equal(sch.timestep_map, g[f"s{S}/timestep_map"])
    tm = np.array([sch.t_model(i) for i in range(S)], dtype=np.float32)
    assert np.array_equal(tm, g[f"s{S}/t_model_raw"])
    assert np.array_equal(sch.fixed_large_variance.astype(np.float32), g[f"s{S}/fixed_large_var_f32"])
    assert np.array_equal(sch.fixed_large_log_variance.astype(np.float32), g[f"s{S}/fixed_large_logvar_f32"])


def test_t_rule_edges():
    # cross_model.py:575-580: strictly >600 -> 2, strictly between -> 1, 600/300/<=300 raw
    assert O.t_rule(600.0001) == 2.0 and O.t_rule(600.0) == 600.0
    assert O.t_rule(599.9) == 1.0 and O.t_rule(300.0) == 300.0 and O.t_rule(300.1) == 1.0
    assert O.t_rule(0.0) == 0.0 and O.t_rule(200.0) == 200.0


def test_t_model_sequence_s10():
    sch = O.Schedule(10)
    seq = [O.t_rule(float(sch.t_model(i))) for i in range(9, -1, -1)]
    assert seq == [2.0, 2.0, 2.0, 600.0, 1.0, 1.0, 300.0, 200.0, 100.0, 0.0]


# ----------------------------------------------------------------------------- G4
def test_ddim_step_all_t():
    g = load("ddim_step.npz")
    sch = O.Schedule(50)
    x_t, x0 = torch.from_numpy(g["x_t"]), torch.from_numpy(g["x0"])
    for i in range(50):
        got = O.ddim_step(sch, i, x_t, x0).numpy()
        assert np.array_equal(got, g["ddim50/sample"][i]), f"t={i} not bit-identical to the reference"


def test_ddpm_mean_logvar():
    g = load("ddim_step.npz")
    sch = O.Schedule(250)
    x_t, x0 = torch.from_numpy(g["x_t"]), torch.from_numpy(g["x0"])
    for i in range(250):
        mean, lv = O.ddpm_mean_logvar(sch, i, x_t, x0)
        assert np.array_equal(mean.numpy(), g["ddpm250/mean"][i]), i
        assert np.float32(lv.reshape(-1)[0]) == g["ddpm250/log_variance"][i]


# ----------------------------------------------------------------------------- G5 / G6
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_unwarp_tail(tag):
    g = load("unwarp.npz")
    flow = torch.from_numpy(g[f"{tag}/flow"])
    src = torch.from_numpy(g[f"{tag}/src_u8"]).permute(2, 0, 1)[None].float()
    grid, out, out_u8 = O.unwarp_tail(flow, src)
    np.testing.assert_allclose(grid.numpy(), g[f"{tag}/grid"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out.numpy(), g[f"{tag}/out_f32"], rtol=0, atol=2e-3)
    assert (np.abs(out_u8.astype(int) - g[f"{tag}/out_u8"].astype(int)) <= 1).all()
    assert (out_u8 != g[f"{tag}/out_u8"]).mean() < 1e-3


def test_grid_sample_dropin():
    g = load("grid_sample.npz")
    feat = torch.from_numpy(synth.uniform("g6/feat", (2, 256, 16, 16), 0.0, 2.0, 1234))
    x0 = torch.from_numpy(g["x0"])
    base = O.base_grid(16, 16)
    np.testing.assert_allclose(base.numpy(), g["base"], rtol=0, atol=1e-7)
    grid = (x0 + base) * 2 - 1
    np.testing.assert_allclose(O.grid_sample_ref(feat, grid).numpy(), g["out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.grid_sample_manual(feat, grid).numpy(), g["out"], rtol=0, atol=2e-5)


# ----------------------------------------------------------------------------- G2
def _doc(grid):
    d = synth.synth_document(0, grid, 1234)
    return {k: torch.from_numpy(v)[None] for k, v in d.items()}


@pytest.mark.parametrize("grid", [16, 32, 64])
def test_forward_vs_reference(grid):
    g = load(f"forward_g{grid}.npz")
    sd = synth.synth_state_dict(grid, int(g["seed_w"]), blocks=[11])
    orc = O.Oracle(sd, grid)
    doc = _doc(grid)
    inv1 = orc.prepare(doc["y512"], doc["mask_cat"], doc["mask_y512"], doc["line_msk"])
    inv = {k: v.repeat(2, *([1] * (v.dim() - 1))) for k, v in inv1.items()}
    x = torch.from_numpy(g["x"])
    init_flow = torch.from_numpy(g["init_flow"])
    init_feat = torch.from_numpy(synth.uniform("g2/init_feat", (2, 256, grid, grid), 0.0, 1.5, 1234))
    np.testing.assert_allclose(inv1["feat"][0, ::16, ::max(1, grid // 16), ::max(1, grid // 16)].numpy(),
                               g["feat/sub"], rtol=0, atol=2e-5)
    for tag in ("t2", "t1", "raw0", "raw600", "raw200"):
        ck = {}
        x0, _ = orc.forward(x, float(g[f"{tag}/t_in"]), inv, init_flow, init_feat, ck=ck)
        for name, v in ck.items():
            key = f"{tag}/ck/{name}/head"
            if key in g.files:
                scale = max(1.0, float(g[f"{tag}/ck/{name}/stats"][2]))
                np.testing.assert_allclose(v.reshape(-1)[:64].numpy(), g[key], rtol=0, atol=2e-5 * scale,
                                           err_msg=f"{tag}/{name}")
                np.testing.assert_allclose(v.reshape(-1)[-64:].numpy(), g[f"{tag}/ck/{name}/tail"], rtol=0,
                                           atol=2e-5 * scale, err_msg=f"{tag}/{name} tail")
        err = np.abs(x0.numpy() - g[f"{tag}/x0"]).max()
        assert err < 2e-5, (tag, err)


# ----------------------------------------------------------------------------- G3
@pytest.mark.parametrize("grid,steps", [(16, 3), (32, 3), (64, 3)])
def test_loop_vs_reference(grid, steps):
    g = load(f"loop_g{grid}_s{steps}.npz")
    sd = synth.synth_state_dict(grid, int(g["seed_w"]), blocks=[11])
    orc = O.Oracle(sd, grid)
    sch = O.Schedule(steps)
    trace = []
    out = orc.sample_loop(sch, torch.from_numpy(g["x_T"]), _doc(grid), mean_hyp=(grid == 64), trace=trace)
    for k, x0 in enumerate(trace):
        err = np.abs(x0.numpy() - g["x0_steps"][k]).max()
        assert err < 5e-5, (k, err)
    np.testing.assert_allclose(out.numpy(), g["sample"], rtol=0, atol=5e-5)


# ----------------------------------------------------------------------------- G7
@pytest.mark.parametrize("grid,steps", [(16, 10), (32, 3)])
def test_training_rollout_mode_train_vs_reference(grid, steps):
    """ddim_sample_loop_for_training(mode='train', n_batch=1, timestep=k) of the real reference
    (gaussian_diffusion.py:694-782,921-946): raw model time embedded (no override), the caller's init_flow at the
    first step, roll-out stops at timestep+1, clamp only."""
    g = load(f"rollout_train_g{grid}_s{steps}.npz")
    sd = synth.synth_state_dict(grid, 7, blocks=[11])
    orc = O.Oracle(sd, grid)
    sch = O.Schedule(steps)
    assert [float(sch.t_model(i)) for i in range(steps - 1, int(g["timestep"]), -1)] == [float(t) for t in g["t_model"]]
    trace = []
    out = orc.sample_loop(sch, torch.from_numpy(g["x_T"]), _doc(grid), mean_hyp=False, trace=trace,
                          init_flow=torch.from_numpy(g["init_flow"]), last_step=int(g["timestep"]) + 1, mode="train")
    assert len(trace) == len(g["x0_steps"])
    for k, x0 in enumerate(trace):
        err = np.abs(x0.numpy() - g["x0_steps"][k]).max()
        assert err < 5e-5, (k, err)
    np.testing.assert_allclose(out.numpy(), g["sample"], rtol=0, atol=5e-5)


@pytest.mark.slow
def test_loop_s10_vs_reference():
    g = load("loop_g64_s10.npz")
    sd = synth.synth_state_dict(64, int(g["seed_w"]), blocks=[11])
    orc = O.Oracle(sd, 64)
    sch = O.Schedule(10)
    assert [O.t_rule(float(t)) for t in g["t_model"]] == [2.0, 2.0, 2.0, 600.0, 1.0, 1.0, 300.0, 200.0, 100.0, 0.0]
    out = orc.sample_loop(sch, torch.from_numpy(g["x_T"]), _doc(64))
    np.testing.assert_allclose(out.numpy(), g["sample"], rtol=0, atol=1e-4)


# ----------------------------------------------------------------------------- G9
@pytest.mark.parametrize("grid,steps", [(64, 10), (32, 50)])
def test_tame_family_loop_vs_reference(grid, steps):
    """The TAME weight family (final linear layer scaled by 1.6 / S: x0 stays inside (-1, 1), nothing hides behind the
    final clamp) against the REAL reference's roll-out: the un-clamped x0 of every kept step and the final map.  (The
    50-step G = 64 golden is checked on the GPU only: the oracle needs 3 s per step there.)"""
    g = load(f"loop_g{grid}_s{steps}_tame.npz")
    assert float(g["last_x0_saturated"]) < 0.01 and 0.2 < float(g["last_x0_std"]) < 0.6
    sd = synth.synth_state_dict(grid, int(g["seed_w"]), blocks=[11], out_gain=float(g["out_gain"]))
    assert float(g["out_gain"]) == synth.tame_gain(steps)
    orc = O.Oracle(sd, grid)
    trace = []
    out = orc.sample_loop(O.Schedule(steps), torch.from_numpy(g["x_T"]), _doc(grid), mean_hyp=(grid == 64), trace=trace)
    for k, i in enumerate(g["kept_steps"]):
        err = np.abs(trace[int(i)].numpy() - g["x0_steps"][k]).max()
        assert err < 2e-5, (int(i), err)
    np.testing.assert_allclose(out.numpy(), g["sample"], rtol=0, atol=2e-5)


# ----------------------------------------------------------------------------- G8
def test_prestage_oracle_vs_reference():
    """The pre-stage nets' restatement (oracle/prestage_oracle.py) against the reference's own U2NETP / Seg / UNet
    modules and the glue of evaluation.py:162-216 (golden G8)."""
    import torch.nn.functional as F
    from oracle import prestage_oracle as PO
    g = load("prestage_g16.npz")
    grid = int(g["grid"])
    sd_a = synth.synth_convnet_state_dict("u2netp", 11)
    sd_b = synth.synth_convnet_state_dict("u2netp", 22, prefix="msk.")
    sd_l = synth.synth_convnet_state_dict("unet", 13)
    src = torch.from_numpy(synth.smooth_image("g8/src", 512, 512, 1234))[None]
    with torch.no_grad():
        out = PO.prestage(sd_a, sd_b, sd_l, src, grid)
    assert np.array_equal(np.packbits((out["d0"] > 0.5)[0, 0].numpy()), g["mask_bits"])
    np.testing.assert_allclose(out["mask_cat"][0, 0, ::8, ::8].numpy(), g["mask_cat_sub"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["hx"][0][0].numpy(), g["hx6"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["hx"][5][0, ::8, ::16, ::16].numpy(), g["hx1d_sub"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["mask_y512"][0].numpy(), g["mask_y512"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["line_map"][0, ::8, ::16, ::16].numpy(), g["line_map_sub"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["line_msk"][0].numpy(), g["line_msk"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("shape", [(16, 97, 131), (32, 240, 172), (64, 301, 415)])
def test_aten_order_restatement(shape):
    """oracle/aten_order.py - the arithmetic ORDER the HIP warps implement (FMA contractions of the torch-CPU kernels included)
    - against the torch build of THIS box, bit for bit: grid, f32 image and truncated bytes of the whole tail, and the golden
    G5 vectors of the real reference."""
    from oracle import aten_order as A
    g, H, W = shape
    rng = np.random.RandomState(g)
    flow = (rng.randn(1, 2, g, g) * 0.06).astype(np.float32)
    src = (rng.rand(1, 3, H, W) * 255).astype(np.float32)
    grid, out, u8 = O.unwarp_tail(torch.from_numpy(flow), torch.from_numpy(src))
    grid2, out2, u82 = A.unwarp_tail(flow, src)
    assert np.array_equal(grid.numpy(), grid2) and np.array_equal(out.numpy(), out2) and np.array_equal(u8, u82)
    gold = load("unwarp.npz")
    for tag in ("a", "b", "c"):
        s8 = gold[f"{tag}/src_u8"]
        g3, o3, u3 = A.unwarp_tail(gold[f"{tag}/flow"], s8.transpose(2, 0, 1)[None].astype(np.float32))
        assert np.array_equal(g3.reshape(gold[f"{tag}/grid"].shape), gold[f"{tag}/grid"])
        assert np.array_equal(o3, gold[f"{tag}/out_f32"]) and np.array_equal(u3, gold[f"{tag}/out_u8"])
