"""GPU parity of the HIP denoiser engine against the CPU oracle (same synthetic weights and inputs)
and against the golden vectors made from the real reference.  Tolerances: the path computes its
per-step GEMMs and attention with f16 operands / fp32 accumulation; north_star's bar is a coordinate
L2 (RMSE over the [2,G,G] map) below 1e-3."""
import os

import numpy as np
import pytest
import torch

from dvd_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
SEED_W, SEED_IN = 7, 1234
_cache = {}


def setup(grid, docs=1, hyp=2, out_gain=1.0):
    """Engine + oracle on the same synthetic weights and documents.  out_gain = 1 is the family of the golden vectors;
    synth.tame_gain(S) the TAME family whose S-step roll-out stays inside (-1, 1), so the long-loop tests can assert on
    the un-clamped, un-averaged x0 of the last step (with the plain family most of its pixels saturate the clamp)."""
    key = (grid, docs, hyp, out_gain)
    if key in _cache:
        return _cache[key]
    from dvd_amd.engine import Engine
    from oracle import dvd_oracle as O
    sd = synth.synth_state_dict(grid, SEED_W, blocks=[11], out_gain=out_gain)
    eng = Engine(grid, docs, hyp)
    eng.load_state_dict(sd)
    orc = O.Oracle(sd, grid)
    ds = [synth.synth_document(d, grid, SEED_IN) for d in range(docs)]
    doc_t = {k: torch.from_numpy(np.stack([d[k] for d in ds])) for k in ("y512", "mask_cat", "mask_y512", "line_msk")}
    eng.prepare(*[doc_t[k].cuda() for k in ("y512", "mask_cat", "mask_y512", "line_msk")])
    inv = [orc.prepare(*[doc_t[k][d:d + 1] for k in ("y512", "mask_cat", "mask_y512", "line_msk")]) for d in range(docs)]
    _cache[key] = (eng, orc, doc_t, inv)
    return _cache[key]


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("grid", [16, 32, 64])
def test_prepare_docs(grid):
    eng, orc, doc_t, inv = setup(grid)
    feat = eng.feat_nchw().cpu()
    assert rel(feat, inv[0]["feat"]) < 2e-5, rel(feat, inv[0]["feat"])
    T = (grid // 2) ** 2
    wi, bi = orc.W("blocks.11.cross_attn.in_proj_weight"), orc.W("blocks.11.cross_attn.in_proj_bias")
    for name, key in (("kc16", "cond"), ("km16", "msk6"), ("kl16", "line")):
        k_ref = torch.nn.functional.linear(inv[0][key], wi[384:768], bi[384:768])[0]
        got = eng.debug(name, torch.float16, (T, 384)).float().cpu()
        assert rel(got, k_ref) < 2e-3, (name, rel(got, k_ref))
    for name, key in (("vtc16", "cond"), ("vtm16", "msk6"), ("vtl16", "line")):
        v_ref = torch.nn.functional.linear(inv[0][key], wi[768:], bi[768:])[0].t()
        got = eng.debug(name, torch.float16, (384, T)).float().cpu()
        assert rel(got, v_ref) < 2e-3, (name, rel(got, v_ref))


def _inputs(grid, n):
    x = torch.from_numpy(synth.synth_noise(0, n, grid, SEED_IN))
    flow = torch.from_numpy(synth.uniform("g2/init_flow", (n, 2, grid, grid), -0.3, 0.3, SEED_IN))
    return x, flow


@pytest.mark.parametrize("grid", [16, 32])
@pytest.mark.parametrize("tcase", [(666.6667, 1), (400.0, 2), (0.0, 2)])
def test_forward_stages_vs_oracle(grid, tcase):
    from dvd_amd import schedule
    from oracle import dvd_oracle as O
    t_model, mode = tcase
    eng, orc, doc_t, inv1 = setup(grid)
    n, T = 2, (grid // 2) ** 2
    inv = {k: v.repeat(n, *([1] * (v.dim() - 1))) for k, v in inv1[0].items()}
    x, flow = _inputs(grid, n)
    init_feat = O.grid_sample_ref(inv["feat"], (flow + O.base_grid(grid, grid)) * 2 - 1)
    ck = {}
    x0_ref, _ = orc.forward(x, t_model, inv, flow, init_feat, ck=ck)
    errs = {}
    try:
        for stage, name in ((2, "blk"), (3, "dec_pos"), (4, "dec0"), (6, "dec2"), (9, "dec5")):
            eng.debug_stop(stage)
            eng.denoise(x.cuda(), schedule.embedded_time(t_model), mode, flow.cuda())
            z = eng.debug("z", torch.float32, (n, T, 1536)).cpu()
            ref = torch.cat([ck[f"blk_x{i}"] for i in (1, 2, 3, 4)], dim=2) if name == "blk" else ck[name]
            errs[name] = rel(z, ref)
    finally:
        eng.debug_stop(0)
    x0 = eng.denoise(x.cuda(), schedule.embedded_time(t_model), mode, flow.cuda()).cpu()
    errs["x0_max"] = (x0 - x0_ref).abs().max().item()
    errs["x0_rmse"] = (x0 - x0_ref).pow(2).mean().sqrt().item()
    print("stage errors", grid, tcase, errs)
    # bars = 3x the values measured on MI355X (blk 1.3e-4, dec_pos 1.2e-4, dec5 2.7e-4, x0 rmse 5.4e-5): a precision
    # regression of more than that fails here long before north_star's 1e-3 coordinate bar is in danger
    assert errs["blk"] < 4e-4 and errs["dec_pos"] < 4e-4 and errs["dec5"] < 8e-4, errs
    assert errs["x0_rmse"] < 1.6e-4, errs


@pytest.mark.parametrize("grid", [16, 32, 64])
def test_forward_vs_reference_golden(grid):
    """x0 of one denoiser call against the REAL reference's output (golden G2)."""
    from dvd_amd import schedule
    g = np.load(os.path.join(GOLD, f"forward_g{grid}.npz"))
    eng, orc, doc_t, inv1 = setup(grid)
    x, flow = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["init_flow"]).cuda()
    # the golden call passed an arbitrary init_feat; only the t > 600 case (init_feat <- feat) is engine-reachable
    x0 = eng.denoise(x, schedule.embedded_time(float(g["t2/t_in"])), 1, flow).cpu().numpy()
    err = np.sqrt(((x0 - g["t2/x0"]) ** 2).mean())
    print("golden forward rmse", grid, err, np.abs(x0 - g["t2/x0"]).max())
    assert err < 1.6e-4, err          # measured 5.3e-5 (x3); north_star's bar is 1e-3


@pytest.mark.parametrize("grid,steps", [(16, 3), (32, 3), (64, 3), (64, 10)])
def test_sampling_loop_vs_reference_golden(grid, steps):
    """Whole DDIM loop against the REAL reference (golden G3): coordinate L2 < 1e-3."""
    from dvd_amd import sampler, schedule
    g = np.load(os.path.join(GOLD, f"loop_g{grid}_s{steps}.npz"))
    eng, orc, doc_t, inv1 = setup(grid)
    tab = schedule.Tables(schedule.named_betas("cosine", steps))
    trace = []
    out = sampler.sample(eng, tab, torch.from_numpy(g["x_T"]).cuda(), mean_hyp=(grid == 64), trace=trace)
    per_step = [float(np.sqrt(((t.cpu().numpy() - g["x0_steps"][k]) ** 2).mean())) for k, t in enumerate(trace)]
    err = float(np.sqrt(((out.cpu().numpy() - g["sample"]) ** 2).mean()))
    print("loop rmse", grid, steps, err, "per-step", per_step)
    assert err < 2.7e-4, (err, per_step)   # measured 6.5e-5 .. 9.0e-5 (x3); north_star's bar is 1e-3


@pytest.mark.parametrize("grid,steps", [(64, 10), (64, 50), (32, 50)])
def test_tame_family_loop_vs_reference_golden(grid, steps):
    """Golden G9: the REAL reference's roll-out on the tame weight family (x0 inside (-1, 1)): the engine's UN-CLAMPED x0
    at every kept step (incl. the last of a 50-step loop) and the final map."""
    from dvd_amd import sampler, schedule
    g = np.load(os.path.join(GOLD, f"loop_g{grid}_s{steps}_tame.npz"))
    eng, orc, doc_t, inv1 = setup(grid, 1, 2, float(g["out_gain"]))
    tab = schedule.Tables(schedule.named_betas("cosine", steps))
    trace = []
    out = sampler.sample(eng, tab, torch.from_numpy(g["x_T"]).cuda(), mean_hyp=(grid == 64), trace=trace)
    per = {int(i): float(np.sqrt(((trace[int(i)].cpu().numpy() - g["x0_steps"][k]) ** 2).mean()))
           for k, i in enumerate(g["kept_steps"])}
    err = float(np.sqrt(((out.cpu().numpy() - g["sample"]) ** 2).mean()))
    print(f"tame-family loop vs the real reference G={grid} S={steps}: final {err:.2e}, un-clamped x0 per kept step {per}, "
          f"last x0 std {float(g['last_x0_std']):.3f}, saturated {float(g['last_x0_saturated']):.4f}")
    # measured on MI355X: last un-clamped x0 2.8e-5 (S=10, G=64), 1.28e-5 (S=50, G=64), 1.31e-5 (S=50, G=32); bar ~3x
    assert per[steps - 1] < 9e-5 and err < 9e-5, (err, per)     # north_star's bar is 1e-3


def _saturated(x):
    return float((x.abs() >= 1).float().mean())


@pytest.mark.parametrize("grid,steps,hyp,family", [(16, 50, 2, "plain"), (16, 50, 2, "tame"), (32, 50, 2, "tame"),
                                                   (96, 50, 1, "tame"), (72, 25, 1, "tame")])
def test_long_loop_vs_oracle(grid, steps, hyp, family):
    """BASELINE's 50-step DDIM (not reference-runnable natively: local.py has 3 steps) against the CPU oracle.
    G = 96 is an UP-sampling, non-native grid like BASELINE's 288 (feat 64 -> G, T % 64 == 0: the LDS-DMA attention
    kernels, warped-feat branch live from step 2); G = 72 is ragged (T = 1296, T % 64 = 16: the register-staged
    attention fallback and the GEMM edge tiles).  The two large grids run one hypothesis (and the ragged one 25 steps) to
    keep the oracle - which dominates this suite's run time - inside the driver's time limit; the whole 50-step loop at
    G = 288 is recorded once per round in profiles/ (tests/tools/parity_g288.py).
    The bar of 1e-3 is asserted on the UN-CLAMPED, UN-AVERAGED x0 of the last step; on the tame family (every pixel
    inside (-1, 1), like a trained model's coordinates) no error hides behind the final clamp."""
    from dvd_amd import sampler, schedule
    from oracle import dvd_oracle as O
    gain = synth.tame_gain(steps) if family == "tame" else 1.0
    eng, orc, doc_t, inv1 = setup(grid, 1, hyp, gain)
    tab = schedule.Tables(schedule.named_betas("cosine", steps))
    xT = torch.from_numpy(synth.synth_noise(0, hyp, grid, SEED_IN))
    tr_ref, tr = [], []
    ref = orc.sample_loop(O.Schedule(steps), xT, {k: v[:1] for k, v in doc_t.items()}, trace=tr_ref)
    out = sampler.sample(eng, tab, xT.cuda(), trace=tr)
    per = [float((a.cpu() - b).pow(2).mean().sqrt()) for a, b in zip(tr, tr_ref)]
    err = float((out.cpu() - ref).pow(2).mean().sqrt())
    sat = _saturated(tr_ref[-1])
    print(f"long loop rmse G={grid} S={steps} H={hyp} {family}: final {err:.2e}, un-clamped last x0 {per[-1]:.2e}, "
          f"last x0 std {float(tr_ref[-1].std()):.3f}, saturated pixels {sat:.4f}, per-step[::7]", per[::7])
    if grid >= 66:
        # large grids dither the weights of the 256-wide GEMMs by default (one GEMM pass); the same loop with the
        # (hi, lo) split everywhere (2x the GEMM MFMAs, round 2's default) and with plain f16 weights, for the record
        try:
            for name, opts in (("split (dither off)", {"dither": 0}), ("plain f16 (no split, no dither)",
                                                                       {"dither": 0, "split_weights": 0})):
                for k, v in opts.items():
                    eng.set_option(k, v)
                tr2 = []
                out2 = sampler.sample(eng, tab, xT.cuda(), trace=tr2)
                per2 = [float((a.cpu() - b).pow(2).mean().sqrt()) for a, b in zip(tr2, tr_ref)]
                print(f"   {name}: final {float((out2.cpu() - ref).pow(2).mean().sqrt()):.2e}, un-clamped last x0 "
                      f"{per2[-1]:.2e}, per-step[::7]", per2[::7])
                if "split_weights" not in opts:
                    assert per2[-1] < 1e-3, per2[-1]
        finally:
            eng.set_option("dither", 1)
            eng.set_option("split_weights", 1)
    assert per[-1] < 1e-3, per[-1]          # un-clamped, un-averaged x0 of the last step
    if family == "tame":
        assert sat < 0.01, sat              # the family does what it is for
        assert per[-1] < 6e-5, per[-1]      # measured 1.3e-5 .. 2.1e-5 on MI355X (x3), dithered or split alike
    assert err < 3.8e-4, (err, per[-1])


@pytest.mark.parametrize("steps", [25, 250])
def test_ddpm_loop_vs_oracle(steps):
    """BASELINE configs[3]'s sampler (DDPM ancestral, FIXED_LARGE variance, fixed noise table) at G = 16 against the
    oracle, short and at full length: 250 noise-driven steps accumulate the denoiser's error.  Tame family: the bar is
    asserted on the un-clamped, un-averaged last x0."""
    from dvd_amd import sampler, schedule
    from oracle import dvd_oracle as O
    grid = 16
    eng, orc, doc_t, inv1 = setup(grid, 1, 2, synth.tame_gain(steps))
    tab = schedule.Tables(schedule.named_betas("cosine", steps))
    xT = torch.from_numpy(synth.synth_noise(0, 2, grid, SEED_IN))
    noises = {i: torch.from_numpy(synth.synth_noise(0, 2, grid, SEED_IN, step=i)) for i in range(steps)}
    tr_ref, tr = [], []
    ref = orc.sample_loop(O.Schedule(steps), xT, {k: v[:1] for k, v in doc_t.items()}, sampler="ddpm", noises=noises,
                          trace=tr_ref)
    out = sampler.sample(eng, tab, xT.cuda(), sampler="ddpm", noise_fn=lambda i: noises[i].cuda(), trace=tr)
    per = [float((a.cpu() - b).pow(2).mean().sqrt()) for a, b in zip(tr, tr_ref)]
    err = float((out.cpu() - ref).pow(2).mean().sqrt())
    sat = _saturated(tr_ref[-1])
    print(f"ddpm loop S={steps}: final {err:.2e}, un-clamped last x0 {per[-1]:.2e}, last x0 std "
          f"{float(tr_ref[-1].std()):.3f}, saturated pixels {sat:.4f}, per-step[::25]", per[::25])
    assert sat < 0.01, sat
    assert per[-1] < 1e-3 and err < 1e-3, (err, per[-1])
    assert per[-1] < 6e-5, per[-1]          # measured 2.1e-5 (25 steps) / 1.1e-5 (250 steps) on MI355X (x3)


def _ddpm_large_grid(steps):
    """BASELINE configs[3] runs its ancestral steps at G = 288, where the weights of the 256-wide GEMMs are dithered:
    the same sampler on a LARGE-tile grid (G = 72: T = 1296 > 1024 tokens, ragged), one hypothesis, tame family -
    un-clamped last x0 against the oracle, dithered and split.  40 steps here (the oracle costs 2 s per step); the full
    250 steps are run once per round by tests/tools/ddpm250_large_grid.py (8.5 minutes; profiles/r3_ddpm250_g72.txt:
    6.5e-6 dithered, 6.2e-6 split)."""
    from dvd_amd import sampler, schedule
    from oracle import dvd_oracle as O
    grid = 72
    eng, orc, doc_t, inv1 = setup(grid, 1, 1, synth.tame_gain(steps))
    tab = schedule.Tables(schedule.named_betas("cosine", steps))
    xT = torch.from_numpy(synth.synth_noise(0, 1, grid, SEED_IN))
    noises = {i: torch.from_numpy(synth.synth_noise(0, 1, grid, SEED_IN, step=i)) for i in range(steps)}
    tr_ref = []
    orc.sample_loop(O.Schedule(steps), xT, {k: v[:1] for k, v in doc_t.items()}, sampler="ddpm", noises=noises, trace=tr_ref)
    res = {}
    try:
        for name, dither in (("dither", 1), ("split", 0)):
            eng.set_option("dither", dither)
            tr = []
            sampler.sample(eng, tab, xT.cuda(), sampler="ddpm", noise_fn=lambda i: noises[i].cuda(), trace=tr)
            res[name] = float((tr[-1].cpu() - tr_ref[-1]).pow(2).mean().sqrt())
    finally:
        eng.set_option("dither", 1)
    print(f"ddpm {steps} steps at G=72: un-clamped last x0 rmse {res}, last x0 std {float(tr_ref[-1].std()):.3f}, saturated "
          f"{_saturated(tr_ref[-1]):.4f}")
    assert _saturated(tr_ref[-1]) < 0.01
    assert res["dither"] < 1e-4 and res["split"] < 1e-4, res     # measured 6.5e-6 / 6.2e-6 at 250 steps
    return res


def test_ddpm_large_grid_vs_oracle():
    _ddpm_large_grid(40)


def test_batched_documents_match_single():
    """Two documents x two hypotheses in one engine == each document alone (no cross-document math)."""
    from dvd_amd import sampler, schedule
    from dvd_amd.engine import Engine
    grid = 16
    sd = synth.synth_state_dict(grid, SEED_W, blocks=[11])
    tab = schedule.Tables(schedule.named_betas("cosine", 3))
    ds = [synth.synth_document(d, grid, SEED_IN) for d in range(2)]
    keys = ("y512", "mask_cat", "mask_y512", "line_msk")
    xT = torch.from_numpy(np.concatenate([synth.synth_noise(d, 2, grid, SEED_IN) for d in range(2)])).cuda()
    both = Engine(grid, 2, 2)
    both.load_state_dict(sd)
    both.prepare(*[torch.from_numpy(np.stack([d[k] for d in ds])).cuda() for k in keys])
    out2 = sampler.sample(both, tab, xT).cpu()
    one = Engine(grid, 1, 2)
    one.load_state_dict(sd)
    for d in range(2):
        one.prepare(*[torch.from_numpy(ds[d][k][None]).cuda() for k in keys])
        o = sampler.sample(one, tab, xT[2 * d:2 * d + 2].contiguous()).cpu()
        assert torch.equal(o[0], out2[d]), d


@pytest.mark.parametrize("grid", [16, 72])
def test_graph_replay_equals_eager(grid):
    """A sampling loop whose denoiser evaluations are replayed as captured hipGraphs (the default for grids <= 128) gives
    the same bits as the eagerly enqueued launch sequence - first use of an address triple runs eagerly, the second
    captures, later ones replay, so three roll-outs exercise all three.  G = 72 is a large-tile grid: the weights are
    re-rounded (dithered) before every evaluation OUTSIDE the captured graph, whose GEMMs read the re-rounded copy - replay
    must pick up each step's copy, and a roll-out must be a function of its inputs alone (dither_step = loop index)."""
    from dvd_amd import sampler, schedule
    eng, orc, doc_t, inv1 = setup(grid)
    tab = schedule.Tables(schedule.named_betas("cosine", 10))
    xT = torch.from_numpy(synth.synth_noise(0, 2, grid, SEED_IN)).cuda()
    try:
        eng.set_option("graphs", 0)
        eager = sampler.sample(eng, tab, xT).clone()
        eng.set_option("graphs", 1)
        runs = [sampler.sample(eng, tab, xT).clone() for _ in range(3)]
    finally:
        eng.set_option("graphs", 1)
    for r in runs:
        assert torch.equal(r, eager)


def test_engine_at_the_baseline_grid_vs_oracle():
    """G = 288 (BASELINE configs[1]-[4]) under -m gpu: Engine(288, 2 documents, 2 hypotheses).  (a) the first loop step
    (t_model > 600: init_feat <- feat) and an evaluation on the warped-feature branch (feat_mode 2, init_flow = the
    first step's x0) of document 0 / hypothesis 0 against the CPU oracle, coordinate RMSE < 1e-3; (b) the two documents
    batched give the same bits as each document alone (no cross-document arithmetic at the production size)."""
    from dvd_amd import sampler, schedule
    from dvd_amd.engine import Engine
    from oracle import dvd_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    G, S = 288, 50
    sd = synth.synth_state_dict(G, SEED_W, blocks=[11])
    keys = ("y512", "mask_cat", "mask_y512", "line_msk")
    ds = [synth.synth_document(d, G, SEED_IN) for d in range(2)]
    doc_t = {k: torch.from_numpy(np.stack([d[k] for d in ds])) for k in keys}
    xT = torch.from_numpy(np.concatenate([synth.synth_noise(d, 2, G, SEED_IN) for d in range(2)]))
    tab = schedule.Tables(schedule.named_betas("cosine", S))
    both = Engine(G, 2, 2)
    both.load_state_dict(sd)
    both.prepare(*[doc_t[k].cuda() for k in keys])
    t_first = tab.model_time(S - 1)
    i_mid = max(i for i in range(S) if tab.model_time(i) <= 600.0)
    t_mid = tab.model_time(i_mid)
    zeros = torch.zeros(4, 2, G, G, device="cuda")
    x0_first = both.denoise(xT.cuda(), schedule.embedded_time(t_first), sampler.feat_mode_for(t_first, 4, True), zeros,
                            dither_step=0).clone()
    x0_mid = both.denoise(xT.cuda(), schedule.embedded_time(t_mid), 2, x0_first, dither_step=1).clone()
    # (b) batch == single, bit for bit
    one = Engine(G, 1, 2)
    one.bind_blob(both.blob)
    for d in range(2):
        one.prepare(*[doc_t[k][d:d + 1].cuda() for k in keys])
        a = one.denoise(xT[2 * d:2 * d + 2].cuda(), schedule.embedded_time(t_first), sampler.feat_mode_for(t_first, 2, True),
                        zeros[:2], dither_step=0).clone()
        b = one.denoise(xT[2 * d:2 * d + 2].cuda(), schedule.embedded_time(t_mid), 2, x0_first[2 * d:2 * d + 2].contiguous(),
                        dither_step=1)
        assert torch.equal(a, x0_first[2 * d:2 * d + 2]), d
        assert torch.equal(b, x0_mid[2 * d:2 * d + 2]), d
    del one
    # (a) against the oracle (document 0, hypothesis 0)
    orc = O.Oracle(sd, G)
    with torch.no_grad():
        inv = orc.prepare(*[doc_t[k][:1] for k in keys])
        x = xT[:1]
        ref_first, _ = orc.forward(x, float(t_first), inv, torch.zeros_like(x), inv["feat"])
        e1 = float((ref_first - x0_first[:1].cpu()).pow(2).mean().sqrt())
        flow = x0_first[:1].cpu()
        init_feat = O.grid_sample_ref(inv["feat"], (flow + O.base_grid(G, G)) * 2 - 1)
        ref_mid, _ = orc.forward(x, float(t_mid), inv, flow, init_feat)
        e2 = float((ref_mid - x0_mid[:1].cpu()).pow(2).mean().sqrt())
    print(f"G=288 engine vs oracle: first step {e1:.2e}, warped-feature branch {e2:.2e}")
    # ONE evaluation sees the whole f16 rounding of the dithered weights (the dither only averages out over steps):
    # measured 1.05e-4 / 1.25e-4 (5.2e-5 with the split in every GEMM); north_star's bar is 1e-3
    assert e1 < 3e-4 and e2 < 3e-4, (e1, e2)
