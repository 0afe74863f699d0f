#!/usr/bin/env python3
"""(Checker-side tool: lives under tests/ because it runs the oracle.)  Writes the CPU ORACLE's roll-outs of the long
loops to tests/golden/oracle_<case>.npz, so that the `-m gpu` tests run only the HIP engine and compare - the oracle's
share of the GPU suite was 80 % of its 833 s (round-3 VERDICT, weak 5), and the whole 50-step loop at G = 288 (95 minutes
of oracle time on 8 cores) could not be under pytest at all.

A fixture holds NO tensor input: every input (weights, document, x_T, DDPM noise table) is regenerated on both sides from
the seeds below by dvd_amd/synth.py's counter-based generator; the file holds the oracle's un-clamped x0 at the KEPT steps
(loop positions k = 0 .. S-1; k = S-1 is the last step), the final map, and the parameters the test re-derives the inputs
from.  One live-oracle loop stays in the GPU suite (test_long_loop_vs_oracle at G = 32) so the oracle still executes on
the box.

usage: python tests/tools/gen_oracle_traces.py <case> [<case> ...]      (cases: see CASES; `all` = every case)
       ORACLE_THREADS=8 ...                                             (default: all cores)
"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvd_amd import synth  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")
SEED_W, SEED_IN = 7, 1234          # the seeds of tests/test_gpu_engine.py

# name: (grid, steps, hypotheses, family, sampler, kept loop positions)
CASES = {
    "ddim_g288_s50_tame": (288, 50, 1, "tame", "ddim", [0, 14, 28, 42, 49]),
    "ddim_g96_s50_tame": (96, 50, 1, "tame", "ddim", [0, 7, 14, 21, 28, 35, 42, 49]),
    "ddim_g96_s50_plain": (96, 50, 1, "plain", "ddim", [0, 7, 14, 21, 28, 35, 42, 49]),
    "ddim_g72_s25_tame": (72, 25, 1, "tame", "ddim", [0, 7, 14, 21, 24]),
    "ddim_g16_s50_plain": (16, 50, 2, "plain", "ddim", list(range(0, 50, 7)) + [49]),
    "ddim_g16_s50_tame": (16, 50, 2, "tame", "ddim", list(range(0, 50, 7)) + [49]),
    "ddpm_g16_s25_tame": (16, 25, 2, "tame", "ddpm", [0, 12, 24]),
    "ddpm_g16_s250_tame": (16, 250, 2, "tame", "ddpm", list(range(0, 250, 25)) + [249]),
    "ddpm_g72_s40_tame": (72, 40, 1, "tame", "ddpm", [0, 13, 26, 39]),
    # round 5 (VERDICT r4 missing 3 / 6): BASELINE configs[3]'s ancestral sampler on the 20 736-token engine, and the
    # headline loop on the PLAIN family (the golden vectors' family) with the reference's two hypotheses
    "ddpm_g288_s10_tame": (288, 10, 1, "tame", "ddpm", [0, 4, 9]),
    "ddim_g288_s50_plain": (288, 50, 2, "plain", "ddim", [0, 28, 49]),
    # round 6: a family with PEAKED decoder attention (logits x 12: std ~ 10 instead of ~ 0.9, the largest
    # probability of a row ~ 0.5 instead of ~ 0.01 - the deferred-rescale branch of the online softmax and f16 P with few
    # dominant terms), and BASELINE configs[3]'s sampler at its REAL LENGTH (250 ancestral steps) on a grid that runs the
    # headline kernels (T = 6400 >= 5376: r64x / h64x attention, 384 x 256 GEMMs on dithered weights).  250 steps at G = 288
    # are ~8 h of oracle on all 8 cores of the build box (114 s per step), G = 160 is ~1.5 h.
    "ddim_g96_s50_peaked": (96, 50, 1, "peaked", "ddim", [0, 7, 14, 21, 28, 35, 42, 49]),
    "ddpm_g160_s250_tame": (160, 250, 1, "tame", "ddpm", list(range(0, 250, 25)) + [249]),
    # ... and the longest ancestral chain the build box affords at the REAL grid: 100 steps at G = 288 (~3.5 h of oracle on 7 cores)
    "ddpm_g288_s100_tame": (288, 100, 1, "tame", "ddpm", [0, 24, 49, 74, 99]),
    # the peaked-attention family on the HEADLINE kernels (r64x / h64x generated loops with their deferred-rescale blocks, T = 20 736)
    "ddim_g288_s50_peaked": (288, 50, 1, "peaked", "ddim", [0, 28, 49]),
}
PEAK_LOGIT_GAIN = 12.0


def inputs(grid, steps, hyp, family, sampler):
    """Everything a roll-out of a case needs, from seeds (used by this tool AND by the tests)."""
    gain = synth.tame_gain(steps) if family in ("tame", "peaked") else 1.0
    sd = synth.synth_state_dict(grid, SEED_W, blocks=[11], out_gain=gain)
    if family == "peaked":       # the tame family with the decoder's q / k projections scaled: logits x PEAK_LOGIT_GAIN
        for k in list(sd):
            if k.endswith("attn.linear_q.weight") or k.endswith("attn.linear_k.weight"):
                sd[k] = (np.asarray(sd[k]) * np.float32(PEAK_LOGIT_GAIN ** 0.5)).astype(np.float32)
    d0 = synth.synth_document(0, grid, SEED_IN)
    doc = {k: torch.from_numpy(d0[k])[None] for k in ("y512", "mask_cat", "mask_y512", "line_msk")}
    xT = torch.from_numpy(synth.synth_noise(0, hyp, grid, SEED_IN))
    noises = None
    if sampler == "ddpm":
        noises = {i: torch.from_numpy(synth.synth_noise(0, hyp, grid, SEED_IN, step=i)) for i in range(steps)}
    return sd, doc, xT, noises, gain


def generate(name):
    grid, steps, hyp, family, sampler, kept = CASES[name]
    sd, doc, xT, noises, gain = inputs(grid, steps, hyp, family, sampler)
    orc = O.Oracle(sd, grid)
    tr = []
    t0 = time.time()
    with torch.no_grad():
        final = orc.sample_loop(O.Schedule(steps), xT, doc, sampler=sampler, noises=noises, trace=tr)
    dt = time.time() - t0
    last = tr[-1]
    out = os.path.join(GOLD, f"oracle_{name}.npz")
    np.savez_compressed(out, kept=np.asarray(kept, dtype=np.int32), x0=np.stack([tr[k].numpy() for k in kept]),
                        final=final.numpy(), grid=grid, steps=steps, hyp=hyp, family=family, sampler=sampler,
                        out_gain=np.float64(gain), seed_w=SEED_W, seed_in=SEED_IN,
                        x0_std=np.asarray([float(t.std()) for t in tr], dtype=np.float32),
                        last_x0_saturated=float((last.abs() >= 1).float().mean()), oracle_seconds=round(dt, 1),
                        oracle_threads=torch.get_num_threads())
    print(f"{name}: {dt:.0f} s on {torch.get_num_threads()} threads, last x0 std {float(last.std()):.3f}, "
          f"saturated {float((last.abs() >= 1).float().mean()):.4f} -> {out} ({os.path.getsize(out) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 8)))
    names = sys.argv[1:]
    if names == ["all"]:
        names = list(CASES)
    if not names or any(n not in CASES for n in names):
        raise SystemExit(__doc__)
    for n in names:
        generate(n)
