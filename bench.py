#!/usr/bin/env python3
"""Headline benchmark: documents/second of the DvD sampling path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of documents (BASELINE.json configs[1]):
  8 documents x 2 hypotheses per GPU, 50-step DDIM on the 288x288 coordinate grid
  = per-document conditioning (conv pyramid, c/m/l embeddings, their K/V)   [dvd_engine_prepare_docs]
  + 50 x denoiser evaluation + fused DDIM step                              [dvd_engine_denoise_step, dvd_sched_step]
  + hypothesis mean/clamp + full-resolution (3508x2480) u8 unwarp           [dvd_hyp_mean_clamp, dvd_unwarp_u8]
with every input already resident in HBM.  Documents shard across ranks with no data-path
collective (weak scaling, 8 docs per GPU); the only collective is the one-shot RCCL broadcast of the
packed weight blob from rank 0.  Synthetic inputs and weights (no checkpoint / dataset offline).

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  roofline     : the dominant kernel (head_dim-256 decoder flash attention), timed per launch with HIP
                 events on the launch stream inside the timed region, against the dense f16 MFMA peak;
  cpu_baseline : the CPU oracle (a port pinned to the reference by golden vectors) timed on this
                 host's cores on a bounded sample and scaled to documents/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F16_TFLOPS = 2500.0      # dense f16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HID, DEC, FFN = 384, 1536, 2048


def step_flops(grid):
    """Algorithmic FLOPs of ONE denoiser evaluation for ONE sample (live block only, invariants hoisted;
    SURVEY Appendix A.7) and the share of the head_dim-256 decoder attention."""
    T = (grid // 2) ** 2
    D, E, F = HID, DEC, FFN
    block = 2 * T * 8 * D + 2 * T * 1032 * D + (2 * T * D * D + 4 * T * D * D + 4 * 4 * T * T * D + 4 * 2 * T * D * D) \
        + (4 * 6 * T * D * D + 4 * 4 * T * T * D + 4 * 2 * T * D * D) + 4 * 16 * T * D * D
    dec = 6 * (6 * T * E * E + 4 * T * T * E + 2 * T * E * E + 4 * T * E * F + 18 * T * F) + 16 * T * E
    return block + dec, 4 * T * T * E          # (total per sample-step, one decoder-attention launch per sample)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--grid", type=int, default=288)
    ap.add_argument("--docs", type=int, default=8, help="documents per GPU per step")
    ap.add_argument("--hyp", type=int, default=2, help="hypotheses per document (reference n_batch)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ddpm"])
    ap.add_argument("--full-res", default="3508x2480")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-weights", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    from dvd_amd import ops, sampler, schedule, synth
    from dvd_amd.engine import Engine

    G, B, H, S = args.grid, args.docs, args.hyp, args.ddim_steps
    FH, FW = (int(v) for v in args.full_res.split("x"))
    eng = Engine(G, B, H, device=dev)
    if args.no_split_weights:
        eng.set_option("split_weights", 0)

    # ---- weights: rank 0 builds + packs, ONE flat RCCL broadcast (the only collective of the path) ----
    _, blob_bytes = eng.blob_layout()
    if rank == 0:
        sd = synth.synth_state_dict(G, seed=7, blocks=[11])
        blob = eng.pack_blob(sd).to(dev)
        del sd
    else:
        blob = torch.empty(blob_bytes, dtype=torch.uint8, device=dev)
    bcast_ms = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        dist.broadcast(blob, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    eng.bind_blob(blob)

    # ---- synthetic per-rank documents, resident in HBM (value ranges as SURVEY 8(d)) ----
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    y512 = torch.rand(B, 3, 512, 512, device=dev, generator=gen)
    mask_cat = torch.rand(B, 1, 512, 512, device=dev, generator=gen)
    mask_y512 = torch.randn(B, 384, G, G, device=dev, generator=gen).clamp_min_(0)
    line_msk = torch.randn(B, 64, G, G, device=dev, generator=gen).clamp_min_(0)
    x_T = torch.randn(B * H, 2, G, G, device=dev, generator=gen)
    src_u8 = torch.randint(0, 256, (B, FH, FW, 3), device=dev, dtype=torch.uint8, generator=gen)
    tab = schedule.Tables(schedule.named_betas("cosine", S))
    noise_fn = None
    if args.sampler == "ddpm":
        noise_fn = lambda i: torch.randn(B * H, 2, G, G, device=dev, generator=gen)  # noqa: E731

    def one_step():
        eng.prepare(y512, mask_cat, mask_y512, line_msk)
        flow = sampler.sample(eng, tab, x_T, sampler=args.sampler, noise_fn=noise_fn)     # [B,2,G,G]
        outs = [ops.unwarp_u8(flow[d:d + 1], src_u8[d]) for d in range(B)]
        return flow, outs

    for _ in range(args.warmup):
        one_step()
    eng.profile(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        flow, outs = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    launches, attn_ms = eng.profile_read()
    eng.profile(False)
    assert bool(torch.isfinite(flow).all()), "non-finite coordinates"

    if rank == 0:
        docs_total = B * world * args.steps
        value = docs_total / elapsed
        per_sample_step, attn_launch_per_sample = step_flops(G)
        n = B * H
        roof = None
        if launches:
            avg_s = attn_ms / launches * 1e-3
            achieved = attn_launch_per_sample * n / avg_s / 1e12
            roof = {"kernel": "flash_attn_r64_kernel<0> (decoder self-attention, 6 heads x 256)", "bound": "mfma",
                    "achieved": round(achieved, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": 4 * n * (G // 2) ** 2 * 1536 * 2,   # Q, K, V^T read + O written, f16
                    "launches_timed": launches, "avg_launch_ms": round(avg_s * 1e3, 3),
                    "flops_per_launch": attn_launch_per_sample * n,
                    "share_of_step_time": round(attn_ms * 1e-3 / elapsed, 3)}
            # HBM bytes per launch of this kernel come from PMC counters, which need their own rocprofv3 passes
            # (benchmarks/pmc_traffic.sh); bench.py reports the committed measurement when the launch shape matches.
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_traffic.json")
            if os.path.exists(tpath) and (n, G) == (16, 288):
                tj = json.load(open(tpath))
                key = [k for k in tj if "flash_attn_r64_kernel" in k] or [k for k in tj if "flash_attn_glds_kernel<256" in k]
                if key:
                    roof["traffic"] = int(tj[key[0]]["hbm_bytes_per_launch"])
                    roof["traffic_unit"] = "bytes/launch"
                    roof["traffic_source"] = "profiles/r1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; 2*FETCH+WRITE)"
            mpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_mfma.json")
            if os.path.exists(mpath) and (n, G) == (16, 288):
                mj = json.load(open(mpath)).get("flash_attn_r64_kernel")
                if mj:   # hardware-side view of the same kernel at the same launch shape (separate PMC pass)
                    roof["mfma_busy_pmc"] = mj["mfma_busy"]
                    roof["sustained_clock_ghz_pmc"] = mj["sustained_clock_ghz"]
                    roof["pmc_source"] = "profiles/r1_pmc_mfma.json"
        flops_total = per_sample_step * n * S * world * args.steps
        if not args.no_split_weights:
            pass   # split weights double the GEMM MFMAs; algorithmic FLOPs are unchanged by definition
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            # engine output for the run's first loop step (all samples; sample 0 is compared with the oracle)
            t_first = tab.model_time(S - 1)
            eng.prepare(y512, mask_cat, mask_y512, line_msk)
            x0_first = eng.denoise(x_T, schedule.embedded_time(t_first), sampler.feat_mode_for(t_first, B * H, True),
                                   torch.zeros_like(x_T))
            check = {"doc": [a[:1].float().cpu() for a in (y512, mask_cat, mask_y512, line_msk)], "x": x_T[:1].cpu(),
                     "t_model": float(t_first), "x0_gpu": x0_first[:1].cpu()}
            cpu = cpu_baseline(G, H, S, check)
            if cpu.get("parity") and not cpu["parity"]["ok"]:
                raise SystemExit(f"PARITY FAILURE at G={G}: {cpu['parity']}")
        line = {
            "metric": "documents/sec (50-step DDIM, 288x288 grid)" if (G, S, args.sampler) == (288, 50, "ddim")
            else f"documents/sec ({S}-step {args.sampler.upper()}, {G}x{G} grid)",
            "value": round(value, 5), "unit": "documents/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: batch={B} documents/GPU x {H} hypotheses, {S}-step "
                                   f"{args.sampler.upper()}, {G}x{G} coordinate grid, + {FH}x{FW} u8 unwarp",
                       "docs_per_gpu": B, "hypotheses": H, "sampler_steps": S, "grid": G,
                       "weights": "synthetic (seed 7), f16 hi/lo split" if not args.no_split_weights else "synthetic, f16",
                       "parallelism": f"dp{world} (documents sharded, one weight broadcast)"},
            "algorithmic_tflops": round(flops_total / elapsed / 1e12, 1),
            "weight_broadcast_ms": None if bcast_ms is None else round(bcast_ms, 2),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(grid, hyp, steps, check=None):
    """Time the CPU oracle (kind 'port': a restatement pinned to the reference by the golden vectors) on a
    bounded sample: ONE denoiser evaluation for ONE sample (of hyp*steps per document) with the same hoisted
    algebra as the GPU engine, scaled linearly to documents/s.  The evaluation is run at the benchmark grid
    when the host can do it in well under a minute, otherwise at G=64 and scaled by the FLOP ratio."""
    from dvd_amd import synth
    from oracle import dvd_oracle as O
    # torch's intra-op pool stops scaling (and then collapses) long before a 256-thread host is full on these
    # op sizes; 32 threads is at or past the knee on every host tried.  `cores` reports the threads used.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)

    def time_one(g):
        sd = synth.synth_state_dict(g, seed=7, blocks=[11])
        orc = O.Oracle(sd, g)
        gen = torch.Generator().manual_seed(5)
        T = (g // 2) ** 2
        inv = {"feat": torch.rand(1, 256, g, g, generator=gen),
               "cond": torch.randn(1, T, 384, generator=gen), "msk6": torch.randn(1, T, 384, generator=gen),
               "line": torch.randn(1, T, 384, generator=gen)}
        x = torch.randn(1, 2, g, g, generator=gen)
        flow = torch.zeros(1, 2, g, g)
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.forward(x, 666.7, inv, flow, inv["feat"])
        return time.perf_counter() - t0

    def time_and_check(g):
        """The timed evaluation doubles as a FULL-SIZE parity check: it is the first loop step of document 0 /
        hypothesis 0 of this very run (same synthetic weights, conditioning and x_T), compared with what the engine
        produced for that sample."""
        sd = synth.synth_state_dict(g, seed=7, blocks=[11])
        orc = O.Oracle(sd, g)
        with torch.no_grad():
            inv = orc.prepare(*check["doc"])                      # once-per-document work: not part of the timed step
            t0 = time.perf_counter()
            x0_ref, _ = orc.forward(check["x"], check["t_model"], inv, torch.zeros_like(check["x"]), inv["feat"])
            dt = time.perf_counter() - t0
        rmse = float((x0_ref - check["x0_gpu"]).pow(2).mean().sqrt())
        return dt, {"what": f"x0 prediction of the first loop step (t_model = {check['t_model']:.1f}) of document 0 / "
                            f"hypothesis 0 at G={g}: HIP engine vs CPU oracle", "coord_rmse": rmse, "bar": 1e-3,
                    "ok": bool(rmse < 1e-3)}

    t64 = time_one(64)
    f64, _ = step_flops(64)
    fg, _ = step_flops(grid)
    est = t64 * fg / f64
    parity = None
    if grid != 64 and est < 45.0 and check is not None:
        t_step, parity = time_and_check(grid)
        sample = (f"1 sample x 1 denoiser evaluation at G={grid} (of {hyp * steps} per document): the run's own first "
                  "loop step of document 0, hoisted algebra")
    elif grid != 64 and est < 45.0:
        t_step, sample = time_one(grid), f"1 sample x 1 denoiser evaluation at G={grid} (of {hyp * steps} per document), hoisted algebra"
    else:
        t_step = est
        sample = (f"1 sample x 1 denoiser evaluation at G=64 ({t64:.2f} s), scaled by the FLOP ratio to G={grid} "
                  f"(of {hyp * steps} per document), hoisted algebra")
    docs_per_s = 1.0 / (hyp * steps * t_step)
    out = {"value": round(docs_per_s, 7), "unit": "documents/s", "cores": cores, "kind": "port", "sample": sample,
           "seconds_per_sample_step": round(t_step, 3)}
    if parity is not None:
        out["parity"] = parity
    return out


if __name__ == "__main__":
    main()
