#!/usr/bin/env python3
"""Single-document latency at the REFERENCE'S NATIVE operating point (admin/local.py:28-35,82 of the reference):
G = 64 coordinate grid, 3 DDIM steps, 2 hypotheses, one document - conditioning resident in HBM -> unwarped u8 image.

At this size a denoiser evaluation is ~110 launches of a few microseconds each, i.e. launch-bound: the engine replays
each evaluation as one captured hipGraph (dvd_engine_set_option "graphs").  Prints one JSON object with the latency
with and without graph replay and, beside it, the CPU oracle (hoisted algebra) on this host's cores.

    python benchmarks/latency_native.py [--full-res 1024x768] [--reps 30] [--no-cpu]
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvd_amd import ops, sampler, schedule, synth  # noqa: E402
from dvd_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--hyp", type=int, default=2)
    ap.add_argument("--full-res", default="1024x768")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    G, S, H = a.grid, a.steps, a.hyp
    FH, FW = (int(v) for v in a.full_res.split("x"))
    dev = torch.device("cuda", 0)
    sd = synth.synth_state_dict(G, seed=7, blocks=[11])
    eng = Engine(G, 1, H, device=dev)
    eng.load_state_dict(sd)
    doc = synth.synth_document(0, G, 1234, full_res=(FH, FW))
    keys = ("y512", "mask_cat", "mask_y512", "line_msk")
    cond = [torch.from_numpy(doc[k][None]).to(dev) for k in keys]
    src = torch.from_numpy(doc["src_u8"][None]).to(dev)
    xT = torch.from_numpy(synth.synth_noise(0, H, G, 1234)).to(dev)
    tab = schedule.Tables(schedule.named_betas("cosine", S))

    def one():
        eng.prepare(*cond)
        flow = sampler.sample(eng, tab, xT)
        return ops.unwarp_u8_batch(flow, src), flow

    def split():
        """prepare / loop / unwarp separately (each synchronised)"""
        ts = []
        torch.cuda.synchronize(); t0 = time.perf_counter(); eng.prepare(*cond); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); flow = sampler.sample(eng, tab, xT); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ops.unwarp_u8_batch(flow, src); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return ts

    res = {"what": f"one document, G={G}, {S}-step DDIM, {H} hypotheses, + {FH}x{FW} u8 unwarp (reference-native point)"}
    flows = {}
    for tag, on, small in (("eager_256tiles", 0, 0), ("eager", 0, 1), ("graphs", 1, 1)):
        eng.set_option("graphs", on)
        eng.set_option("small_tiles", small)
        for _ in range(4):
            one()
        torch.cuda.synchronize()
        lat = []
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out, flow = one()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        parts = np.median(np.array([split() for _ in range(10)]), axis=0) * 1e3
        flows[tag] = flow.clone()
        res[tag] = {"ms_per_document_median": round(statistics.median(lat), 3), "ms_min": round(min(lat), 3),
                    "documents_per_s": round(1e3 / statistics.median(lat), 2),
                    "ms_prepare_docs": round(float(parts[0]), 3), "ms_sampling_loop": round(float(parts[1]), 3),
                    "ms_unwarp": round(float(parts[2]), 3)}
    res["bit_identical_eager_vs_graphs"] = bool(torch.equal(flows["eager"], flows["graphs"]))
    if not a.no_cpu:
        from oracle import dvd_oracle as O
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        orc = O.Oracle(sd, G)
        d1 = {k: torch.from_numpy(doc[k][None]) for k in keys}
        with torch.no_grad():
            t0 = time.perf_counter()
            ref = orc.sample_loop(O.Schedule(S), xT.cpu(), d1)
            dt = time.perf_counter() - t0
        res["cpu_oracle"] = {"seconds_per_document": round(dt, 3), "cores": cores, "kind": "port",
                             "mode": "hoisted (live block only, pyramid once per document)",
                             "coord_rmse_gpu_vs_oracle": float((flows["graphs"].cpu() - ref).pow(2).mean().sqrt())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
