#!/usr/bin/env python3
"""Benchmark of the IEA-GAN G+D train step on MI355X (BASELINE.json metric: events/s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload at every N: BASELINE configs[1] per GPU -- one event (40 sensors x 256x768, i.e. 250x768 padded)
per step, bf16 activations, hinge loss only, RRM on; synthetic PXD-like events, seeded orthogonal weights.
N > 1: events shard data-parallel (one per rank and step), gradients are all-reduced over RCCL
(``parallel.py``); value = events all ranks processed / max-over-ranks wall time ("weak" scaling).

One JSON line on stdout (rank 0).  ``roofline`` describes the kernel that dominates GPU time, timed with
HIP events on the launch stream inside the timed region; ``cpu_baseline`` is the CPU oracle (a plain
PyTorch restatement of the reference step, oracle/) timed on this host on a bounded sample.
"""
import argparse
import contextlib
import io
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "iea-gan_amd"))

import numpy as np
import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0         # HBM3E spec (6.3 TB/s measured achievable)
STEP_GFLOP = 5328.8           # algorithmic work per event-step (SURVEY 8d / BASELINE.md section 3)


def synth_event(n, h, w, seed):
    """PXD-like synthetic event: background -1, ~1 % log-normalised hits, 4e-3 dequantisation noise."""
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    hit = rng.random((n, 1, h, w)) < 0.01
    u = rng.uniform(0.03, 1.0, (n, 1, h, w))
    img = np.where(hit, np.log(255.0 * u + 1.0) / math.log(256.0), 0.0) + 4e-3 * rng.random((n, 1, h, w))
    return torch.from_numpy((2.0 * img - 1.0).astype(np.float32))


def bench_config():
    from defaults import default_config
    cfg = default_config()
    # configs[1]: hinge loss only, RRM on.  clip_norm must be finite or the reference's control flow never
    # steps G's optimiser -- a benchmark without the G update would skip real work.
    cfg.update(device="cuda", contra_lambda=0.0, IEA_loss=False, Uniformity_loss=False, clip_norm=1e9)
    return cfg


def cpu_baseline(cfg, sensors=8, threads=None):
    """The oracle's train step at full 256x768 resolution on `sensors` of the 40 sensors of one event
    (BN / RRM / losses over that sub-event), scaled by 40/sensors: bounded to tens of seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ieagan_oracle as O
    threads = threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    c = dict(cfg, device="cpu")
    g_state, d_state = O.synth_nets(c, 101, 202)
    gsd, gp = O.as_trainable(g_state)
    dsd, dp = O.as_trainable(d_state)
    ts = O.TrainState(gsd, dsd, gp, dp, c)
    h, w = c["resolution"], c["resolution"] * c["H_base"]
    x = O.synth_event(sensors, h, w, 303)
    y = torch.arange(sensors)
    gen = torch.Generator().manual_seed(1)
    noise = {}
    for ph in "dg":
        noise["z_" + ph] = torch.randn(sensors, 128, generator=gen)
        noise["rdof_" + ph] = torch.randn(sensors, 4, generator=gen)
        noise["aug_" + ph] = O.diffaug_draws(sensors, h, w, generator=gen)
    t0 = time.time()
    O.train_step(ts, x, y, noise, itr=1)
    dt = time.time() - t0
    return {"value": 1.0 / (dt * 40.0 / sensors), "unit": "events/s", "cores": threads, "kind": "port",
            "sample": f"one oracle train step (fp32, PyTorch CPU) at 256x768 on {sensors} of 40 sensors: {dt:.1f} s, "
                      f"scaled x{40 // sensors} to a full event"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--resolution", type=int, default=256)
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--shape-tags", action="store_true", help="per-kernel timing split by layer shape (tuning aid)")
    args = ap.parse_args()

    import _hip
    import model
    import parallel
    import train_fns
    import utils

    rank, world, local = parallel.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    _hip.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = bench_config()
    cfg["resolution"] = args.resolution
    cfg["hip_graph"] = not args.no_graph
    force_dp = bool(os.environ.get("IEAGAN_FORCE_DP")) and dist.is_initialized()     # 1-rank rehearsal of the DP path
    if args.resolution != 256:
        cfg["H_base"] = 1
    utils.seed_rng(cfg["seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        G = model.Generator(**cfg).to(dev)
        D = model.Discriminator(**cfg).to(dev)
        G_ema = model.Generator(**dict(cfg, skip_init=True, no_optim=True)).to(dev)
        ema = utils.apply_ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    GD = model.G_D(G, D)
    if world > 1 or force_dp:
        parallel.set_context(parallel.GradSync(overlap=True, force=force_dp))
        for net in (G, D):
            net._prepare()
            parallel.broadcast_flat(net._arena.flat)
    z_, y_ = utils.prepare_z_y(40, G.dim_z, cfg["n_classes"], device=dev)
    state = {"itr": 0}
    train = train_fns.GAN_training_function(G, D, GD, z_, y_, ema, state, cfg, dev)
    h, w = cfg["resolution"], cfg["resolution"] * cfg["H_base"]
    x = synth_event(40, h, w, cfg["seed"] + rank).to(dev)
    y = torch.arange(40, device=dev)
    G.train(); D.train(); G_ema.train()

    def step():
        state["itr"] += 1
        return train(x, y)

    for _ in range(max(args.warmup, 3 if cfg["hip_graph"] else 0)):    # graph mode: 2 eager steps + capture
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    timing = not args.no_kernel_timing
    eager_timing = timing and not cfg["hip_graph"]
    if eager_timing:            # eager mode: HIP events bracket every launch inside the timed region itself
        _hip.call("ieagan_prof_reset")
        _hip.prof_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if eager_timing:
        _hip.prof_enable(False)
    prof_steps = args.steps
    if timing and cfg["hip_graph"]:
        # A replayed HIP graph cannot carry per-launch event pairs: time the SAME launches (same shapes,
        # same kernels) in eager steps right after the timed region.
        prof_steps = min(args.steps, 3)
        _hip.call("ieagan_prof_reset")
        _hip.prof_enable(2 if args.shape_tags else 1)
        for _ in range(prof_steps):
            state["itr"] += 1
            train.step_tensor(x, y)
        torch.cuda.synchronize()
        _hip.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    res = {"metric": "events/sec (40x250x768) per G+D train step", "value": world * args.steps / dt, "unit": "events/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[1]: 1 event (40x1x{h}x{w}) per GPU per step, hinge loss only, RRM on, "
                                  "DiffAugment on, ortho reg + Adam + EMA in the timed region, fp32 master weights",
                      "events_per_gpu_per_step": 1, "parallelism": f"dp{world}", "clip_norm": cfg["clip_norm"],
                      "hip_graph": bool(cfg["hip_graph"])},
           "step_tflops_algorithmic": STEP_GFLOP * 1e-3 * world * args.steps / dt if args.resolution == 256 else None,
           "losses_last_step": out}
    if timing:
        recs = sorted(_hip.prof_collect(), key=lambda r: -r["ms"])
        total = sum(r["ms"] for r in recs) or 1.0
        ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)          # ~312 FLOP/B
        pmc = {}
        try:      # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, corrected)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")))["families"]
        except Exception:
            pass

        def roofline_of(r):
            ai = r["flops"] / r["bytes"] if r["bytes"] else float("inf")
            if r["flops"] > 0 and ai >= ridge / 4:      # near / above the ridge: price against the MFMA peak
                ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
                out = {"kernel": r["name"], "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                       "frac": ach / PEAK_BF16_TFLOPS}
            else:
                ach = r["bytes"] / (r["ms"] * 1e-3) / 1e9
                out = {"kernel": r["name"], "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": ach / PEAK_HBM_GBS}
            fam = pmc.get(r["name"].split(" ")[0])
            out.update(traffic=fam["hbm_bytes_per_launch"] if fam else None,
                       algorithmic_bytes_per_launch=r["bytes"] / max(r["launches"], 1),
                       algorithmic_flops_per_launch=r["flops"] / max(r["launches"], 1), launches=r["launches"],
                       avg_launch_ms=r["ms"] / max(r["launches"], 1), share_of_kernel_time=r["ms"] / total,
                       arithmetic_intensity=ai if ai != float("inf") else None)
            return out

        res["roofline"] = roofline_of(recs[0])
        mf = [r for r in recs if r["flops"] > 0 and r["bytes"] and r["flops"] / r["bytes"] >= ridge / 4]
        if mf:
            res["roofline_conv_mfma"] = roofline_of(mf[0])
        res["kernels"] = [{"name": r["name"], "launches": r["launches"], "ms_per_step": r["ms"] / prof_steps,
                           "tflops": (r["flops"] / (r["ms"] * 1e-3) / 1e12) if r["flops"] else None,
                           "GBs": (r["bytes"] / (r["ms"] * 1e-3) / 1e9) if r["bytes"] else None} for r in recs[:(60 if args.shape_tags else 12)]]
        res["kernel_ms_per_step_total"] = total / prof_steps
        res["kernel_timing"] = ("HIP events around every launch inside the timed region" if eager_timing else
                                f"HIP events around every launch in {prof_steps} eager steps run right after the timed graph replays")
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg)
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
