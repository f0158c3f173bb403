#!/usr/bin/env python3
"""Training entry point of the MI355X IEA-GAN path.

Keeps the reference's control flow (reference ``train.py``: run 22-247, main 250-259, epoch loop 158-173):
defaults <- ``config.json`` (optional) <- command line, build G / D / G_ema / G_D, ``utils.prepare_z_y``,
``train_fns.GAN_training_function``, then per iteration ``G.train(); D.train(); metrics = train(x, y)``,
metric / singular-value logging and periodic checkpoints in the reference's file layout
(``<outputroot>/<run_name>/{weights,logs,samples}``).

Data: ``--dataroot DIR`` with one ``*.npy`` per event (uint8 ``[40, 250, 768]`` detector images, or float32
already in [-1, 1]); the pad(3 rows) -> lognorm255 -> +4e-3 U dequantisation -> [-1, 1] chain of the
reference's loader (utils/dataloader.py:59-76, utils/norm.py:8-19) runs on the GPU.  ``--synthetic N`` trains on
N generated events (no files).  Any default of ``defaults.default_config()`` can be overridden as ``--key value``.
Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N train.py ...`` (events are sharded
over the ranks -- every rank takes ``len(events) // N`` per epoch, so all ranks issue the same collectives --
gradients all-reduced over RCCL).
"""
from __future__ import annotations

import argparse
import glob
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model          # noqa: E402
import parallel       # noqa: E402
import train_fns      # noqa: E402
import utils          # noqa: E402
from defaults import default_config      # noqa: E402


def parse(argv):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default=None, help="json file with overrides (the reference's config.json works)")
    ap.add_argument("--dataroot", default=None)
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many generated events instead of files")
    ap.add_argument("--outputroot", default="runs")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--max_iters", type=int, default=0, help="stop after this many iterations (0: run num_epochs)")
    args, rest = ap.parse_known_args(argv)
    cfg = default_config()
    if args.config:
        cfg.update(json.load(open(args.config)))
    it = iter(rest)
    for tok in it:        # generic --key value overrides, typed after the default
        if not tok.startswith("--"):
            raise SystemExit(f"unexpected argument {tok}")
        key = tok[2:]
        if key not in cfg:
            raise SystemExit(f"unknown option --{key}")
        cur = cfg[key]
        if isinstance(cur, bool):
            nxt = next(it, "true")
            cfg[key] = nxt.lower() in ("1", "true", "yes")
        else:
            val = next(it)
            cfg[key] = None if val.lower() in ("none", "null") else (type(cur)(val) if cur is not None else json.loads(val))
    cfg.update(device=args.device, dataroot=args.dataroot, outputroot=args.outputroot, synthetic=args.synthetic,
               max_iters=args.max_iters)
    cfg.setdefault("load_weights", "")
    return cfg


def synthetic_event(n, h, w, seed):
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    hit = rng.random((n, h, w)) < 0.01
    adc = np.where(hit, rng.uniform(8, 255, (n, h, w)), 0.0)
    return adc.astype(np.uint8)


def to_network_range(ev: torch.Tensor, res_h: int) -> torch.Tensor:
    """uint8 [40, 250(+), W] ADC counts -> float32 [40, 1, res_h, W] in [-1, 1]: the reference's pad / lognorm255 / noise /
    normalise chain as one HIP kernel (``utils.ingest_event``); float events are taken as already normalised."""
    if ev.dtype == torch.uint8:
        pad = res_h - ev.shape[1]
        if pad < 0 or pad % 2:
            raise ValueError(f"event height {ev.shape[1]} cannot be padded symmetrically to {res_h}")
        return utils.ingest_event(ev, pad=pad // 2)
    return ev.float().unsqueeze(1).contiguous()


def run(cfg):
    rank, world, local = parallel.init_from_env()
    if world > 1:
        parallel.set_context(parallel.GradSync(overlap=True))
    dev = torch.device(cfg["device"], local) if cfg["device"] == "cuda" else torch.device(cfg["device"])
    utils.seed_rng(cfg["seed"] + rank)
    G = model.Generator(**cfg).to(dev)
    D = model.Discriminator(**cfg).to(dev)
    G_ema = ema = None
    if cfg["ema"]:
        G_ema = model.Generator(**dict(cfg, skip_init=True, no_optim=True)).to(dev)
        ema = utils.apply_ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    GD = model.G_D(G, D)
    state = {"itr": 0, "epoch": 0, "save_num": 0, "save_best_num": 0, "best_FID": 999999}
    name = cfg["run_name"]
    run_dir = os.path.join(cfg["outputroot"], name)          # reference layout: <outputroot>/<run_name>/{weights,logs,samples}
    lroot = os.path.join(run_dir, "logs")
    if rank == 0:
        for sub in ("weights", "logs", "samples"):
            os.makedirs(os.path.join(run_dir, sub), exist_ok=True)
    if cfg["resume"]:
        utils.load_weights(G, D, state, cfg, cfg.get("load_weights") or None, G_ema, load_optim=cfg["load_optim"])
        for net in (G, D):              # fast-forward the lr schedules to the restored epoch (reference train.py:91-94)
            if net.lr_sched is not None:
                net.lr_sched.step(state["epoch"])
    if world > 1:
        # every replica starts from rank 0's state: G, D and the EMA copy (one flat arena each -> one broadcast each)
        for net in (G, D):
            net._prepare()
            parallel.broadcast_flat(net._arena.flat)
        if G_ema is not None:
            parallel.broadcast_flat(ema._arenas()[1].flat)
    utils.count_parameters(G)
    utils.count_parameters(D)
    if rank == 0:
        utils.write_metadata(cfg, state)
    h, w = cfg["resolution"], cfg["resolution"] * cfg["H_base"]
    if cfg["synthetic"]:
        events = [synthetic_event(cfg["n_classes"], h - 6, w, cfg["seed"] + i) for i in range(cfg["synthetic"])]
    else:
        files = sorted(glob.glob(os.path.join(cfg["dataroot"] or "", "*.npy")))
        if not files:
            raise SystemExit("no *.npy events under --dataroot (or use --synthetic N)")
        events = files
    # Every rank must issue the same number of collectives: each takes exactly len(events) // world events per epoch
    # (the remainder is dropped, like DataLoader(drop_last=True)); fewer events than ranks is an error raised before
    # any collective is issued.
    per_rank = parallel.steps_per_rank(len(events), world)
    if per_rank == 0:
        raise SystemExit(f"{len(events)} event(s) cannot be sharded over {world} ranks (need at least one per rank)")
    mine = [events[i] for i in parallel.shard_events(len(events), rank, world)][:per_rank]
    # E events per step (events_per_step): every step consumes E events of this rank's shard, batched on the leading
    # dimension; a trailing group of fewer than E events is dropped (like DataLoader(drop_last=True))
    E = max(int(cfg.get("events_per_step", 1) or 1), 1)
    if len(mine) < E:
        raise SystemExit(f"events_per_step={E} but this rank holds only {len(mine)} event(s)")
    z_, y_ = utils.prepare_z_y(max(cfg["G_batch_size"], cfg["batch_size"]) * E, G.dim_z, cfg["n_classes"], device=dev,
                               z_dist=cfg["z_dist"], threshold=cfg["truncated_threshold"])
    train = train_fns.GAN_training_function(G, D, GD, z_, y_, ema, state, cfg, dev)
    y = torch.arange(cfg["n_classes"], device=dev).repeat(E)
    log = open(os.path.join(lroot, f"metrics_rank{rank}.jsonl"), "a") if rank == 0 else None
    t0 = time.time()

    def checkpoint():
        parallel.quiesce()              # side-stream updates have landed before the arenas are copied to the host
        if rank == 0:
            utils.save_weights(G, D, state, cfg, None, G_ema)

    stop = False
    for epoch in range(state["epoch"], cfg["num_epochs"]):
        order = np.random.permutation(len(mine)) if cfg["shuffle"] else np.arange(len(mine))
        for s0 in range(0, len(order) - E + 1, E):
            state["itr"] += 1
            G.train()
            D.train()
            if G_ema is not None:
                G_ema.train()
            evs = [mine[i] if isinstance(mine[i], np.ndarray) else np.load(mine[i]) for i in order[s0:s0 + E]]
            x = torch.cat([to_network_range(torch.from_numpy(ev), h).to(dev) for ev in evs]) if E > 1 else \
                to_network_range(torch.from_numpy(evs[0]), h).to(dev)
            metrics = train(x, y)
            if log is not None:
                rec = dict(itr=state["itr"], **metrics)
                if cfg["sv_log_interval"] > 0 and state["itr"] % cfg["sv_log_interval"] == 0:
                    rec.update(utils.get_singular_values(G, "G"))
                    rec.update(utils.get_singular_values(D, "D"))
                log.write(json.dumps(rec) + "\n")
                log.flush()
            if state["itr"] % cfg["log_interval"] == 0 and rank == 0:
                print(f"itr {state['itr']}  {(time.time() - t0) / state['itr']:.3f} s/itr  " +
                      "  ".join(f"{k} {v:.4f}" for k, v in metrics.items()))
            if state["itr"] % cfg["save_every"] == 0:
                checkpoint()
            if cfg["max_iters"] and state["itr"] >= cfg["max_iters"]:
                stop = True
                break
        if stop:                        # max_iters reached mid-epoch: the partial epoch neither counts nor steps the schedules
            break
        state["epoch"] += 1
        for net in (G, D):
            if net.lr_sched is not None:
                net.lr_sched.step()
    checkpoint()
    if rank == 0:
        print(f"done: {state['itr']} iterations, weights under {os.path.join(run_dir, 'weights')}")
    train.close()                       # graphs / side streams released before the process group and the interpreter go away
    parallel.shutdown()                 # barrier + destroy_process_group: no rank leaves while another is in a collective
    return state


def main(argv=None):
    run(parse(sys.argv[1:] if argv is None else argv))


if __name__ == "__main__":
    main()
