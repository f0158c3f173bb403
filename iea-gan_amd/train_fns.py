"""The IEA-GAN train step (one D update + one G update) for MI355X.

Entry point kept from reference ``train_fns.py:20-206``:
``GAN_training_function(G, D, GD, z_, y_, ema, state_dict, config, device) -> train(x, y) -> dict`` with the
keys ``G_loss, D_loss_real, D_loss_fake, unif_loss_d, iea_loss``.  Phases, loss composition, call order of
the fake / real discriminator passes and the optimiser placement follow the reference, including its quirk
that ``G.optim.step()`` only runs when ``clip_norm`` is not None.  Differences, all in
places where the reference raises: loss terms that are switched off are reported as 0.0 instead of an
UnboundLocalError (9-Q2), and consistency regularisation with ``split_D`` runs a third discriminator pass on
the augmented real event (9-Q3).  The five losses are read back with ONE host synchronisation.

With ``torch.distributed`` initialised the step is data parallel (``parallel.py``): D's gradient all-reduce
and Adam update run on a side stream underneath the G-phase generator forward.
"""
from __future__ import annotations

import torch

import loss
import ops
import parallel
import utils
from cr_diff_aug import CR_DiffAug


def dummy_training_function():
    def train(x, y):
        return {}
    return train


def _clip(net, max_norm):
    """clip_grad_norm_ on the flat gradient arena, without a host round trip."""
    ar = net.__dict__.get("_arena")
    if ar is not None and ar.grads_attached():
        total = torch.linalg.vector_norm(ar.grad)
        ar.grad.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
    else:
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm)


def GAN_training_function(G, D, GD, z_, y_, ema, state_dict, config, device):
    if config["pos_collected_numerator"]:
        raise NotImplementedError("pos_collected_numerator=True is not part of the MI355X path (reference default: False)")
    bs = config["batch_size"]
    sync = parallel.get_context()
    if sync is not None:
        # the main stream must not evaluate D before its (side-stream) update has landed
        D.register_forward_pre_hook(lambda m, inp: sync.wait("D"))
        G.register_forward_pre_hook(lambda m, inp: sync.wait("G"))

    def finish(net, key, clip, do_step):
        def then():
            if clip is not None:
                _clip(net, clip)
            if do_step:
                net.optim.step()
        if sync is not None and net.__dict__.get("_arena") is not None and net._arena.grad is not None:
            sync.reduce_then(key, net._arena.grad, then)
        else:
            then()

    def step(x, y, noise=None):
        """One D update + one G update; returns the five losses as ONE device tensor (no host sync).
        ``noise`` (optional, parity tests): {'z_d','rdof_d','aug_d','z_g','rdof_g','aug_g'} replaces the
        draws of the two generator passes; None = sample exactly where the reference samples."""
        import diff_aug as _da

        def sample(phase):
            if noise is None:
                z_.sample_()
                return
            z_.copy_(noise["z_" + phase].to(z_.device))
            G.__dict__["_next_rdof"] = noise["rdof_" + phase].to(z_.device)
            if config["diff_aug"]:
                _da.NEXT_DRAWS.append(noise["aug_" + phase])

        G.optim.zero_grad()
        D.optim.zero_grad()
        x_aug = CR_DiffAug(x) if config["Con_reg"] else None
        xs, ys = torch.split(x, bs), torch.split(y, bs)
        xa = torch.split(x_aug, bs) if x_aug is not None else None
        counter = 0
        zero = torch.zeros((), device=x.device)
        contra = config["conditional_strategy"] == "Contra"
        t = 1.0
        if config["toggle_grads"]:
            utils.toggle_grad(D, True)
            utils.toggle_grad(G, False)

        # ------------------------------------------------------------------ D phase
        D_loss_real = D_loss_fake = unif_loss_d = iea_loss = zero
        cls_embed_real = None
        for _ in range(config["num_D_steps"]):
            D.optim.zero_grad()
            for _ in range(config["num_D_accumulations"]):
                sample("d")
                joint_aug = config["Con_reg"] and not config["split_D"]
                outs = GD(z_[:bs], ys[counter], xs[counter], ys[counter], xa[counter] if joint_aug else None, contra=contra,
                          train_G=False, split_D=config["split_D"], diff_aug=config["diff_aug"])
                aug_out = None
                if contra:
                    if len(outs) == 8:
                        *outs, cls_embed_real_aug, D_real_aug = outs
                        aug_out = (cls_embed_real_aug, D_real_aug)
                    _, _, D_fake, cls_proxies_real, cls_embed_real, D_real = outs
                    if config["Con_reg"] and aug_out is None:          # split_D: third pass on the augmented event
                        _, cls_embed_real_aug, D_real_aug = D(xa[counter], ys[counter])
                        aug_out = (cls_embed_real_aug, D_real_aug)
                else:
                    if len(outs) == 3:
                        D_fake, D_real, D_real_aug = outs
                        aug_out = (None, D_real_aug)
                    else:
                        D_fake, D_real = outs
                        if config["Con_reg"]:
                            aug_out = (None, D(xa[counter], ys[counter]))
                # every D-phase loss term in ONE fused launch (value + gradient): hinge real/fake, 2C, uniformity
                use_c = contra and config["contra_lambda"] != 0
                use_u = contra and config["Uniformity_loss"]
                D_loss, terms = ops.loss_block(dfake=D_fake, dreal=D_real, e=cls_embed_real if (use_c or use_u) else None,
                                               p=cls_proxies_real if use_c else None, w_hinge_real=1.0, w_hinge_fake=1.0,
                                               w_contra=config["contra_lambda"] if use_c else 0.0,
                                               w_unif=config["unif_lambda"] if use_u else 0.0, temperature=t)
                D_loss_real, D_loss_fake = terms[1], terms[2]
                if use_u:
                    unif_loss_d = terms[5]
                if aug_out is not None:
                    consistency = loss.l2_loss(D_real, aug_out[1])
                    if aug_out[0] is not None:
                        consistency = consistency + loss.l2_loss(cls_embed_real, aug_out[0])
                    D_loss = D_loss + config["cr_lambda"] * consistency
                with ops.direct_grads():
                    (D_loss / float(config["num_D_accumulations"])).backward()
            if config["D_ortho"] > 0.0:
                utils.ortho(D, config["D_ortho"])
            finish(D, "D", config["clip_norm"], True)

        if config["toggle_grads"]:
            utils.toggle_grad(D, False)
            utils.toggle_grad(G, True)
        G.optim.zero_grad()

        # ------------------------------------------------------------------ G phase
        for _ in range(config["num_G_accumulations"]):
            sample("g")
            if contra:
                cls_proxies_fake, cls_embed_fake, D_fake = GD(z_, ys[counter], x_aug=None, contra=True, train_G=True,
                                                             split_D=config["split_D"], diff_aug=config["diff_aug"])
                use_c = config["contra_lambda"] != 0
                use_i = bool(config["IEA_loss"])
                use_u = use_i and bool(config["Uniformity_loss"])      # nested under IEA_loss, as in the reference (:171-178)
                need_e = use_c or use_i or use_u
                G_loss, terms = ops.loss_block(dfake=D_fake, e=cls_embed_fake if need_e else None,
                                               p=cls_proxies_fake if use_c else None,
                                               er=cls_embed_real.detach() if use_i else None, w_hinge_gen=1.0,
                                               w_contra=config["contra_lambda"] if use_c else 0.0,
                                               w_unif=config["unif_lambda"] if use_u else 0.0,
                                               w_iea=config["IEA_lambda"] if use_i else 0.0, temperature=t)
                if use_i:
                    iea_loss = terms[6]
            else:
                D_fake = GD(z_, y_, x_aug=None, contra=False, train_G=True, split_D=config["split_D"],
                            diff_aug=config["diff_aug"])
                G_loss = loss.loss_hinge_gen(D_fake.reshape(-1))
            G_loss = G_loss / float(config["num_G_accumulations"])
            with ops.direct_grads():
                G_loss.backward()
            counter += 1

        if config["G_ortho"] > 0.0:
            # in data-parallel runs the ortho term is deterministic in W: add it after the gradient mean
            if sync is not None:
                sync.reduce_then("G", G._arena.grad, None)
                sync.wait("G")
            utils.ortho(G, config["G_ortho"], blacklist=[p for p in G.shared.parameters()])
            if config["clip_norm"] is not None:     # the reference steps G only inside this branch (9-Q1)
                _clip(G, config["clip_norm"])
                G.optim.step()
        else:
            finish(G, "G", config["clip_norm"], config["clip_norm"] is not None)

        if config["ema"]:
            if sync is not None:
                sync.wait("G")
            ema.update(state_dict["itr"])
        return torch.stack([v.detach().float().reshape(()) for v in (G_loss, D_loss_real, D_loss_fake, unif_loss_d, iea_loss)])

    keys = ("G_loss", "D_loss_real", "D_loss_fake", "unif_loss_d", "iea_loss")
    graph = {"g": None, "calls": 0, "x": None, "y": None, "out": None}
    use_graph = bool(config.get("hip_graph", False)) and sync is None

    def train(x, y, noise=None):
        """The reference's ``train(x, y) -> dict of 5 python floats`` (ONE host sync, at the end).

        With ``config['hip_graph']`` (single-GPU): after two eager warm-up iterations the whole step --
        both generator passes, three discriminator passes, both backwards, ortho, Adam, EMA: a few thousand
        launches -- is captured once into a HIP graph and replayed, which removes the Python / launch
        overhead from the loop.  Everything step-dependent (Adam step counter, lr, EMA decay, RNG offsets)
        lives in device memory, so the captured graph stays valid."""
        if not use_graph or noise is not None:
            return dict(zip(keys, step(x, y, noise).tolist()))
        if graph["g"] is None:
            if graph["calls"] < 2:
                graph["calls"] += 1
                return dict(zip(keys, step(x, y).tolist()))
            graph["x"], graph["y"] = x.clone(), y.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            ops.reset_zero_pool()              # every zero-filled scratch chunk of the step must be filled INSIDE the graph
            with torch.cuda.graph(g):
                graph["out"] = step(graph["x"], graph["y"])
            ops.reset_zero_pool()              # ... and eager code must never carve from graph-owned memory
            graph["g"] = g
        graph["x"].copy_(x)
        graph["y"].copy_(y)
        for net in (G, D):
            net.optim.push_hyper()
        if config["ema"]:
            ema.push_decay(state_dict["itr"])
        graph["g"].replay()
        return dict(zip(keys, graph["out"].tolist()))

    train.step_tensor = step
    return train
