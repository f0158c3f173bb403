"""Shared parity harness: HIP path (product modules on cuda) against the CPU oracle on identical
synthetic weights / noise.  Test infrastructure (imports oracle/); used by tests/ and smoke()."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "iea-gan_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ieagan_oracle as O  # noqa: E402


def make_cfg(**over):
    from defaults import default_config
    cfg = default_config()
    cfg.update(device="cuda", ema=False)
    cfg.update(over)
    return cfg


def build_product(cfg, g_state, d_state, device):
    import model
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        G = model.Generator(**dict(cfg, skip_init=True)).to(device)
        D = model.Discriminator(**dict(cfg, skip_init=True)).to(device)
    G.load_state_dict(g_state)
    D.load_state_dict(d_state)
    G.train()
    D.train()
    return G, D


def make_noise(n, res_h, res_w, seed, rdof_dim=4, dim_z=128):
    gen = torch.Generator().manual_seed(seed)
    noise = {}
    for ph in "dg":
        noise["z_" + ph] = torch.randn(n, dim_z, generator=gen)
        noise["rdof_" + ph] = torch.randn(n, rdof_dim, generator=gen)
        noise["aug_" + ph] = O.diffaug_draws(n, res_h, res_w, generator=gen)
    return noise


def rel_l2(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


def cosine(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-20))


def forward_parity(resolution=64, H_base=1, device="cuda:0", n=40, seed=5, z=None, rdof=None, **cfg_over):
    """G(z,y) then D(G_z, y) in training mode: outputs + updated buffers vs the oracle."""
    cfg = make_cfg(resolution=resolution, H_base=H_base, n_classes=max(n, 40), **cfg_over)
    g_state, d_state = O.synth_nets(cfg, 101, 202)
    G, D = build_product(cfg, g_state, d_state, device)
    gen = torch.Generator().manual_seed(seed)
    y = torch.arange(n)
    if z is None:
        z, rdof = torch.randn(n, 128, generator=gen), torch.randn(n, 4, generator=gen)
    with torch.no_grad():
        gz = G(z.to(device), y.to(device), rdof=rdof.to(device))
        pr, em, do = D(gz, y.to(device))
        torch.cuda.synchronize()
        gsd = {k: v.clone() for k, v in g_state.items()}
        dsd = {k: v.clone() for k, v in d_state.items()}
        t0 = time.time()
        gz_o = O.generator(gsd, cfg, z, y, rdof, True)
        pr_o, em_o, do_o = O.discriminator(dsd, cfg, gz.float().cpu(), y, True)
        t_cpu = time.time() - t0
    rep = {"G_rel_l2": rel_l2(gz, gz_o), "G_maxabs": float((gz.cpu() - gz_o).abs().max()),
           "D_out_rel_l2": rel_l2(do, do_o), "D_embed_rel_l2": rel_l2(em, em_o), "D_proxy_rel_l2": rel_l2(pr, pr_o),
           "G_u_linear": rel_l2(G.state_dict()["linear.u0"], gsd["linear.u0"]),
           "G_bn_mean": rel_l2(G.state_dict()["blocks.0.0.bn1.stored_mean"], gsd["blocks.0.0.bn1.stored_mean"]),
           "G_bn_var_last": rel_l2(G.state_dict()["output_layer.0.stored_var"], gsd["output_layer.0.stored_var"]),
           "D_u_conv": rel_l2(D.state_dict()["blocks.0.0.conv2.u0"], dsd["blocks.0.0.conv2.u0"]), "oracle_s": t_cpu}
    return rep, (G, D, gz, gz_o)


# stated bf16 tolerances (CPU bf16-autocast of the reference itself gives G 3.0e-2,
# D logits 1.6e-2, embeddings 9e-3 rel-L2 at this geometry)
TOL = {"G_rel_l2": 5e-2, "D_out_rel_l2": 5e-2, "D_embed_rel_l2": 3e-2, "D_proxy_rel_l2": 1e-4, "loss": 5e-2,
       "grad_cos": 0.97}


def _flat_grads(net, grads):
    ar = net._arena
    f = torch.zeros(ar.n_param)
    names = [k for k, _ in net.named_parameters()]
    for (p, o, cnt), k in zip(ar.param_slices, names):
        f[o:o + cnt] = grads[k].reshape(-1)
    return f


def step_parity(resolution=64, H_base=1, device="cuda:0", verbose=False, n=40, events=1, state_check=False, oracle_bf16=False,
                inputs=None, y_g=None, sn_warm=0, hip_only=False, **cfg_over):
    """One full train(x, y) on the HIP path vs the oracle on identical weights / noise: the 5 losses and the flat G / D
    gradients (cosine, rel-L2); with ``state_check`` also the post-step state (parameters after Adam, spectral-norm
    ``u0`` / ``sv0``, BatchNorm running statistics).  ``n`` < 40 runs a sub-event of the first n sensors (full
    256x768 resolution stays affordable for the CPU oracle), ``events`` > 1 the E-events-per-step path (configs[3]).
    ``oracle_bf16``: additionally run the oracle with bf16-rounded conv operands / outputs (``O.ROUND_BF16``) and report
    how far THAT moves the same quantities -- the rounding-noise floor the tolerances are stated against.
    ``hip_only``: no oracle run -- the HIP path's losses and flat gradients alone, for comparisons of two HIP runs with each other."""
    import model
    import train_fns
    import utils
    over = dict(clip_norm=1e9)
    over.update(cfg_over)
    cfg = make_cfg(resolution=resolution, H_base=H_base, batch_size=n, events_per_step=events, **over)
    hh, ww = resolution, resolution * H_base
    g_state, d_state = O.synth_nets(cfg, 101, 202)
    if sn_warm:
        # converge the spectral-norm power iteration first (``sn_warm`` training-mode oracle passes advance u0 / sv0 in place): a fresh
        # u0 is random, its first iterates move sigma by O(1) -- a run that is only allowed to differ in WHICH iterate a pass sees
        # (data-parallel real-first order) has to be compared where consecutive iterates agree, as they do after a few steps
        gw = torch.Generator().manual_seed(17)
        with torch.no_grad():
            for _ in range(sn_warm):
                zz, rr = torch.randn(n, 128, generator=gw), torch.randn(n, 4, generator=gw)
                gz = O.generator(g_state, cfg, zz, torch.arange(n), rr, True)
                O.discriminator(d_state, cfg, gz, torch.arange(n), True)
    G, D = build_product(cfg, g_state, d_state, device)
    GD = model.G_D(G, D)
    z_, y_ = utils.prepare_z_y(n * events, G.dim_z, cfg["n_classes"], device=device)
    if y_g is not None:                 # projection strategy: the label vector the G phase feeds (train_fns.py:153), explicit
        y_ = y_g.to(device)
    train = train_fns.GAN_training_function(G, D, GD, z_, y_, None, {"itr": 1}, cfg, device)
    y = torch.arange(n)
    if inputs is not None:              # events + draws of a committed fixture
        xs, noises = inputs
        if y_g is not None:
            noises = [dict(nz, y_g=y_g) for nz in noises]
    else:
        xs = [O.synth_event(n, hh, ww, 303 + e) for e in range(events)]
        noises = [make_noise(n, hh, ww, 909 + e) for e in range(events)]
        if cfg.get("Con_reg"):
            for e, nz in enumerate(noises):
                nz["cr"] = O.cr_draws(n, hh, ww, generator=torch.Generator().manual_seed(77 + e))
    if events == 1:
        out = train(xs[0].to(device), y.to(device), noise=noises[0])
    else:
        out = train(torch.cat(xs).to(device), y.repeat(events).to(device), noise=noises)
    torch.cuda.synchronize()
    g_grad = G._arena.grad.clone().cpu()
    d_grad = D._arena.grad.clone().cpu()
    if hip_only:
        return {"losses": out, "G_grad": g_grad, "D_grad": d_grad}

    def run_oracle():
        gsd, gp = O.as_trainable(g_state)
        dsd, dp = O.as_trainable(d_state)
        ts = O.TrainState(gsd, dsd, gp, dp, cfg)
        ref = O.train_step_events(ts, xs, y, noises, itr=1)
        og, od = ts.last_grads
        return ref, _flat_grads(G, og), _flat_grads(D, od), gsd, dsd

    ref, og_f, od_f, gsd, dsd = run_oracle()
    rep = {"losses": out, "ref_losses": ref, "G_grad_cos": cosine(g_grad, og_f), "D_grad_cos": cosine(d_grad, od_f),
           "G_grad_rel": rel_l2(g_grad, og_f), "D_grad_rel": rel_l2(d_grad, od_f)}
    ok = all(abs(out[k] - ref[k]) <= TOL["loss"] * max(1.0, abs(ref[k])) for k in ref)
    ok = ok and rep["G_grad_cos"] >= TOL["grad_cos"] and rep["D_grad_cos"] >= TOL["grad_cos"]
    if state_check:
        st = {}
        for name, net, osd, init in (("G", G, gsd, g_state), ("D", D, dsd, d_state)):
            sd = net.state_dict()
            u = [k for k in sd if k.endswith(".u0")]
            st[name + "_u0_rel_max"] = max(rel_l2(sd[k], osd[k]) for k in u)
            st[name + "_sv0_rel_max"] = max(rel_l2(sd[k], osd[k]) for k in sd if k.endswith(".sv0"))
            rs = [k for k in sd if k.endswith("stored_mean") or k.endswith("stored_var")]
            if rs:
                st[name + "_bn_running_rel_max"] = max(rel_l2(sd[k], osd[k]) for k in rs)
            params = [k for k, _ in net.named_parameters()]
            upd = torch.cat([(sd[k].detach().cpu() - init[k]).reshape(-1) for k in params])
            upd_o = torch.cat([(osd[k].detach() - init[k]).reshape(-1) for k in params])
            # beta1 = 0: the first Adam update is lr * g / (|g| + eps), i.e. ~lr * sign(g) -- sign agreement is the metric
            big = upd_o.abs() > 0.5 * float(upd_o.abs().max())
            st[name + "_update_cos"] = cosine(upd, upd_o)
            st[name + "_update_sign_agree"] = float((torch.sign(upd[big]) == torch.sign(upd_o[big])).float().mean())
            st[name + "_param_rel"] = rel_l2(torch.cat([sd[k].detach().cpu().reshape(-1) for k in params]),
                                             torch.cat([osd[k].detach().reshape(-1) for k in params]))
        rep["state"] = st
    if oracle_bf16:
        O.ROUND_BF16 = True
        try:
            ref_b, og_b, od_b, _, _ = run_oracle()
        finally:
            O.ROUND_BF16 = False
        rep["bf16_floor"] = {"G_grad_rel": rel_l2(og_b, og_f), "D_grad_rel": rel_l2(od_b, od_f), "G_grad_cos": cosine(og_b, og_f),
                             "D_grad_cos": cosine(od_b, od_f), "loss_rel": {k: abs(ref_b[k] - ref[k]) / max(1.0, abs(ref[k])) for k in ref},
                             "hip_vs_bf16_oracle_G_grad_rel": rel_l2(g_grad, og_b), "hip_vs_bf16_oracle_D_grad_rel": rel_l2(d_grad, od_b)}
    rep["ok"] = bool(ok)
    if verbose:
        print(rep)
    return rep
