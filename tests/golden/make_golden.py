"""Generate the golden fixtures under tests/golden/  (development container only).

Imports the *reference* (read-only at /root/reference, SURVEY Appendix A recipe), feeds it and the
oracle (oracle/ieagan_oracle.py) identical synthetic weights / noise, asserts that they agree, and
writes small input/output vectors.  The reference never travels: only this script and the data
files it emits are committed.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import functools
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(1, "/root/reference")
for _n in ["boost_histogram", "cv2", "torchvision", "torchvision.utils", "torchvision.transforms",
           "torchvision.datasets", "seaborn", "mycleanfid", "mycleanfid.fid"]:
    _m = types.ModuleType(_n)
    _m.__path__ = []
    sys.modules[_n] = _m
sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
sys.modules["mycleanfid"].fid = sys.modules["mycleanfid.fid"]

import ieagan_oracle as O          # noqa: E402
import model as R_model            # noqa: E402  (reference)
import layers as R_layers          # noqa: E402
import RRM as R_RRM                # noqa: E402
import loss as R_loss              # noqa: E402
import diff_aug as R_da            # noqa: E402
import cr_diff_aug as R_cr         # noqa: E402
import utils as R_utils            # noqa: E402
import train_fns as R_train        # noqa: E402

torch.set_num_threads(8)
CFG = json.load(open("/root/reference/config.json"))
CFG["device"] = "cpu"
SN_EPS, BN_EPS = CFG["SN_eps"], CFG["BN_eps"]


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f"  wrote {name}: {sum(a.nbytes for a in out.values()) / 1e3:.1f} kB")


def close(a, b, tol=2e-5, what=""):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1e-6)
    assert err <= tol * max(ref, 1.0), f"{what}: max|diff| {err:.3e} (ref scale {ref:.3e})"
    return err


def load_synth(module, seed, prefix="m"):
    """Fill a reference module with oracle.synth_state numbers; return the oracle-side mapping."""
    spec = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    st = O.synth_state(spec, seed)
    module.load_state_dict(st)
    return {f"{prefix}.{k}" if prefix else k: v.clone() for k, v in st.items()}


def trainable(sd):
    out = {}
    for k, v in sd.items():
        out[k] = v.clone().requires_grad_(not O.is_buffer(k))
    return out


# ------------------------------------------------------------------------------------------------
def contract():
    """State-dict contract at the baseline and the 64x64 plumbing geometry."""
    out = {}
    for tag, over in (("256x768", {}), ("64x64", {"resolution": 64, "H_base": 1})):
        cfg = dict(CFG, **over)
        G, D = R_model.Generator(**cfg), R_model.Discriminator(**cfg)
        for net, name, spec in ((G, "G", O.g_spec(cfg)), (D, "D", O.d_spec(cfg))):
            ref = {k: list(v.shape) for k, v in net.state_dict().items()}
            assert ref == {k: list(s) for k, s in spec.items()}, f"{name} spec mismatch at {tag}"
            params = sorted(k for k, _ in net.named_parameters())
            assert params == sorted(k for k in spec if not O.is_buffer(k))
            out[f"{name}_{tag}"] = {"keys": ref, "params": params,
                                    "n_params": int(sum(p.numel() for p in net.parameters()))}
    json.dump(out, open(os.path.join(HERE, "state_dict_contract.json"), "w"))
    print("  wrote state_dict_contract.json",
          {k: (len(v["keys"]), v["n_params"]) for k, v in out.items()})


def ops():
    torch.manual_seed(0)
    # ---- SNConv2d 3x3 / 1x1, SNLinear, SNEmbedding: fwd, u/sv update, grads incl. sigma term
    for name, mod, x in (
        ("snconv3", R_layers.SNConv2d(32, 48, 3, padding=1, eps=SN_EPS), torch.randn(3, 32, 10, 14)),
        ("snconv1", R_layers.SNConv2d(64, 16, 1, padding=0, eps=SN_EPS), torch.randn(3, 64, 6, 10)),
        ("snlinear", R_layers.SNLinear(132, 128, eps=SN_EPS), torch.randn(40, 132)),
    ):
        sd = trainable(load_synth(mod, 11))
        mod.train()
        x.requires_grad_(True)
        go = torch.randn_like(mod(x).detach())
        mod.load_state_dict({k[2:]: v.detach() for k, v in sd.items()})     # reset u after probe call
        y = mod(x)
        gx, gw, gb = torch.autograd.grad(y, [x, mod.weight, mod.bias], go)
        if y.dim() == 4:
            yo = O.conv(sd, "m", x, True, SN_EPS, mod.padding[0])
        else:
            yo = O.linear(sd, "m", x, True, SN_EPS)
        ox, ow, ob = torch.autograd.grad(yo, [x, sd["m.weight"], sd["m.bias"]], go)
        for a, b, w in ((yo, y, "y"), (ox, gx, "gx"), (ow, gw, "gw"), (ob, gb, "gb"),
                        (sd["m.u0"], mod.u0, "u"), (sd["m.sv0"], mod.sv0, "sv")):
            close(a, b, what=f"{name}.{w}")
        npz(f"op_{name}.npz", x=x, go=go, y=y, gx=gx, gw=gw, gb=gb, u_after=mod.u0, sv_after=mod.sv0)
    emb = R_layers.SNEmbedding(40, 1024, eps=SN_EPS)
    sd = trainable(load_synth(emb, 12))
    emb.train()
    idx = torch.arange(40)
    y = emb(idx)
    yo = O.embedding(sd, "m", idx, True, SN_EPS)
    close(yo, y, what="snembed")
    close(sd["m.u0"], emb.u0, what="snembed.u")
    npz("op_snembed.npz", y=y, u_after=emb.u0, sv_after=emb.sv0)

    # ---- ccbn / bn (training mode: batch statistics, running-stat update)
    which_lin = functools.partial(R_layers.SNLinear, bias=False, eps=SN_EPS)
    cb = R_layers.ccbn(32, 256, which_lin, eps=BN_EPS)
    sd = trainable(load_synth(cb, 13))
    cb.train()
    x = torch.randn(6, 32, 8, 12, requires_grad=True)
    yv = torch.randn(6, 256, requires_grad=True)
    go = torch.randn(6, 32, 8, 12)
    y = cb(x, yv)
    gx, gy, gwg, gwb = torch.autograd.grad(y, [x, yv, cb.gain.weight, cb.bias.weight], go)
    yo = O.ccbn(sd, "m", x, yv, True, BN_EPS, SN_EPS)
    ox, oy, owg, owb = torch.autograd.grad(yo, [x, yv, sd["m.gain.weight"], sd["m.bias.weight"]], go)
    for a, b, w in ((yo, y, "y"), (ox, gx, "gx"), (oy, gy, "gy"), (owg, gwg, "gwg"), (owb, gwb, "gwb"),
                    (sd["m.stored_mean"], cb.stored_mean, "rm"), (sd["m.stored_var"], cb.stored_var, "rv")):
        close(a, b, what=f"ccbn.{w}")
    npz("op_ccbn.npz", x=x, yv=yv, go=go, y=y, gx=gx, gy=gy, gwg=gwg, gwb=gwb,
        mean_after=cb.stored_mean, var_after=cb.stored_var)
    pb = R_layers.bn(32, eps=BN_EPS)
    sd = trainable(load_synth(pb, 14))
    pb.train()
    y = pb(x)
    gx, gg, gb = torch.autograd.grad(y, [x, pb.gain, pb.bias], go)
    yo = O.plain_bn(sd, "m", x, True, BN_EPS)
    ox, og, ob = torch.autograd.grad(yo, [x, sd["m.gain"], sd["m.bias"]], go)
    for a, b, w in ((yo, y, "y"), (ox, gx, "gx"), (og, gg, "gg"), (ob, gb, "gb")):
        close(a, b, what=f"bn.{w}")
    npz("op_bn.npz", x=x, go=go, y=y, gx=gx, gg=gg, gb=gb, mean_after=pb.stored_mean, var_after=pb.stored_var)

    # ---- GBlock (upsample, Cin != Cout) and DBlock (downsample, learnable shortcut)
    conv_g = functools.partial(R_layers.SNConv2d, kernel_size=3, padding=1, eps=SN_EPS)
    bn_g = functools.partial(R_layers.ccbn, which_linear=which_lin, input_size=256, eps=BN_EPS)
    act = torch.nn.ReLU(inplace=True)
    for tag, cin, cout, up in (("up", 64, 32, True), ("same", 64, 64, False)):
        gb_ = R_model.GBlock(cin, cout, conv_g, bn_g, act,
                             functools.partial(F.interpolate, scale_factor=2) if up else None)
        sd = trainable(load_synth(gb_, 15))
        gb_.train()
        x = torch.randn(4, cin, 6, 8, requires_grad=True)
        yv = torch.randn(4, 256, requires_grad=True)
        y = gb_(x, yv)
        go = torch.randn_like(y.detach())
        names = [k for k, _ in gb_.named_parameters()]
        grads = torch.autograd.grad(y, [x, yv] + [p for _, p in gb_.named_parameters()], go)
        yo = O.g_block(sd, "m", x, yv, cin, cout, up, True, CFG)
        ograds = torch.autograd.grad(yo, [x, yv] + [sd["m." + k] for k in names], go)
        close(yo, y, what=f"gblock_{tag}.y")
        for n_, a, b in zip(["x", "yv"] + names, ograds, grads):
            close(a, b, tol=5e-5, what=f"gblock_{tag}.g[{n_}]")
        npz(f"op_gblock_{tag}.npz", x=x, yv=yv, go=go, y=y, gx=grads[0], gy=grads[1],
            **{"gw." + n_: g for n_, g in zip(names, grads[2:])})
    conv_d = functools.partial(R_layers.SNConv2d, kernel_size=3, padding=1, eps=SN_EPS)
    for tag, cin, cout, down, pre in (("down", 32, 64, True, True), ("first", 32, 64, True, False),
                                      ("same", 64, 64, False, True)):
        db = R_model.DBlock(cin, cout, conv_d, True, pre, act, torch.nn.AvgPool2d(2) if down else None)
        sd = trainable(load_synth(db, 16))
        db.train()
        x = torch.randn(4, cin, 8, 12, requires_grad=True)
        y = db(x)
        go = torch.randn_like(y.detach())
        names = [k for k, _ in db.named_parameters()]
        grads = torch.autograd.grad(y, [x] + [p for _, p in db.named_parameters()], go)
        yo = O.d_block(sd, "m", x, cin, cout, down, pre, True, SN_EPS)
        ograds = torch.autograd.grad(yo, [x] + [sd["m." + k] for k in names], go)
        close(yo, y, what=f"dblock_{tag}.y")
        for n_, a, b in zip(["x"] + names, ograds, grads):
            close(a, b, tol=5e-5, what=f"dblock_{tag}.g[{n_}]")
        npz(f"op_dblock_{tag}.npz", x=x, go=go, y=y, gx=grads[0],
            **{"gw." + n_: g for n_, g in zip(names, grads[1:])})

    # ---- non-local attention
    at = R_layers.Attention(64, conv_d)
    sd = trainable(load_synth(at, 17))
    at.train()
    x = torch.randn(2, 64, 8, 12, requires_grad=True)
    y = at(x)
    go = torch.randn_like(y.detach())
    names = [k for k, _ in at.named_parameters()]
    grads = torch.autograd.grad(y, [x] + [p for _, p in at.named_parameters()], go)
    yo = O.nonlocal_attention(sd, "m", x, True, SN_EPS)
    ograds = torch.autograd.grad(yo, [x] + [sd["m." + k] for k in names], go)
    close(yo, y, what="attention.y")
    for n_, a, b in zip(["x"] + names, ograds, grads):
        close(a, b, tol=5e-5, what=f"attention.g[{n_}]")
    npz("op_attention.npz", x=x, go=go, y=y, gx=grads[0], **{"gw." + n_: g for n_, g in zip(names, grads[1:])})

    # ---- RRM, G flavour (nn.Linear, d=128, 2 heads) and D flavour (SNLinear, d=512, 4 heads)
    for tag, dim, heads, ff, wl in (("g", 128, 2, 128, torch.nn.Linear),
                                    ("d", 512, 4, 512, functools.partial(R_layers.SNLinear, eps=SN_EPS))):
        rr = R_RRM.RelationalReasoning(num_layers=1, input_dim=dim, dim_feedforward=ff, which_linear=wl,
                                       num_heads=heads, dropout=0.0, hidden_dim=dim)
        sd = trainable(load_synth(rr, 18))
        rr.train()
        x = torch.randn(1, 40, dim, requires_grad=True)
        y = rr(x)
        go = torch.randn_like(y.detach())
        names = [k for k, _ in rr.named_parameters()]
        grads = torch.autograd.grad(y, [x] + [p for _, p in rr.named_parameters()], go)
        yo = O.rrm(sd, "m", x, heads, True, SN_EPS)
        ograds = torch.autograd.grad(yo, [x] + [sd["m." + k] for k in names], go)
        close(yo, y, what=f"rrm_{tag}.y")
        for n_, a, b in zip(["x"] + names, ograds, grads):
            close(a, b, tol=5e-5, what=f"rrm_{tag}.g[{n_}]")
        # weight grads: full tensors for the small G flavour, norms only for the 512-d D flavour
        wsave = ({"gw." + n_: g for n_, g in zip(names, grads[1:])} if tag == "g" else
                 {"gwnorm." + n_: g.norm() for n_, g in zip(names, grads[1:])})
        npz(f"op_rrm_{tag}.npz", x=x, go=go, y=y, gx=grads[0], **wsave)

    # ---- DiffAugment / CR_DiffAug with replayed draws (global generator, reference call order)
    x = torch.rand(5, 1, 32, 48) * 2 - 1
    x.requires_grad_(True)
    torch.manual_seed(1234)
    y = R_da.DiffAugment(x, policy="color,translation,cutout")
    go = torch.randn_like(y.detach())
    gx, = torch.autograd.grad(y, [x], go)
    torch.manual_seed(1234)
    dr = O.diffaug_draws(5, 32, 48)
    yo = O.diff_augment(x, dr)
    ox, = torch.autograd.grad(yo, [x], go)
    close(yo, y, what="diffaug.y")
    close(ox, gx, what="diffaug.gx")
    npz("op_diffaug.npz", x=x, go=go, y=y, gx=gx, **{"d_" + k: v for k, v in dr.items()})
    for seed in (7, 8):
        x = torch.rand(6, 1, 32, 48) * 2 - 1
        torch.manual_seed(seed)
        y = R_cr.CR_DiffAug(x)
        torch.manual_seed(seed)
        dr = O.cr_draws(6, 32, 48)
        yo = O.cr_diff_augment(x, dr)
        close(yo, y, tol=0, what="cr_diffaug")
        npz(f"op_crdiffaug_{seed}.npz", x=x, y=y, **{"d_" + k: v for k, v in dr.items()})

    # ---- losses
    e = F.normalize(torch.randn(40, 1024), dim=1).requires_grad_(True)
    p = F.normalize(torch.randn(40, 1024), dim=1).requires_grad_(True)
    e2 = F.normalize(torch.randn(40, 1024), dim=1)
    dfk, drl = torch.randn(40, requires_grad=True), torch.randn(40, requires_grad=True)
    crit = R_loss.Conditional_Contrastive_loss("cpu", 40, False)
    mask = R_utils.make_mask(torch.arange(40), 40, "cpu")
    ref = {
        "contra": crit(e, p, mask, torch.arange(40), 1.0, 0),
        "unif": R_loss.unif_loss(e),
        "iea": R_loss.IEA_loss(e, e2),
        "hinge_real": R_loss.loss_hinge_dis(dfk, drl)[0],
        "hinge_fake": R_loss.loss_hinge_dis(dfk, drl)[1],
        "hinge_gen": R_loss.loss_hinge_gen(dfk),
        "l2": R_loss.l2_loss(e, e2),
    }
    orc = {
        "contra": O.contrastive_loss(e, p), "unif": O.unif_loss(e), "iea": O.iea_loss(e, e2),
        "hinge_real": O.hinge_dis(dfk, drl)[0], "hinge_fake": O.hinge_dis(dfk, drl)[1],
        "hinge_gen": O.hinge_gen(dfk), "l2": O.l2_loss(e, e2),
    }
    sav = dict(e=e, p=p, e2=e2, dfk=dfk, drl=drl)
    for k in ref:
        close(orc[k], ref[k], what=f"loss.{k}")
        sav[k] = ref[k]
    tot_r = ref["contra"] + 0.1 * ref["unif"] + ref["iea"]
    tot_o = orc["contra"] + 0.1 * orc["unif"] + orc["iea"]
    ge, gp = torch.autograd.grad(tot_r, [e, p])
    oe, op_ = torch.autograd.grad(tot_o, [e, p])
    close(oe, ge, what="loss.ge")
    close(op_, gp, what="loss.gp")
    sav.update(g_e=ge, g_p=gp)
    npz("op_losses.npz", **sav)

    # ---- ortho, Adam, EMA
    lin = torch.nn.Linear(48, 96)
    lin.weight.grad = torch.zeros_like(lin.weight)
    lin.bias.grad = torch.zeros_like(lin.bias)
    R_utils.ortho(lin, 1e-4)
    close(O.ortho_grad(lin.weight.detach(), 1e-4), lin.weight.grad, what="ortho")
    npz("op_ortho.npz", w=lin.weight, g=lin.weight.grad)
    w0 = torch.randn(300)
    wr = w0.clone().requires_grad_(True)
    opt = torch.optim.Adam([wr], lr=5e-5, betas=(0.0, 0.999), eps=1e-6)
    wo, m, v = w0.clone(), torch.zeros(300), torch.zeros(300)
    gs = []
    for step in range(1, 4):
        g = torch.randn(300)
        gs.append(g)
        wr.grad = g.clone()
        opt.step()
        O.adam_step(wo, g, m, v, step, 5e-5, 0.0, 0.999, 1e-6)
        close(wo, wr, tol=1e-6, what=f"adam step {step}")
    npz("op_adam.npz", w0=w0, grads=torch.stack(gs), w3=wr)
    src, tgt = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)
    ema = R_utils.apply_ema(src, tgt, 0.9, start_itr=5)
    with torch.no_grad():
        src.weight.add_(1.0)
    t_o = {k: v.clone() for k, v in tgt.state_dict().items()}
    for itr in (3, 7):
        ema.update(itr)
        O.ema_update(t_o, src.state_dict(), itr, 0.9, 5)
        close(t_o["weight"], tgt.weight, what=f"ema itr {itr}")


def networks_and_step():
    """G(z,y), D(x,y) and train(x,y) at the 64x64 plumbing geometry (BASELINE configs[0])."""
    cfg = dict(CFG, resolution=64, H_base=1, ema=False)
    n, res = 40, 64
    y = torch.arange(n)
    g_state, d_state = O.synth_nets(cfg, 101, 202)
    x = O.synth_event(n, res, res, 303)

    G, D = R_model.Generator(**cfg), R_model.Discriminator(**cfg)
    G.load_state_dict(g_state)
    D.load_state_dict(d_state)
    G.train()
    D.train()
    # ---- forward parity (training mode: SN u update + BN batch stats)
    torch.manual_seed(5)
    z = torch.randn(n, 128)
    torch.manual_seed(6)
    rdof = torch.randn(40, 4)
    torch.manual_seed(6)
    with torch.no_grad():
        gz = G(z, y)
    gsd = {k: v.clone() for k, v in g_state.items()}
    with torch.no_grad():
        gz_o = O.generator(gsd, cfg, z, y, rdof, True)
    close(gz_o, gz, tol=5e-5, what="G(z,y)")
    for k in gsd:
        close(gsd[k], G.state_dict()[k], tol=5e-5, what=f"G state after fwd [{k}]")
    with torch.no_grad():
        pr, em, do = D(gz, y)
    dsd = {k: v.clone() for k, v in d_state.items()}
    with torch.no_grad():
        pr_o, em_o, do_o = O.discriminator(dsd, cfg, gz, y, True)
    close(pr_o, pr, what="D proxy")
    close(em_o, em, tol=5e-5, what="D embed")
    close(do_o, do, tol=5e-5, what="D out")
    # eval-mode generator (running stats, frozen u)
    G.eval()
    torch.manual_seed(6)
    with torch.no_grad():
        gz_e = G(z, y)
    gsd_e = {k: v.clone() for k, v in G.state_dict().items()}
    with torch.no_grad():
        gz_eo = O.generator(gsd_e, cfg, z, y, rdof, False)
    close(gz_eo, gz_e, tol=5e-5, what="G eval")
    G.train()
    npz("net_64.npz", z=z, rdof=rdof, gz=gz, d_proxy=pr, d_embed=em, d_out=do, gz_eval=gz_e,
        g_u_linear=G.linear.u0, g_bn_mean=G.blocks[0][0].bn1.stored_mean,
        g_bn_var=G.blocks[0][0].bn1.stored_var, export=R_model.F.threshold(gz, -0.26, -1)[:2])

    # ---- train step: finite clip_norm (G steps) and the default clip_norm=None quirk (9-Q1)
    for tag, clip in (("clip1e9", 1e9), ("default", None)):
        c = dict(cfg, clip_norm=clip)
        G, D = R_model.Generator(**c), R_model.Discriminator(**c)
        G.load_state_dict(g_state)
        D.load_state_dict(d_state)
        GD = R_model.G_D(G, D)
        z_, y_ = R_utils.prepare_z_y(n, G.dim_z, n, device="cpu")
        train = R_train.GAN_training_function(G, D, GD, z_, y_, None, {"itr": 1}, c, "cpu")
        G.train()
        D.train()
        torch.manual_seed(909)
        out_r = train(x, y)
        # replay the draws in the reference's order (train_fns.py:53,151; model.py:466; diff_aug.py)
        torch.manual_seed(909)
        noise = {}
        for ph in ("d", "g"):
            noise["z_" + ph] = torch.empty(n, 128).normal_(0, 1.0)
            noise["rdof_" + ph] = torch.randn(40, 4)
            noise["aug_" + ph] = O.diffaug_draws(n, res, res)
        gsd, gp = O.as_trainable(g_state)
        dsd, dp = O.as_trainable(d_state)
        ts = O.TrainState(gsd, dsd, gp, dp, c)
        out_o = O.train_step(ts, x, y, noise, itr=1)
        for k in out_r:
            assert abs(out_r[k] - out_o[k]) <= 2e-4 * max(1.0, abs(out_r[k])), (tag, k, out_r[k], out_o[k])
        g_grads, d_grads = ts.last_grads
        sav = {"loss_" + k: v for k, v in out_r.items()}
        for net, sd_o, name in ((G, gsd, "G"), (D, dsd, "D")):
            ref_sd = net.state_dict()
            sums, asums = [], []
            for k in ref_sd:
                close(sd_o[k], ref_sd[k], tol=1e-4, what=f"{tag} {name} post-step [{k}]")
                sums.append(ref_sd[k].double().sum().item())
                asums.append(ref_sd[k].double().abs().sum().item())
            sav[f"{name}_sum"] = np.array(sums)
            sav[f"{name}_abssum"] = np.array(asums)
        gn = []
        for k, p in G.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            close(g_grads[k], gr, tol=2e-4, what=f"{tag} G grad [{k}]")
            gn.append(gr.norm().item())
        dn = []
        for k, p in D.named_parameters():
            # D.grad holds the D-phase gradient (the G phase runs with requires_grad False)
            close(d_grads[k], p.grad, tol=2e-4, what=f"{tag} D grad [{k}]")
            dn.append(p.grad.norm().item())
        moved = sum(int(not torch.equal(p.detach(), g_state[k])) for k, p in G.named_parameters())
        sav.update(G_gradnorm=np.array(gn), D_gradnorm=np.array(dn), G_params_moved=moved,
                   **{"noise_" + k: v for k, v in noise.items() if not isinstance(v, dict)},
                   **{f"noise_aug_{ph}_{k}": v for ph in "dg" for k, v in noise["aug_" + ph].items()})
        print(f"  step[{tag}]: {out_r}  G params moved: {moved}")
        npz(f"step_64_{tag}.npz", **sav)


if __name__ == "__main__":
    print("contract"); contract()
    print("ops"); ops()
    print("networks + step"); networks_and_step()
    print("golden fixtures regenerated; oracle agrees with the reference on every vector")
