"""The oracle (oracle/ieagan_oracle.py) against the committed golden vectors (CPU, no reference).

The vectors were produced by the reference itself (tests/golden/make_golden.py); these tests are
what pins the oracle on any machine where /root/reference is absent.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ieagan_oracle as O


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


def _close(a, b, tol=2e-5, what=""):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1.0)
    assert err <= tol * scale, f"{what}: max|diff|={err:.3e} scale={scale:.3e}"


def _module_state(shapes, seed, prefix="m"):
    st = O.synth_state(shapes, seed)
    return {f"{prefix}.{k}": v.clone().requires_grad_(not O.is_buffer(k)) for k, v in st.items()}


def _sn_shapes(out_f, in_shape, bias=True, n_u=None):
    s = {"weight": (out_f,) + tuple(in_shape)}
    if bias:
        s["bias"] = (out_f,)
    s["u0"] = (1, n_u or out_f)
    s["sv0"] = (1,)
    return s


def test_contract_matches_spec(golden_dir, ref_cfg):
    import json
    contract = json.load(open(os.path.join(golden_dir, "state_dict_contract.json")))
    for tag, over in (("256x768", {}), ("64x64", {"resolution": 64, "H_base": 1})):
        cfg = dict(ref_cfg, **over)
        for name, spec in (("G", O.g_spec(cfg)), ("D", O.d_spec(cfg))):
            ent = contract[f"{name}_{tag}"]
            assert {k: list(v) for k, v in spec.items()} == ent["keys"]
            assert sorted(k for k in spec if not O.is_buffer(k)) == ent["params"]
    assert contract["G_256x768"]["n_params"] == 11693473
    assert contract["D_256x768"]["n_params"] == 4476610


@pytest.mark.parametrize("name,shapes,pad", [
    ("snconv3", _sn_shapes(48, (32, 3, 3)), 1),
    ("snconv1", _sn_shapes(16, (64, 1, 1)), 0),
    ("snlinear", _sn_shapes(128, (132,)), None),
])
def test_sn_layers(golden_dir, ref_cfg, name, shapes, pad):
    g = _load(golden_dir, f"op_{name}.npz")
    sd = _module_state(shapes, 11)
    x = g["x"].clone().requires_grad_(True)
    eps = ref_cfg["SN_eps"]
    y = O.conv(sd, "m", x, True, eps, pad) if pad is not None else O.linear(sd, "m", x, True, eps)
    gx, gw, gb = torch.autograd.grad(y, [x, sd["m.weight"], sd["m.bias"]], g["go"])
    for a, k in ((y, "y"), (gx, "gx"), (gw, "gw"), (gb, "gb"), (sd["m.u0"], "u_after"), (sd["m.sv0"], "sv_after")):
        _close(a, g[k], what=f"{name}.{k}")


def test_sn_embedding(golden_dir, ref_cfg):
    g = _load(golden_dir, "op_snembed.npz")
    sd = _module_state(_sn_shapes(40, (1024,), bias=False, n_u=40), 12)
    y = O.embedding(sd, "m", torch.arange(40), True, ref_cfg["SN_eps"])
    _close(y, g["y"])
    _close(sd["m.u0"], g["u_after"])
    _close(sd["m.sv0"], g["sv_after"])


def _ccbn_shapes(c, cond):
    s = {"stored_mean": (c,), "stored_var": (c,)}
    for nm in ("gain", "bias"):
        for k, v in _sn_shapes(c, (cond,), bias=False).items():
            s[f"{nm}.{k}"] = v
    return s


def test_ccbn_and_bn(golden_dir, ref_cfg):
    g = _load(golden_dir, "op_ccbn.npz")
    sd = _module_state(_ccbn_shapes(32, 256), 13)
    x, yv = g["x"].clone().requires_grad_(True), g["yv"].clone().requires_grad_(True)
    y = O.ccbn(sd, "m", x, yv, True, ref_cfg["BN_eps"], ref_cfg["SN_eps"])
    gx, gy, gwg, gwb = torch.autograd.grad(y, [x, yv, sd["m.gain.weight"], sd["m.bias.weight"]], g["go"])
    for a, k in ((y, "y"), (gx, "gx"), (gy, "gy"), (gwg, "gwg"), (gwb, "gwb"),
                 (sd["m.stored_mean"], "mean_after"), (sd["m.stored_var"], "var_after")):
        _close(a, g[k], what=f"ccbn.{k}")
    g2 = _load(golden_dir, "op_bn.npz")
    sd = _module_state({"gain": (32,), "bias": (32,), "stored_mean": (32,), "stored_var": (32,)}, 14)
    x = g2["x"].clone().requires_grad_(True)
    y = O.plain_bn(sd, "m", x, True, ref_cfg["BN_eps"])
    gx, gg, gb = torch.autograd.grad(y, [x, sd["m.gain"], sd["m.bias"]], g2["go"])
    for a, k in ((y, "y"), (gx, "gx"), (gg, "gg"), (gb, "gb"), (sd["m.stored_mean"], "mean_after"),
                 (sd["m.stored_var"], "var_after")):
        _close(a, g2[k], what=f"bn.{k}")


def _gblock_shapes(cin, cout, cond=256):
    hid = cin // 4
    s = {}
    for nm, o, i in (("conv1", hid, (cin, 1, 1)), ("conv2", hid, (hid, 3, 3)), ("conv3", hid, (hid, 3, 3)),
                     ("conv4", cout, (hid, 1, 1))):
        for k, v in _sn_shapes(o, i).items():
            s[f"{nm}.{k}"] = v
    for j, c in enumerate((cin, hid, hid, hid), 1):
        for k, v in _ccbn_shapes(c, cond).items():
            s[f"bn{j}.{k}"] = v
    return s


def _dblock_shapes(cin, cout):
    hid = cout // 4
    s = {}
    convs = [("conv1", hid, (cin, 1, 1)), ("conv2", hid, (hid, 3, 3)), ("conv3", hid, (hid, 3, 3)),
             ("conv4", cout, (hid, 1, 1))]
    if cin != cout:
        convs.append(("conv_sc", cout - cin, (cin, 1, 1)))
    for nm, o, i in convs:
        for k, v in _sn_shapes(o, i).items():
            s[f"{nm}.{k}"] = v
    return s


@pytest.mark.parametrize("tag,cin,cout,up", [("up", 64, 32, True), ("same", 64, 64, False)])
def test_gblock(golden_dir, ref_cfg, tag, cin, cout, up):
    g = _load(golden_dir, f"op_gblock_{tag}.npz")
    sd = _module_state(_gblock_shapes(cin, cout), 15)
    x, yv = g["x"].clone().requires_grad_(True), g["yv"].clone().requires_grad_(True)
    y = O.g_block(sd, "m", x, yv, cin, cout, up, True, ref_cfg)
    names = [k[3:] for k in g if k.startswith("gw.")]
    grads = torch.autograd.grad(y, [x, yv] + [sd["m." + n] for n in names], g["go"])
    _close(y, g["y"], what="y")
    _close(grads[0], g["gx"], tol=5e-5, what="gx")
    _close(grads[1], g["gy"], tol=5e-5, what="gy")
    for n, gr in zip(names, grads[2:]):
        _close(gr, g["gw." + n], tol=5e-5, what=n)


@pytest.mark.parametrize("tag,cin,cout,down,pre", [("down", 32, 64, True, True), ("first", 32, 64, True, False),
                                                   ("same", 64, 64, False, True)])
def test_dblock(golden_dir, ref_cfg, tag, cin, cout, down, pre):
    g = _load(golden_dir, f"op_dblock_{tag}.npz")
    sd = _module_state(_dblock_shapes(cin, cout), 16)
    x = g["x"].clone().requires_grad_(True)
    y = O.d_block(sd, "m", x, cin, cout, down, pre, True, ref_cfg["SN_eps"])
    names = [k[3:] for k in g if k.startswith("gw.")]
    grads = torch.autograd.grad(y, [x] + [sd["m." + n] for n in names], g["go"])
    _close(y, g["y"], what="y")
    _close(grads[0], g["gx"], tol=5e-5, what="gx")
    for n, gr in zip(names, grads[1:]):
        _close(gr, g["gw." + n], tol=5e-5, what=n)


def test_attention(golden_dir, ref_cfg):
    g = _load(golden_dir, "op_attention.npz")
    shapes = {"gamma": ()}
    for nm, o, i in (("theta", 8, 64), ("phi", 8, 64), ("g", 32, 64), ("o", 64, 32)):
        for k, v in _sn_shapes(o, (i, 1, 1), bias=False).items():
            shapes[f"{nm}.{k}"] = v
    sd = _module_state(shapes, 17)
    x = g["x"].clone().requires_grad_(True)
    y = O.nonlocal_attention(sd, "m", x, True, ref_cfg["SN_eps"])
    names = [k[3:] for k in g if k.startswith("gw.")]
    grads = torch.autograd.grad(y, [x] + [sd["m." + n] for n in names], g["go"])
    _close(y, g["y"])
    _close(grads[0], g["gx"], tol=5e-5)
    for n, gr in zip(names, grads[1:]):
        _close(gr, g["gw." + n], tol=5e-5, what=n)


def _rrm_shapes(dim, ff, sn):
    spec = {}
    O._rrm_spec(spec, "R", dim, ff, sn)
    return {k[2:]: v for k, v in spec.items()}


@pytest.mark.parametrize("tag,dim,heads,ff,sn", [("g", 128, 2, 128, False), ("d", 512, 4, 512, True)])
def test_rrm(golden_dir, ref_cfg, tag, dim, heads, ff, sn):
    g = _load(golden_dir, f"op_rrm_{tag}.npz")
    sd = _module_state(_rrm_shapes(dim, ff, sn), 18)
    x = g["x"].clone().requires_grad_(True)
    y = O.rrm(sd, "m", x, heads, True, ref_cfg["SN_eps"])
    _close(y, g["y"])
    gx, = torch.autograd.grad(y, [x], g["go"], retain_graph=True)
    _close(gx, g["gx"], tol=5e-5)
    for k in g:
        if k.startswith("gw."):
            gr, = torch.autograd.grad(y, [sd["m." + k[3:]]], g["go"], retain_graph=True)
            _close(gr, g[k], tol=5e-5, what=k)
        elif k.startswith("gwnorm."):
            gr, = torch.autograd.grad(y, [sd["m." + k[7:]]], g["go"], retain_graph=True)
            assert abs(gr.norm().item() - g[k].item()) <= 1e-4 * max(1.0, g[k].item()), k


def test_augmentations(golden_dir):
    g = _load(golden_dir, "op_diffaug.npz")
    x = g["x"].clone().requires_grad_(True)
    dr = {k[2:]: g[k] for k in g if k.startswith("d_")}
    y = O.diff_augment(x, dr)
    gx, = torch.autograd.grad(y, [x], g["go"])
    _close(y, g["y"])
    _close(gx, g["gx"])
    # the draw helper replays the reference's RNG call order
    torch.manual_seed(1234)
    dr2 = O.diffaug_draws(5, 32, 48)
    for k in dr:
        assert torch.equal(dr2[k].reshape(-1).to(dr[k].dtype), dr[k].reshape(-1)), k
    for seed in (7, 8):
        g = _load(golden_dir, f"op_crdiffaug_{seed}.npz")
        dr = {k[2:]: g[k] for k in g if k.startswith("d_")}
        assert torch.equal(O.cr_diff_augment(g["x"], dr), g["y"])


def test_losses(golden_dir):
    g = _load(golden_dir, "op_losses.npz")
    e, p = g["e"].clone().requires_grad_(True), g["p"].clone().requires_grad_(True)
    vals = {"contra": O.contrastive_loss(e, p), "unif": O.unif_loss(e), "iea": O.iea_loss(e, g["e2"]),
            "hinge_real": O.hinge_dis(g["dfk"], g["drl"])[0], "hinge_fake": O.hinge_dis(g["dfk"], g["drl"])[1],
            "hinge_gen": O.hinge_gen(g["dfk"]), "l2": O.l2_loss(e, g["e2"])}
    for k, v in vals.items():
        _close(v, g[k], what=k)
    ge, gp = torch.autograd.grad(vals["contra"] + 0.1 * vals["unif"] + vals["iea"], [e, p])
    _close(ge, g["g_e"])
    _close(gp, g["g_p"])


def test_ortho_and_adam(golden_dir):
    g = _load(golden_dir, "op_ortho.npz")
    _close(O.ortho_grad(g["w"], 1e-4), g["g"])
    a = _load(golden_dir, "op_adam.npz")
    w, m, v = a["w0"].clone(), torch.zeros(300), torch.zeros(300)
    for step in range(1, 4):
        O.adam_step(w, a["grads"][step - 1], m, v, step, 5e-5, 0.0, 0.999, 1e-6)
    _close(w, a["w3"], tol=1e-6)


def test_networks_64(golden_dir, ref_cfg):
    """G(z,y), D(x,y) at the 40x64x64 plumbing geometry (BASELINE configs[0])."""
    g = _load(golden_dir, "net_64.npz")
    cfg = dict(ref_cfg, resolution=64, H_base=1)
    gsd, dsd = O.synth_nets(cfg, 101, 202)
    y = torch.arange(40)
    with torch.no_grad():
        gz = O.generator(gsd, cfg, g["z"], y, g["rdof"], True)
        _close(gz, g["gz"], tol=5e-5, what="G(z,y)")
        _close(gsd["linear.u0"], g["g_u_linear"])
        _close(gsd["blocks.0.0.bn1.stored_mean"], g["g_bn_mean"], tol=5e-5)
        _close(gsd["blocks.0.0.bn1.stored_var"], g["g_bn_var"], tol=5e-5)
        pr, em, do = O.discriminator(dsd, cfg, g["gz"], y, True)
        _close(pr, g["d_proxy"])
        _close(em, g["d_embed"], tol=5e-5)
        _close(do, g["d_out"], tol=5e-5)
        gz_e = O.generator(gsd, cfg, g["z"], y, g["rdof"], False)
        _close(gz_e, g["gz_eval"], tol=5e-5, what="G eval")


@pytest.mark.parametrize("tag,clip,moved", [("clip1e9", 1e9, 151), ("default", None, 0)])
def test_train_step_64(golden_dir, ref_cfg, tag, clip, moved):
    """One full G+D step at 40x64x64: the 5 losses, per-parameter grad norms, post-step checksums;
    ``default`` pins the reference quirk that G never steps when clip_norm is None."""
    g = _load(golden_dir, f"step_64_{tag}.npz")
    cfg = dict(ref_cfg, resolution=64, H_base=1, ema=False, clip_norm=clip)
    g0, d0 = O.synth_nets(cfg, 101, 202)
    gsd, gp = O.as_trainable(g0)
    dsd, dp = O.as_trainable(d0)
    noise = {}
    for ph in "dg":
        noise["z_" + ph] = g["noise_z_" + ph]
        noise["rdof_" + ph] = g["noise_rdof_" + ph]
        noise["aug_" + ph] = {k.split("_", 3)[3]: g[k] for k in g if k.startswith(f"noise_aug_{ph}_")}
    ts = O.TrainState(gsd, dsd, gp, dp, cfg)
    out = O.train_step(ts, O.synth_event(40, 64, 64, 303), torch.arange(40), noise, itr=1)
    for k, v in out.items():
        ref = g["loss_" + k].item()
        assert abs(v - ref) <= 2e-4 * max(1.0, abs(ref)), (k, v, ref)
    g_grads, d_grads = ts.last_grads
    gn = torch.tensor([g_grads[k].norm().item() for k in gp])
    dn = torch.tensor([d_grads[k].norm().item() for k in dp])
    _close(gn, g["G_gradnorm"], tol=5e-4, what="G grad norms")
    _close(dn, g["D_gradnorm"], tol=5e-4, what="D grad norms")
    for name, sd, spec in (("G", gsd, O.g_spec(cfg)), ("D", dsd, O.d_spec(cfg))):
        sums = torch.tensor([sd[k].double().sum().item() for k in spec], dtype=torch.float64)
        asums = torch.tensor([sd[k].double().abs().sum().item() for k in spec], dtype=torch.float64)
        assert torch.allclose(sums, g[f"{name}_sum"], rtol=1e-4, atol=1e-3), name
        assert torch.allclose(asums, g[f"{name}_abssum"], rtol=1e-4, atol=1e-3), name
    n_moved = sum(int(not torch.equal(gsd[k].detach(), g0[k])) for k in gp)
    assert n_moved == moved == int(g["G_params_moved"])


def _step_noise(g):
    noise = {}
    for ph in "dg":
        noise["z_" + ph] = g["noise_z_" + ph]
        noise["rdof_" + ph] = g["noise_rdof_" + ph]
        noise["aug_" + ph] = {k.split("_", 3)[3]: g[k] for k in g if k.startswith(f"noise_aug_{ph}_")}
    if "noise_y_g" in g:
        noise["y_g"] = g["noise_y_g"].long()
    return noise


@pytest.mark.parametrize("tag,over", [("joint", {"split_D": False}), ("proj", {"conditional_strategy": "Proj"})])
def test_train_step_64_joint_pass_and_projection_head(golden_dir, ref_cfg, tag, over):
    """``split_D=False`` (D evaluated once on cat[G_z, x]: model.py:1024-1068; fixture = the reference's own train()) and
    ``conditional_strategy='Proj'`` (model.py:939-944; the reference's train() under try/except + a composition of its
    G_D / hinge calls, tests/golden/make_golden_r3.py): losses, per-parameter gradient norms, post-step checksums."""
    g = _load(golden_dir, f"step_64_{tag}.npz")
    cfg = dict(ref_cfg, resolution=64, H_base=1, ema=False, clip_norm=1e9, **over)
    g0, d0 = O.synth_nets(cfg, 101, 202)
    gsd, gp = O.as_trainable(g0)
    dsd, dp = O.as_trainable(d0)
    ts = O.TrainState(gsd, dsd, gp, dp, cfg)
    out = O.train_step(ts, O.synth_event(40, 64, 64, 303), torch.arange(40), _step_noise(g), itr=1)
    for k, v in out.items():
        ref = g["loss_" + k].item()
        assert abs(v - ref) <= 2e-4 * max(1.0, abs(ref)), (k, v, ref)
    g_grads, d_grads = ts.last_grads
    _close(torch.tensor([g_grads[k].norm().item() for k in gp]), g["G_gradnorm"], tol=5e-4, what="G grad norms")
    _close(torch.tensor([d_grads[k].norm().item() for k in dp]), g["D_gradnorm"], tol=5e-4, what="D grad norms")
    for name, sd, spec in (("G", gsd, O.g_spec(cfg)), ("D", dsd, O.d_spec(cfg))):
        sums = torch.tensor([sd[k].double().sum().item() for k in spec], dtype=torch.float64)
        asums = torch.tensor([sd[k].double().abs().sum().item() for k in spec], dtype=torch.float64)
        assert torch.allclose(sums, g[f"{name}_sum"], rtol=1e-4, atol=1e-3), name
        assert torch.allclose(asums, g[f"{name}_abssum"], rtol=1e-4, atol=1e-3), name
    if tag == "proj":       # RR_D / norm exist (RRM_embed) but the projection head never evaluates them
        assert all(float(d_grads[k].abs().max()) == 0.0 for k in dp if k.startswith("RR_D.") or k.startswith("norm."))
        assert torch.equal(dsd["RR_D.layers.0.linear_net.0.u0"], d0["RR_D.layers.0.linear_net.0.u0"])


def test_ingest_and_frechet_against_reference_vectors(golden_dir):
    """Event ingestion chain and the Frechet distance (fixtures written by tests/golden/make_golden_io.py from the
    reference's fn_lognorm255 / UniformNoise / frechet_distance)."""
    g = _load(golden_dir, "op_ingest.npz")
    out = O.ingest_event(g["ev"], g["u"])
    _close(out, g["out"], 1e-6)
    assert out.shape == (5, 1, 16, 16) and float(out.min()) >= -1.0 and float(out.max()) <= 1.0 + 8e-3
    f = np.load(os.path.join(golden_dir, "op_frechet.npz"))
    for name, (p, q) in {"ab": ("a", "b"), "aa": ("a", "a"), "ac": ("a", "c")}.items():
        a, b = f[p], f[q]
        fd = O.frechet_distance(a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
        assert abs(fd - float(f["fd_" + name])) <= 1e-8 * max(1.0, abs(fd)), name


def test_export_against_reference_vectors(golden_dir):
    """model.generate's post-processing and utils.norm.denorm (fixture written by make_golden_io.py from the reference's
    own functions): bit-exact, including pixels exactly on / either side of the -0.26 threshold."""
    g = _load(golden_dir, "op_export.npz")
    assert torch.equal(O.generate_export(g["img"]), g["adu"])
    assert torch.equal(O.denorm(g["img"]), g["denorm"])
    assert g["adu"].shape == (40, 10, 24) and float(g["adu"].min()) == 0.0 and float(g["adu"].max()) == 255.0


def _cfg3_inputs(g, n=40, res=64):
    E = int(g["E"])
    noises = []
    for e in range(E):
        nz = {}
        for k in g:
            if not k.startswith(f"noise{e}_"):
                continue
            name = k[len(f"noise{e}_"):]
            if "." in name:
                a, b = name.split(".", 1)
                nz.setdefault(a, {})[b] = g[k]
            else:
                nz[name] = g[k]
        noises.append(nz)
    return [O.synth_event(n, res, res, 303 + e) for e in range(E)], noises


def test_train_step_events_con_reg_64(golden_dir, ref_cfg):
    """BASELINE configs[3] semantics at 40x64x64: E = 2 events per step, DiffAugment + CR_DiffAug consistency
    regularisation (third discriminator pass) + uniformity loss.  The fixture is the composition of the reference's own
    modules (tests/golden/make_golden_cfg3.py); E = 1 of the same code path is pinned by test_train_step_64."""
    g = _load(golden_dir, "step_64_cfg3.npz")
    cfg = dict(ref_cfg, resolution=64, H_base=1, ema=False, clip_norm=1e9, Con_reg=True)
    xs, noises = _cfg3_inputs(g)
    g0, d0 = O.synth_nets(cfg, 101, 202)
    gsd, gp = O.as_trainable(g0)
    dsd, dp = O.as_trainable(d0)
    ts = O.TrainState(gsd, dsd, gp, dp, cfg)
    out = O.train_step_events(ts, xs, torch.arange(40), noises, itr=1)
    for k, v in out.items():
        ref = g["loss_" + k].item()
        assert abs(v - ref) <= 2e-4 * max(1.0, abs(ref)), (k, v, ref)
    g_grads, d_grads = ts.last_grads
    _close(torch.tensor([g_grads[k].norm().item() for k in gp]), g["G_gradnorm"], tol=5e-4, what="G grad norms")
    _close(torch.tensor([d_grads[k].norm().item() for k in dp]), g["D_gradnorm"], tol=5e-4, what="D grad norms")
    for name, sd, spec in (("G", gsd, O.g_spec(cfg)), ("D", dsd, O.d_spec(cfg))):
        sums = torch.tensor([sd[k].double().sum().item() for k in spec], dtype=torch.float64)
        asums = torch.tensor([sd[k].double().abs().sum().item() for k in spec], dtype=torch.float64)
        assert torch.allclose(sums, g[f"{name}_sum"], rtol=1e-4, atol=1e-3), name
        assert torch.allclose(asums, g[f"{name}_abssum"], rtol=1e-4, atol=1e-3), name
