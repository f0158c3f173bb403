"""Round-3 golden fixtures (development container only; same import recipe as make_golden.py):

* ``step_64_joint.npz``  -- the reference's own ``train()`` with ``split_D=False`` (model.py:1024-1068: D evaluated ONCE on
  cat[G_z, x], so RR_D relates 80 tokens), default loss composition, ``clip_norm=1e9``.
* ``step_64_proj.npz``   -- ``conditional_strategy='Proj'`` (projection head, model.py:939-944).  The reference's ``train()``
  raises UnboundLocalError for this strategy AFTER both optimizers have stepped: the step is run under
  try/except for the post-step state and gradients, and the loss values come from a composition of the reference's own
  ``G_D`` + ``loss.loss_hinge_*`` calls on a second copy of the networks (same draws).
* ``ckpt_ref/``          -- a checkpoint WRITTEN BY the reference's ``utils.save_weights`` (tiny G_ch = D_ch = 2 networks) plus
  ``ckpt_ref.npz`` with per-key checksums: ``utils.load_weights`` of this package must read it.

The oracle must agree with the reference before anything is written.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3.py
"""
import io
import contextlib
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                     # noqa: E402,F401  (sets up the stubbed imports of the reference)
from make_golden import O, R_model, R_loss, R_utils, R_train, CFG, close, npz      # noqa: E402


def _nets(cfg, g_state, d_state):
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = R_model.Generator(**cfg), R_model.Discriminator(**cfg)
    G.load_state_dict(g_state)
    D.load_state_dict(d_state)
    G.train()
    D.train()
    return G, D


def _draws(n, res, seed, y_g=None):
    """The draws of one reference train() call in ITS order (train_fns.py:53,151; model.py:466; diff_aug.py)."""
    torch.manual_seed(seed)
    noise = {}
    for ph in ("d", "g"):
        noise["z_" + ph] = torch.empty(n, 128).normal_(0, 1.0)
        noise["rdof_" + ph] = torch.randn(40, 4)
        noise["aug_" + ph] = O.diffaug_draws(n, res, res)
    if y_g is not None:
        noise["y_g"] = y_g
    return noise


def _save_step(tag, out_r, G, D, gsd, dsd, ts, noise, extra=None):
    g_grads, d_grads = ts.last_grads
    sav = {"loss_" + k: float(v) for k, v in out_r.items()}
    for net, sd_o, name in ((G, gsd, "G"), (D, dsd, "D")):
        ref_sd = net.state_dict()
        sums, asums = [], []
        for k in ref_sd:
            close(sd_o[k], ref_sd[k], tol=1e-4, what=f"{tag} {name} post-step [{k}]")
            sums.append(ref_sd[k].double().sum().item())
            asums.append(ref_sd[k].double().abs().sum().item())
        sav[f"{name}_sum"], sav[f"{name}_abssum"] = np.array(sums), np.array(asums)
    gn, dn = [], []
    for k, p in G.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        close(g_grads[k], gr, tol=2e-4, what=f"{tag} G grad [{k}]")
        gn.append(gr.norm().item())
    for k, p in D.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        close(d_grads[k], gr, tol=2e-4, what=f"{tag} D grad [{k}]")
        dn.append(gr.norm().item())
    sav.update(G_gradnorm=np.array(gn), D_gradnorm=np.array(dn),
               **{"noise_" + k: v for k, v in noise.items() if not isinstance(v, dict)},
               **{f"noise_aug_{ph}_{k}": v for ph in "dg" for k, v in noise["aug_" + ph].items()})
    if extra:
        sav.update(extra)
    print(f"  step[{tag}]: {out_r}")
    npz(f"step_64_{tag}.npz", **sav)


def joint(n=40, res=64):
    cfg = dict(CFG, resolution=res, H_base=1, ema=False, clip_norm=1e9, split_D=False)
    y = torch.arange(n)
    g_state, d_state = O.synth_nets(cfg, 101, 202)
    x = O.synth_event(n, res, res, 303)
    G, D = _nets(cfg, g_state, d_state)
    z_, y_ = R_utils.prepare_z_y(n, G.dim_z, n, device="cpu")
    train = R_train.GAN_training_function(G, D, R_model.G_D(G, D), z_, y_, None, {"itr": 1}, cfg, "cpu")
    torch.manual_seed(909)
    out_r = train(x, y)
    noise = _draws(n, res, 909)
    gsd, gp = O.as_trainable(g_state)
    dsd, dp = O.as_trainable(d_state)
    ts = O.TrainState(gsd, dsd, gp, dp, cfg)
    out_o = O.train_step(ts, x, y, noise, itr=1)
    for k in out_r:
        assert abs(out_r[k] - out_o[k]) <= 2e-4 * max(1.0, abs(out_r[k])), ("joint", k, out_r[k], out_o[k])
    _save_step("joint", out_r, G, D, gsd, dsd, ts, noise)


def proj(n=40, res=64):
    cfg = dict(CFG, resolution=res, H_base=1, ema=False, clip_norm=1e9, conditional_strategy="Proj")
    y = torch.arange(n)
    g_state, d_state = O.synth_nets(cfg, 101, 202)
    x = O.synth_event(n, res, res, 303)
    # ---- the reference's train(): raises at the very end (train_fns.py:198-202), after both optimizers have stepped
    G, D = _nets(cfg, g_state, d_state)
    torch.manual_seed(4321)
    z_, y_ = R_utils.prepare_z_y(n, G.dim_z, n, device="cpu")      # y_: the permuted label vector the Proj G phase feeds (train_fns.py:153)
    y_g = y_.clone().long()
    train = R_train.GAN_training_function(G, D, R_model.G_D(G, D), z_, y_, None, {"itr": 1}, cfg, "cpu")
    torch.manual_seed(909)
    try:
        train(x, y)
        raise AssertionError("the reference was expected to raise UnboundLocalError for Proj (SURVEY 9-Q2)")
    except UnboundLocalError:
        pass
    noise = _draws(n, res, 909, y_g=y_g)
    # ---- loss values: composition of the reference's GD + hinge calls on a second copy, same draws
    G2, D2 = _nets(cfg, g_state, d_state)
    GD2 = R_model.G_D(G2, D2)
    torch.manual_seed(909)
    z = torch.empty(n, 128).normal_(0, 1.0)
    R_utils.toggle_grad(D2, True)
    R_utils.toggle_grad(G2, False)
    D_fake, D_real = GD2(z, y, x, y, x_aug=None, contra=False, train_G=False, split_D=True, diff_aug=True)
    l_real, l_fake = R_loss.loss_hinge_dis(D_fake, D_real)
    D2.optim.zero_grad()
    (l_real + l_fake).backward()
    torch.nn.utils.clip_grad_norm_(D2.parameters(), cfg["clip_norm"])
    D2.optim.step()
    R_utils.toggle_grad(D2, False)
    R_utils.toggle_grad(G2, True)
    z = torch.empty(n, 128).normal_(0, 1.0)
    D_fake = GD2(z, y_g, x_aug=None, contra=False, train_G=True, split_D=True, diff_aug=True)
    G_loss = R_loss.loss_hinge_gen(D_fake)
    out_r = {"G_loss": float(G_loss), "D_loss_real": float(l_real), "D_loss_fake": float(l_fake), "unif_loss_d": 0.0, "iea_loss": 0.0}
    gsd, gp = O.as_trainable(g_state)
    dsd, dp = O.as_trainable(d_state)
    ts = O.TrainState(gsd, dsd, gp, dp, cfg)
    out_o = O.train_step(ts, x, y, noise, itr=1)
    for k in out_r:
        assert abs(out_r[k] - out_o[k]) <= 2e-4 * max(1.0, abs(out_r[k])), ("proj", k, out_r[k], out_o[k])
    # the RR_D / norm parameters exist (RRM_embed) but the projection head never evaluates them: no gradient, u0 untouched
    assert all(p.grad is None for k, p in D.named_parameters() if k.startswith("RR_D.") or k.startswith("norm."))
    assert torch.equal(D.state_dict()["RR_D.layers.0.linear_net.0.u0"], d_state["RR_D.layers.0.linear_net.0.u0"])
    _save_step("proj", out_r, G, D, gsd, dsd, ts, noise, extra={"D_out_fake_g": D_fake.detach()})


def checkpoint():
    """A checkpoint written by the reference's own save_weights (utils/__init__.py:689-726)."""
    cfg = dict(CFG, resolution=64, H_base=1, G_ch=2, D_ch=2, dim_z=8, hypersphere_dim=32, ema=False)
    torch.manual_seed(77)
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = R_model.Generator(**cfg), R_model.Discriminator(**cfg)
    root = os.path.join(HERE, "ckpt_ref")
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(root)
    state = {"itr": 7, "epoch": 1, "save_num": 0, "save_best_num": 0, "best_IS": 0, "best_FID": 999999}
    os.makedirs(os.path.join(root, "run", "weights"))
    with contextlib.redirect_stdout(io.StringIO()):
        R_utils.save_weights(G, D, state, {"outputroot": root, "run_name": "run"}, None, None)
    files = sorted(os.listdir(os.path.join(root, "run", "weights")))
    sums = {}
    for net, name in ((G, "G"), (D, "D")):
        for k, v in net.state_dict().items():
            sums[f"{name}.{k}"] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    npz("ckpt_ref.npz", files=np.array(files), **sums)
    tot = sum(os.path.getsize(os.path.join(root, "run", "weights", f)) for f in files)
    print(f"  wrote ckpt_ref/run: {files} ({tot / 1e6:.2f} MB)")


if __name__ == "__main__":
    which = sys.argv[1:] or ["joint", "proj", "checkpoint"]
    for w in which:
        print(w)
        {"joint": joint, "proj": proj, "checkpoint": checkpoint}[w]()
    print("round-3 fixtures regenerated; the oracle agrees with the reference")
