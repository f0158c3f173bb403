"""Golden fixture for BASELINE configs[3]: E events per step with DiffAugment + CR_DiffAug consistency regularisation +
uniformity loss  (development container only; same import recipe as make_golden.py).

The reference cannot run this configuration end to end: one step consumes exactly one event and
``Con_reg`` with ``split_D`` raises before the loss is formed (9-Q3).  The step is therefore COMPOSED here from the
reference's own pieces -- ``model.Generator`` / ``model.Discriminator`` / ``DiffAugment`` / ``CR_DiffAug`` /
``loss.*`` / ``utils.ortho`` / the networks' own ``optim.Adam`` -- with the semantics DESIGN.md section 7 defines
(every event an independent pass from the same weights and buffers, third discriminator pass on the augmented real
event, loss terms of train_fns.py:93-102 plus the uniformity term, gradients averaged, running statistics = mean of
the per-event updates), and the oracle's ``train_step_events`` must agree with that composition before the data file
is written.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cfg3.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                     # noqa: E402  (sets up the stubbed imports of the reference)
from make_golden import O, R_model, R_loss, R_da, R_cr, R_utils, CFG, close, npz      # noqa: E402


def main(E=2, n=40, res=64):
    cfg = dict(CFG, resolution=res, H_base=1, ema=False, clip_norm=1e9, Con_reg=True, cr_lambda=10, diff_aug=True,
               Uniformity_loss=True, IEA_loss=True)
    y = torch.arange(n)
    g_state, d_state = O.synth_nets(cfg, 101, 202)
    xs = [O.synth_event(n, res, res, 303 + e) for e in range(E)]
    G, D = R_model.Generator(**cfg), R_model.Discriminator(**cfg)
    G.load_state_dict(g_state)
    D.load_state_dict(d_state)
    G.train()
    D.train()
    contra = R_loss.Conditional_Contrastive_loss("cpu", n, cfg["pos_collected_numerator"])
    is_buf = lambda k: O.is_buffer(k)

    def buffers(net):
        return {k: v.clone() for k, v in net.state_dict().items() if is_buf(k)}

    def restore(net, snap):
        sd = net.state_dict()
        with torch.no_grad():
            for k, v in snap.items():
                sd[k].copy_(v)

    def mean(snaps):
        return {k: torch.stack([s[k] for s in snaps]).mean(0) for k in snaps[0]}

    noises = []
    # ---------------------------------------------------------------- D phase
    R_utils.toggle_grad(D, True)
    R_utils.toggle_grad(G, False)
    G.optim.zero_grad()
    D.optim.zero_grad()
    g0, d0 = buffers(G), buffers(D)
    ends, emb_reals, dvals = [], [], []
    for e in range(E):
        restore(G, g0)
        restore(D, d0)
        # replay the draw order of the reference's train(): CR_DiffAug first (train_fns.py:27-28), then z (:53), rdof
        # (model.py:466), DiffAugment (diff_aug.py)
        torch.manual_seed(1000 + e)
        nz = {"cr": O.cr_draws(n, res, res)}
        nz["z_d"] = torch.empty(n, 128).normal_(0, 1.0)
        nz["rdof_d"] = torch.randn(40, 4)
        nz["aug_d"] = O.diffaug_draws(n, res, res)
        torch.manual_seed(1000 + e)
        x_aug = R_cr.CR_DiffAug(xs[e])
        assert torch.equal(x_aug, O.cr_diff_augment(xs[e], nz["cr"])), "CR_DiffAug draw replay"
        z = torch.empty(n, 128).normal_(0, 1.0)
        with torch.no_grad():
            G_z = G(z, y)
            G_z = R_da.DiffAugment(G_z, policy="color,translation,cutout")
        pf, ef, D_fake = D(G_z, y)
        pr, er, D_real = D(xs[e], y)
        _, ea, D_real_aug = D(x_aug, y)                                         # 9-Q3: third pass
        l_real, l_fake = R_loss.loss_hinge_dis(D_fake, D_real)
        D_loss = l_real + l_fake
        mask = R_utils.make_mask(y, cfg["n_classes"], "cpu")
        D_loss = D_loss + cfg["contra_lambda"] * contra(er, pr, mask, y, 1.0, 0)
        unif_d = R_loss.unif_loss(er)
        D_loss = D_loss + cfg["unif_lambda"] * unif_d
        cons = R_loss.l2_loss(D_real, D_real_aug) + R_loss.l2_loss(er, ea)      # train_fns.py:93-97
        D_loss = D_loss + cfg["cr_lambda"] * cons
        (D_loss / E).backward()
        ends.append((buffers(G), buffers(D)))
        emb_reals.append(er.detach())
        dvals.append((float(l_real), float(l_fake), float(unif_d)))
        noises.append(nz)
    restore(G, mean([t[0] for t in ends]))
    restore(D, mean([t[1] for t in ends]))
    d_grads_ref = {k: p.grad.clone() for k, p in D.named_parameters()}
    torch.nn.utils.clip_grad_norm_(D.parameters(), cfg["clip_norm"])
    D.optim.step()
    # ---------------------------------------------------------------- G phase
    R_utils.toggle_grad(D, False)
    R_utils.toggle_grad(G, True)
    G.optim.zero_grad()
    g0, d0 = buffers(G), buffers(D)
    ends, gvals = [], []
    for e in range(E):
        restore(G, g0)
        restore(D, d0)
        torch.manual_seed(2000 + e)
        noises[e]["z_g"] = torch.empty(n, 128).normal_(0, 1.0)
        noises[e]["rdof_g"] = torch.randn(40, 4)
        noises[e]["aug_g"] = O.diffaug_draws(n, res, res)
        torch.manual_seed(2000 + e)
        z = torch.empty(n, 128).normal_(0, 1.0)
        G_z = R_da.DiffAugment(G(z, y), policy="color,translation,cutout")
        pf, ef, D_fake = D(G_z, y)
        G_loss = R_loss.loss_hinge_gen(D_fake)
        mask = R_utils.make_mask(y, cfg["n_classes"], "cpu")
        G_loss = G_loss + cfg["contra_lambda"] * contra(ef, pf, mask, y, 1.0, 0)
        iea = R_loss.IEA_loss(ef, emb_reals[e])
        G_loss = G_loss + cfg["IEA_lambda"] * iea + cfg["unif_lambda"] * R_loss.unif_loss(ef)
        (G_loss / E).backward()
        ends.append((buffers(G), buffers(D)))
        gvals.append((float(G_loss), float(iea)))
    restore(G, mean([t[0] for t in ends]))
    restore(D, mean([t[1] for t in ends]))
    R_utils.ortho(G, cfg["G_ortho"], blacklist=[p for p in G.shared.parameters()])
    torch.nn.utils.clip_grad_norm_(G.parameters(), cfg["clip_norm"])
    G.optim.step()
    ref_out = {"G_loss": np.mean([v[0] for v in gvals]), "D_loss_real": np.mean([v[0] for v in dvals]),
               "D_loss_fake": np.mean([v[1] for v in dvals]), "unif_loss_d": np.mean([v[2] for v in dvals]),
               "iea_loss": np.mean([v[1] for v in gvals])}
    # ---------------------------------------------------------------- oracle on the same draws
    gsd, gp = O.as_trainable(g_state)
    dsd, dp = O.as_trainable(d_state)
    ts = O.TrainState(gsd, dsd, gp, dp, cfg)
    out = O.train_step_events(ts, xs, y, noises, itr=1)
    for k, v in ref_out.items():
        assert abs(out[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, out[k], v)
    g_grads, d_grads = ts.last_grads
    sav = {"loss_" + k: float(v) for k, v in ref_out.items()}
    for k, p in D.named_parameters():
        close(d_grads[k], d_grads_ref[k], tol=2e-4, what=f"D grad [{k}]")
    for k, p in G.named_parameters():
        close(g_grads[k], p.grad, tol=2e-4, what=f"G grad [{k}]")
    for net, sd_o, name in ((G, gsd, "G"), (D, dsd, "D")):
        ref_sd = net.state_dict()
        sums, asums = [], []
        for k in ref_sd:
            close(sd_o[k], ref_sd[k], tol=1e-4, what=f"{name} post-step [{k}]")
            sums.append(ref_sd[k].double().sum().item())
            asums.append(ref_sd[k].double().abs().sum().item())
        sav[f"{name}_sum"], sav[f"{name}_abssum"] = np.array(sums), np.array(asums)
    sav["G_gradnorm"] = np.array([p.grad.norm().item() for _, p in G.named_parameters()])
    sav["D_gradnorm"] = np.array([d_grads_ref[k].norm().item() for k, _ in D.named_parameters()])
    for e, nz in enumerate(noises):
        for k, v in nz.items():
            if isinstance(v, dict):
                sav.update({f"noise{e}_{k}.{kk}": vv for kk, vv in v.items()})
            else:
                sav[f"noise{e}_{k}"] = v
    sav["E"] = E
    print("  cfg3 step:", ref_out)
    npz("step_64_cfg3.npz", **sav)


if __name__ == "__main__":
    main()
    print("configs[3] fixture regenerated; the oracle agrees with the composition of reference pieces")
