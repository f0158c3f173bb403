"""Run-to-run reproducibility probe of the G-phase gradient (development aid): same nets / event / draws several times."""
import os, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), os.path.join(root, "tests"), os.path.join(root, "oracle"), root]
import torch
import model, ops, train_fns, utils
from parity_util import O, build_product, make_cfg, make_noise, rel_l2

n = int(os.environ.get("NP_N", 8))
res = int(os.environ.get("NP_RES", 256))
hb = 3 if res == 256 else 1
cfg = make_cfg(resolution=res, H_base=hb, clip_norm=1e9, hip_graph=False, ema=False, batch_size=n, D_lr=0.0)
x, y = O.synth_event(n, res, res * hb, 404).cuda(), torch.arange(n).cuda()
noise = make_noise(n, res, res * hb, 919)
runs = []
# (weight gradients on the side stream, two-stage accumulation, fused 1x1 backward, fused 3x3 backward)
settings = [(True, True, True, True), (True, True, True, True), (False, False, True, True), (False, False, False, False),
            (True, True, False, True), (True, True, True, False)]
for side, two, f1, f3 in settings:
    g_state, d_state = O.synth_nets(cfg, 111, 222)
    G, D = build_product(cfg, g_state, d_state, "cuda:0")
    z_, y_ = utils.prepare_z_y(n, G.dim_z, cfg["n_classes"], device="cuda:0")
    train = train_fns.GAN_training_function(G, D, model.G_D(G, D), z_, y_, None, {"itr": 1}, cfg, "cuda:0")
    for net in (G, D):
        ops.set_options(net, wgrad_side_stream=side, two_stage_wgrad=two, fuse_1x1_backward=f1, fuse_3x3_backward=f3)
    out = train(x, y, noise=noise)
    torch.cuda.synchronize()
    names = [k for k, _ in G.named_parameters()]
    grads = {k: p.grad.detach().clone() for k, p in G.named_parameters()}
    runs.append((out, G._arena.grad.clone(), grads, D._arena.grad.clone()))
for a, b in ((0, 1), (0, 2), (0, 3), (0, 4), (0, 5)):
    oa, ga, pa, da = runs[a]
    ob, gb, pb, db = runs[b]
    print(f"settings {settings[a]} vs {settings[b]}")
    print(f"runs {a} vs {b}: G flat rel {rel_l2(gb, ga):.3e}  D flat rel {rel_l2(db, da):.3e}  losses {oa} {ob}")
    rows = []
    for k in pa:
        na = float(pa[k].norm())
        rows.append((float((pa[k] - pb[k]).norm()) / max(na, 1e-20), na, k))
    rows.sort(reverse=True)
    for r, na, k in rows[:12]:
        print(f"    {r:.3e}  |g|={na:.3e}  {k}")
    tot = sum(float((pa[k] - pb[k]).norm()) ** 2 for k in pa) ** 0.5
    worst = sorted(((float((pa[k] - pb[k]).norm()) ** 2, k) for k in pa), reverse=True)[:6]
    print("    largest contributions to the flat difference:", [(round(v ** 0.5 / tot, 3), k) for v, k in worst])
