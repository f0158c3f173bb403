import os, sys, math
root = "/root/repo"
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), os.path.join(root, "tests"), root]
import torch
import _hip, ops
from test_hip_ops import make_rec, nhwc
dev = torch.device("cuda:0")
def run(C, Hs, Ws, aff, rs, stats, N):
    torch.manual_seed(5)
    x0 = nhwc(torch.randn(N, C, Hs, Ws, device=dev))
    W0 = torch.randn(C, C, 3, 3, device=dev) / math.sqrt(9 * C)
    u = torch.randn(1, C, device=dev)
    b0 = 0.1 * torch.randn(C, device=dev)
    sc0 = (1 + 0.3 * torch.randn(N, C, device=dev)) if aff else None
    sh0 = 0.2 * torch.randn(N, C, device=dev) if aff else None
    Hc, Wc = (2 * Hs, 2 * Ws) if rs == 1 else (Hs, Ws)
    go = nhwc(torch.randn(N, C, Hc, Wc, device=dev))
    dsum = 0.05 * torch.randn(1, 1, 2, C, device=dev)
    res = {}
    for fused in (False, True):
        ops.DEFAULTS.fuse_3x3_backward, ops.DEFAULTS.fuse_3x3_min_pixels = fused, 1024
        rec, Wv, uv, svv = make_rec(W0.clone(), u.clone(), torch.ones(1, device=dev))
        Wp = Wv.detach().requires_grad_(True)
        xa = x0.clone().requires_grad_(True)
        b2 = b0.clone().requires_grad_(True)
        sc2 = sc0.clone().requires_grad_(True) if aff else None
        sh2 = sh0.clone().requires_grad_(True) if aff else None
        out, st = ops.conv(xa, Wp, b2, rec, 9, scale=sc2, shift=sh2, relu=True, rs=rs, want_stats=stats)
        loss = (out.float() * go.float()).sum()
        if stats:
            loss = loss + (st * dsum).sum()
        leaves = [t for t in (xa, Wp, b2, sc2, sh2) if t is not None]
        grads = torch.autograd.grad(loss, leaves)
        res[fused] = dict(zip([n for n, t in zip(("x", "W", "bias", "scale", "shift"), (xa, Wp, b2, sc2, sh2)) if t is not None], grads))
    line = f"C{C} {Hs}x{Ws} aff{int(aff)} rs{rs} eff{int(stats)}: "
    for k, ref in res[False].items():
        got = res[True][k].float(); ref = ref.float()
        err = float((got - ref).norm() / ref.norm())
        line += f"{k} {err:.2e} "
        if k == "x":
            d = (got - ref).abs()
            nz = (d > 0).float().mean().item()
            line += f"(frac differing {nz:.3f}, max {d.max().item():.3e} of {ref.abs().max().item():.3e}) "
            # where: border rows/cols vs interior
            dd = d.sum(-1)[0]
            line += f"[row0 {dd[0].mean().item():.2e} mid {dd[dd.shape[0]//2].mean().item():.2e} col0 {dd[:,0].mean().item():.2e} ] "
    print(line, flush=True)
for C, Hs, Ws in ((16, 64, 64), (32, 64, 64)):
    for aff, stats in ((False, False), (True, True)):
        for rs in (0, 1):
            run(C, Hs, Ws, aff, rs, stats, 4)
