"""Stand-alone timing of ieagan_conv3x3_bwd at the production shapes (N = 40) next to the separate launches it replaces
(effgrad + dgrad [+ prologue_bwd] + wgrad + reduce), each timed with HIP events over 20 warm launches (development aid)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H
import ops

BF = torch.bfloat16
dev = torch.device("cuda:0")
N = int(os.environ.get("B3_N", 40))
#        name           C   H    W   rs aff eff
CASES = [("D s0.0 c2/3", 16, 256, 768, 0, 0, 0),
         ("D s0.1 c2/3", 16, 128, 384, 0, 0, 0),
         ("D s1.0 c2/3", 32, 128, 384, 0, 0, 0),
         ("D s1.1 c2/3", 32, 64, 192, 0, 0, 0),
         ("G b11 c3", 16, 256, 768, 0, 1, 1),
         ("G b11 c2", 16, 256, 768, 1, 1, 1),
         ("G b10 c2/3", 16, 128, 384, 0, 1, 1),
         ("G b9 c3", 32, 128, 384, 0, 1, 1),
         ("G b9 c2", 32, 128, 384, 1, 1, 1),
         ("G b8 c2/3", 32, 64, 192, 0, 1, 1)]
only = os.environ.get("B3_ONLY")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, C, Hc, Wc, rs, aff, eff in CASES:
    if only and only not in name:
        continue
    Hs, Ws = (Hc // 2, Wc // 2) if rs == 1 else (Hc, Wc)
    x = torch.randn(N, Hs, Ws, C, device=dev).to(BF)
    g = torch.randn(N, Hc, Wc, C, device=dev).to(BF)
    y = torch.randn(N, Hc, Wc, C, device=dev).to(BF) if eff else None
    dstat = 0.01 * torch.randn(1, 2, C, device=dev) if eff else None
    sc = (1 + 0.2 * torch.randn(N, C, device=dev)) if aff else None
    sh = 0.2 * torch.randn(N, C, device=dev) if aff else None
    kpad = ops._kpad(9 * C)
    wb = torch.randn(C, kpad, device=dev).to(BF)
    dx = torch.empty(N, Hs, Ws, C, device=dev, dtype=BF)
    acc = torch.zeros(N, 8, 2, C, device=dev) if aff else None
    dw = torch.zeros(C, kpad, device=dev)
    cs = torch.zeros(32, C, device=dev)
    P, Ps = N * Hc * Wc, N * Hs * Ws
    moved = 2.0 * (P * C * (1 + eff) + 2 * Ps * C)
    d = H.Conv3x3BwdDesc(N, Hc, Wc, C, kpad, H.src_desc(x, C, Hs, Ws, rs, sc, sh, C if aff else 0, True), g.data_ptr(), C, H.ptr(y), H.ptr(dstat), N,
                         wb.data_ptr(), dx.data_ptr(), H.ptr(acc), dw.data_ptr(), None, cs.data_ptr(), 0)
    fused_ok = H.lib().ieagan_conv3x3_bwd_supported(C, rs, aff, 1, eff, Hc, Wc)
    us = float("nan")
    if fused_ok:
        ws = torch.empty(H.lib().ieagan_conv3x3_bwd_workspace(d), device=dev)
        d.partials = ws.data_ptr()
        us = timed(lambda: H.call("ieagan_conv3x3_bwd", d, H.stream()))
    # ---- the separate launches
    geff = torch.empty_like(g)
    da = torch.empty(N, Hc, Wc, C, device=dev, dtype=BF)
    dsc = torch.zeros(N, C, device=dev) if aff else None
    dsh = torch.zeros(N, C, device=dev) if aff else None
    wd = H.WgradDesc(N, Hc, Wc, C, C, 9, kpad, H.src_desc(x, C, Hs, Ws, rs, sc, sh, C if aff else 0, True), geff.data_ptr() if eff else g.data_ptr(), C,
                     dw.data_ptr(), 0, 0, None, None if eff else cs.data_ptr())
    wsn = H.lib().ieagan_conv_wgrad_workspace(wd, 1)
    ws2 = torch.empty(max(wsn, 1), device=dev)
    if wsn > 0:
        wd.partials = ws2.data_ptr()

    def separate():
        gg = g
        if eff:
            H.call("ieagan_effgrad", g.data_ptr(), y.data_ptr(), dstat.data_ptr(), geff.data_ptr(), cs.data_ptr(), P, C, 1, H.stream())
            gg = geff
        if aff and rs == 0:
            ops._conv_launch(gg, C, Hc, Wc, 0, None, None, 0, False, N, Hc, Wc, C, C, 9, kpad, wb, None, None, 0, 0, 0, None, 0, x, dx, acc, npe=1,
                             bnb=(sc, sh, C, True))
        elif rs == 0:
            ops._conv_launch(gg, C, Hc, Wc, 0, None, None, 0, False, N, Hc, Wc, C, C, 9, kpad, wb, None, None, 0, 0, 0, None, 0, x, dx, None)
        else:
            ops._conv_launch(gg, C, Hc, Wc, 0, None, None, 0, False, N, Hc, Wc, C, C, 9, kpad, wb, None, None, 0, 0, 0, None, 0, None, da, None)
            H.call("ieagan_prologue_bwd", da.data_ptr(), x.data_ptr(), C, H.ptr(sc), H.ptr(sh), C if aff else 0, 1, rs, dx.data_ptr(), H.ptr(dsc), H.ptr(dsh),
                   N, Hs, Ws, C, None, 0, 0, 0, 0, H.stream())
        H.call("ieagan_conv_wgrad", wd, 1, H.stream())

    us2 = timed(separate)
    print(f"{name:12s} C{C} {Hc}x{Wc} rs{rs} a{aff} e{eff}: {moved / 1e6:7.0f} MB | fused {us:7.1f} us {moved / us / 1e6:5.2f} TB/s | separate {us2:7.1f} us", flush=True)
