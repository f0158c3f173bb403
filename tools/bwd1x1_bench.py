"""Stand-alone timing of ieagan_conv1x1_bwd at the production shapes (N = 40), 2 vs 3 blocks per CU (development aid)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H
import ops

BF = torch.bfloat16
dev = torch.device("cuda:0")
N = int(os.environ.get("B1_N", 40))
#        name          Cin Cout  H    W   rs aff relu eff  link(mode, lC, lCa)  out_mode  geff
CASES = [("D s0.0 c1", 32, 16, 256, 768, 0, 0, 0, 0, (2, 32, 32), 0, 0),
         ("D s0.0 c4", 16, 64, 128, 384, 2, 0, 1, 0, None, 0, 0),
         ("D s0.0 sc", 32, 32, 128, 384, 2, 0, 0, 0, (0, 64, 32), 1, 0),
         ("D s0.1 c1", 64, 16, 128, 384, 0, 0, 1, 0, (0, 64, 64), 0, 0),
         ("D s0.1 c4", 16, 64, 128, 384, 0, 0, 1, 0, None, 0, 0),
         ("D s1.0 c1", 64, 32, 128, 384, 0, 0, 1, 0, (2, 64, 64), 0, 0),
         ("G b11 c1", 64, 16, 128, 384, 0, 1, 1, 1, (1, 32, 32), 0, 0),
         ("G b11 c4", 16, 32, 256, 768, 0, 1, 1, 1, None, 0, 1),
         ("G b10 c1", 64, 16, 128, 384, 0, 1, 1, 1, (0, 64, 64), 0, 0),
         ("G b10 c4", 16, 64, 128, 384, 0, 1, 1, 1, None, 0, 1),
         ("G b9 c4", 32, 64, 128, 384, 0, 1, 1, 1, None, 0, 1)]
for name, Cin, Cout, Hc, Wc, rs, aff, relu, eff, link, om, gout in CASES:
    Hs, Ws = (2 * Hc, 2 * Wc) if rs == 2 else (Hc, Wc)
    x = torch.randn(N, Hs, Ws, Cin, device=dev).to(BF)
    g = torch.randn(N, Hc, Wc, Cout, device=dev).to(BF)
    y = torch.randn(N, Hc, Wc, Cout, device=dev).to(BF) if eff else None
    dstat = 0.01 * torch.randn(1, 2, Cout, device=dev) if eff else None
    geff = torch.empty_like(g) if gout else None
    sc = (1 + 0.2 * torch.randn(N, Cin, device=dev)) if aff else None
    sh = 0.2 * torch.randn(N, Cin, device=dev) if aff else None
    kpad, kpad2 = ops._kpad(Cin), ops._kpad(Cout)
    wb = torch.randn(Cin, kpad2, device=dev).to(BF)
    lg = None
    lmode = lC = lCa = 0
    if link is not None:
        lmode, lC, lCa = link
        Hd, Wd = (Hc, Wc) if om == 1 else (Hs, Ws)
        Hl, Wl = (Hd, Wd) if lmode == 0 else (2 * Hd, 2 * Wd) if lmode == 1 else (Hd // 2, Wd // 2)
        lg = torch.randn(N, Hl, Wl, lC, device=dev).to(BF)
    dx = torch.empty((N, Hc, Wc, Cin) if om == 1 else (N, Hs, Ws, Cin), device=dev, dtype=BF)
    acc = torch.zeros(N, 8, 2, Cin, device=dev) if aff else None
    dw = torch.zeros(Cout, kpad, device=dev)
    cs = torch.zeros(32, Cout, device=dev)
    P, Ps = N * Hc * Wc, N * Hs * Ws
    moved = 2.0 * (P * Cout * (1 + eff + gout) + Ps * Cin + (P if om == 1 else Ps) * Cin)
    if link is not None:
        moved += 2.0 * (P if om == 1 else Ps) * lCa * (4 if lmode == 1 else 0.25 if lmode == 2 else 1)
    line = f"{name:10s} {Cin:3d}->{Cout:3d} {Hc}x{Wc} rs{rs} a{aff} e{eff} l{lmode if link else -1}: {moved / 1e6:7.0f} MB "
    for flags in (H.B1_OCC2 | H.B1_TP32, H.B1_OCC2, 0):
        d = H.Conv1x1BwdDesc(N, Hc, Wc, Cin, Cout, kpad, kpad2, H.src_desc(x, Cin, Hs, Ws, rs, sc, sh, Cin if aff else 0, relu), g.data_ptr(), Cout,
                             H.ptr(y), H.ptr(dstat), N, H.ptr(geff), wb.data_ptr(), H.ptr(lg), lC, lCa, lmode, dx.data_ptr(), om, H.ptr(acc),
                             dw.data_ptr(), None, cs.data_ptr(), flags)
        ws_n = H.lib().ieagan_conv1x1_bwd_workspace(d)
        ws = torch.empty(max(ws_n, 1), device=dev)
        if ws_n > 0:
            d.partials = ws.data_ptr()
        for _ in range(3):
            H.call("ieagan_conv1x1_bwd", d, H.stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            H.call("ieagan_conv1x1_bwd", d, H.stream())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f"| {('tp32' if flags & H.B1_TP32 else 'occ2' if flags else 'dflt')} {us:7.1f} us {moved / us / 1e6:5.2f} TB/s "
    print(line, flush=True)
