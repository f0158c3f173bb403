"""Stand-alone timing of ieagan_conv_wgrad (two-stage form, as the step launches it) at the production shapes that are NOT covered by the
fused backward kernels, N = 40 (development aid).  WB_LIB=path: another build of the library, for A/B on one box."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H
if os.environ.get("WB_LIB"):
    H.LIB_PATH = os.environ["WB_LIB"]
import ops

H.require_gpu()
dev = "cuda:0"
N = 40
#         Cin Cout  H    W  taps rs  launches/step
SHAPES = [(64, 64, 32, 96, 9, 0, 11), (256, 64, 32, 96, 1, 0, 6), (64, 64, 64, 192, 9, 0, 5), (128, 128, 8, 24, 9, 0, 11),
          (128, 128, 16, 48, 9, 0, 5), (128, 32, 64, 192, 1, 0, 4), (512, 128, 8, 24, 1, 0, 6), (64, 64, 16, 48, 9, 0, 6),
          (32, 128, 64, 192, 1, 0, 3), (128, 64, 64, 192, 1, 0, 2), (256, 128, 32, 96, 1, 0, 2), (128, 128, 4, 12, 9, 0, 6),
          (256, 32, 32, 96, 1, 0, 4), (32, 128, 64, 192, 1, 2, 2), (64, 256, 32, 96, 1, 0, 4), (64, 64, 64, 192, 1, 2, 2),
          (128, 128, 32, 96, 1, 2, 2), (512, 128, 4, 12, 1, 0, 4), (128, 512, 8, 24, 1, 0, 4), (128, 256, 32, 96, 1, 0, 2),
          (256, 64, 16, 48, 1, 0, 4), (64, 256, 32, 96, 1, 2, 2), (256, 256, 8, 24, 1, 2, 2), (256, 128, 16, 48, 1, 0, 2),
          (64, 128, 64, 192, 1, 0, 1), (128, 512, 4, 12, 1, 0, 3), (64, 256, 16, 48, 1, 0, 3), (128, 512, 8, 24, 1, 2, 2)]
only = os.environ.get("WB_ONLY")
tot = 0.0
for Cin, Cout, Hh, Ww, taps, rs, cnt in SHAPES:
    if only and only != f"{Cin}x{Cout}":
        continue
    Hs, Ws = (2 * Hh, 2 * Ww) if rs == 2 else (Hh, Ww)
    x = torch.randn(N, Hs, Ws, Cin, device=dev).to(torch.bfloat16)
    g = torch.randn(N, Hh, Ww, Cout, device=dev).to(torch.bfloat16)
    kpad = ops._kpad(taps * Cin)
    dw = torch.zeros(Cout, kpad, device=dev)
    d = H.WgradDesc(N, Hh, Ww, Cin, Cout, taps, kpad, H.src_desc(x, Cin, Hs, Ws, rs, None, None, 0, True), g.data_ptr(), Cout, dw.data_ptr(), 0, 0, None)
    n = H.lib().ieagan_conv_wgrad_workspace(d, 1)
    ws = torch.empty(max(n, 1), device=dev)
    if n > 0:
        d.partials = ws.data_ptr()
    fn, st = H.lib().ieagan_conv_wgrad, H.stream()
    for _ in range(5):
        fn(d, 1, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn(d, 1, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    mb = 2.0 * N * (Hs * Ws * Cin + Hh * Ww * Cout) / 1e6
    tot += us * cnt
    print(f"{taps}tap {Cin:3d}->{Cout:3d} {Hh:3d}x{Ww:3d} rs{rs} x{cnt:2d}: {us:7.1f} us  {mb / us * 1e3:6.0f} GB/s  slabs {n * 4 / 1e6:6.1f} MB")
print(f"sum over a step: {tot / 1e3:.3f} ms")
