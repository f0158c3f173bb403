"""Micro-benchmark of one fused-conv launch (development aid): python tools/conv_bench.py N H W Cin Cout taps [relu] [mask] [iters]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H
if os.environ.get("CB_LIB"):
    H.LIB_PATH = os.environ["CB_LIB"]
import ops

N, Hh, Ww, Cin, Cout, taps = map(int, sys.argv[1:7])
relu = int(sys.argv[7]) if len(sys.argv) > 7 else 0
mask = int(sys.argv[8]) if len(sys.argv) > 8 else 0
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20
aff = int(os.environ.get("CB_AFF", "0"))
res = int(os.environ.get("CB_RES", "0"))          # 0 none, 1 same-res, 2 upsampled (half-res operand), 3 pooled (double-res operand)
want_stats = int(os.environ.get("CB_STATS", "1"))
flags = int(os.environ.get("CB_FLAGS", "0"))
rs = int(os.environ.get("CB_RS", "0"))             # 2: the source lives at double resolution (2x2 average pool in the prologue)       # 1: force gather kernel, 2: conv3x3_halo instead of conv3x3_lds
H.require_gpu()
dev = "cuda:0"
Hs, Ws = (2 * Hh, 2 * Ww) if rs == 2 else (Hh, Ww)
x = torch.randn(N, Hs, Ws, Cin, device=dev).to(torch.bfloat16)
kpad = ops._kpad(taps * Cin)
w = (torch.randn(Cout, kpad, device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn(Cout, device=dev)
out = torch.empty(N, Hh, Ww, Cout, device=dev, dtype=torch.bfloat16)
mk = torch.randn(N, Hh, Ww, Cout, device=dev).to(torch.bfloat16) if mask else None
stats = torch.zeros(32, 2, Cout, device=dev) if want_stats else None
sc = (1 + 0.1 * torch.randn(N, Cin, device=dev)) if aff else None
sh = (0.1 * torch.randn(N, Cin, device=dev)) if aff else None
ra = None
if res == 1:
    ra = torch.randn(N, Hh, Ww, Cout, device=dev).to(torch.bfloat16)
elif res == 2:
    ra = torch.randn(N, Hh // 2, Ww // 2, Cout, device=dev).to(torch.bfloat16)
elif res == 3:
    ra = torch.randn(N, Hh * 2, Ww * 2, Cout, device=dev).to(torch.bfloat16)


# the descriptor is built ONCE and the C entry point is called directly in the timed loop (see wgrad_bench.py)
d = H.ConvDesc(N, Hh, Ww, Cin, Cout, taps, kpad, H.src_desc(x, Cin, Hs, Ws, rs, sc, sh, Cin if aff else 0, bool(relu)), H.ptr(w), H.ptr(bias),
               H.ptr(ra), Cout if ra is not None else 0, Cout if ra is not None else 0, max(res - 1, 0), 1.0, None, 0, H.ptr(mk), H.ptr(out),
               H.ptr(stats), 0, flags, None, None, 0, 0)
fn = H.lib().ieagan_conv_forward
st = H.stream()
iters = max(iters, 200)
for _ in range(10):
    fn(d, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn(d, st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
flops = 2.0 * N * Hh * Ww * Cout * taps * Cin
byts = 2.0 * N * (Hs * Ws * Cin + Hh * Ww * Cout * (2 if mask else 1))
print(f"conv {taps}tap N{N} {Hh}x{Ww} {Cin}->{Cout} relu{relu} mask{mask} aff{aff} res{res} rs{rs} stats{want_stats} flags{flags}: {us:.1f} us  {flops/us/1e6:.0f} TF  {byts/us/1e3:.0f} GB/s")
