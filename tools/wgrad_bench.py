"""Micro-benchmark of one conv weight-gradient launch (development aid): python tools/wgrad_bench.py N H W Cin Cout taps [iters]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H
if os.environ.get('WB_LIB'):
    H.LIB_PATH = os.environ['WB_LIB']
import ops

N, Hh, Ww, Cin, Cout, taps = map(int, sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
H.require_gpu()
dev = "cuda:0"
x = torch.randn(N, Hh, Ww, Cin, device=dev).to(torch.bfloat16)
g = torch.randn(N, Hh, Ww, Cout, device=dev).to(torch.bfloat16)
kpad = ops._kpad(taps * Cin)
dw = torch.zeros(Cout, kpad, device=dev)


two = int(os.environ.get("WB_TWO", "0"))
# the descriptor is built ONCE and the C entry point is called directly in the timed loop: building a ctypes struct per call costs
# ~40 us of host time, which would be what a short kernel's "duration" measures
d = H.WgradDesc(N, Hh, Ww, Cin, Cout, taps, kpad, H.src_desc(x, Cin, Hh, Ww, 0, None, None, 0, True), g.data_ptr(), Cout, dw.data_ptr(), 0, 0, None)
ws = None
if two:
    n = H.lib().ieagan_conv_wgrad_workspace(d, 1)
    if n > 0:
        ws = torch.empty(n, device=dev)
        d.partials = ws.data_ptr()
fn = H.lib().ieagan_conv_wgrad
st = H.stream()
iters = int(os.environ.get('WB_ITERS', max(iters, 200)))
for _ in range(10):
    fn(d, 1, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn(d, 1, st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
flops = 2.0 * N * Hh * Ww * Cout * taps * Cin
byts = 2.0 * N * Hh * Ww * (Cin + Cout)
print(f"wgrad two{two} {taps}tap N{N} {Hh}x{Ww} {Cin}->{Cout}: {us:.1f} us  {flops/us/1e6:.0f} TF  {byts/us/1e3:.0f} GB/s")
