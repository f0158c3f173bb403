"""Micro-benchmark of one conv weight-gradient launch (development aid): python tools/wgrad_bench.py N H W Cin Cout taps [iters]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H
import ops

N, Hh, Ww, Cin, Cout, taps = map(int, sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
H.require_gpu()
dev = "cuda:0"
x = torch.randn(N, Hh, Ww, Cin, device=dev).to(torch.bfloat16)
g = torch.randn(N, Hh, Ww, Cout, device=dev).to(torch.bfloat16)
kpad = ops._kpad(taps * Cin)
dw = torch.zeros(Cout, kpad, device=dev)


def run():
    d = H.WgradDesc(N, Hh, Ww, Cin, Cout, taps, kpad, H.src_desc(x, Cin, Hh, Ww, 0, None, None, 0, True), g.data_ptr(), Cout, dw.data_ptr(), 0)
    H.call("ieagan_conv_wgrad", d, 1, H.stream())


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
flops = 2.0 * N * Hh * Ww * Cout * taps * Cin
byts = 2.0 * N * Hh * Ww * (Cin + Cout)
print(f"wgrad {taps}tap N{N} {Hh}x{Ww} {Cin}->{Cout}: {us:.1f} us  {flops/us/1e6:.0f} TF  {byts/us/1e3:.0f} GB/s")
