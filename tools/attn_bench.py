"""Micro-benchmark of the non-local attention core at the production shape (development aid): python tools/attn_bench.py [N Lq Lk dqk dv]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H

N, Lq, Lk, dqk, dv = (list(map(int, sys.argv[1:6])) + [40, 3072, 768, 32, 128][len(sys.argv) - 1:])[:5]
H.require_gpu()
dev, BF = "cuda:0", torch.bfloat16
q = (torch.randn(N, Lq, dqk, device=dev) * 0.7).to(BF)
k = (torch.randn(N, Lk, dqk, device=dev) * 0.7).to(BF)
v = torch.randn(N, Lk, dv, device=dev).to(BF)
go = torch.randn(N, Lq, dv, device=dev).to(BF)
o = torch.empty(N, Lq, dv, device=dev, dtype=BF)
lse = torch.empty(N, Lq, device=dev)
delta = torch.empty(N, Lq, device=dev)
dq, dk, dvv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
st = H.stream()


def fwd():
    H.call("ieagan_nl_attention_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), N, Lq, Lk, dqk, dv, st)


def bwd():
    H.call("ieagan_nl_attention_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), go.data_ptr(), lse.data_ptr(), delta.data_ptr(),
           dq.data_ptr(), dk.data_ptr(), dvv.data_ptr(), N, Lq, Lk, dqk, dv, st)


for name, fn, fl in (("fwd", fwd, 2.0 * N * Lq * Lk * (dqk + dv)), ("bwd (q side + k side)", bwd, 2.0 * N * Lq * Lk * (4 * dqk + 3 * dv))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"nl_attention {name} N{N} Lq{Lq} Lk{Lk} d{dqk}/{dv}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s")
