"""Stand-alone timing of ieagan_rrm_attention_fwd / _bwd (development aid): python tools/rrm_attn_bench.py [B S H hd]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
import _hip as H
H.require_gpu()
B, S, Hh, hd = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (1, 40, 4, 128)))
dev = "cuda:0"
qkv = torch.randn(B, S, Hh * 3 * hd, device=dev)
out = torch.empty(B, S, Hh * hd, device=dev)
att = torch.empty(B, Hh, S, S, device=dev)
dout = torch.randn(B, S, Hh * hd, device=dev)
dqkv = torch.empty_like(qkv)
lib, st = H.lib(), H.stream()


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"B{B} S{S} H{Hh} hd{hd}: fwd {timed(lambda: lib.ieagan_rrm_attention_fwd(qkv.data_ptr(), out.data_ptr(), att.data_ptr(), B, S, Hh, hd, st)):.1f} us"
      f"  bwd {timed(lambda: lib.ieagan_rrm_attention_bwd(qkv.data_ptr(), att.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), B, S, Hh, hd, st)):.1f} us")
