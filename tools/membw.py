"""Plain HBM bandwidth probes on this GPU (fill = write only, copy = read + write, sum = read only)."""
import torch, time
dev = "cuda:0"
for mb in (252, 1024, 4096):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev)
    b = torch.empty(n, dtype=torch.bfloat16, device=dev)
    def timeit(fn, iters=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / iters
    t = timeit(lambda: a.fill_(1.0)); print(f"{mb} MB fill : {mb/1024/t/1e0*1.0737:.2f} GB/ms -> {mb*1.048576e6/t/1e12:.2f} TB/s write")
    t = timeit(lambda: b.copy_(a));   print(f"{mb} MB copy : {2*mb*1.048576e6/t/1e12:.2f} TB/s read+write")
    t = timeit(lambda: a.float().sum() if False else torch.sum(a.view(torch.int16)[: n], dtype=torch.int32)); print(f"{mb} MB sum  : {mb*1.048576e6/t/1e12:.2f} TB/s read")
