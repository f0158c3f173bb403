"""Which library (aten / hipBLASLt) ops does one eager train step still issue?  (development aid)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
import model, train_fns, utils, ops

cfg = bench.bench_config()
cfg["hip_graph"] = False
utils.seed_rng(0)
G = model.Generator(**cfg).cuda(); D = model.Discriminator(**cfg).cuda()
G_ema = model.Generator(**dict(cfg, skip_init=True, no_optim=True)).cuda()
ema = utils.apply_ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
z_, y_ = utils.prepare_z_y(40, G.dim_z, cfg["n_classes"], device="cuda")
train = train_fns.GAN_training_function(G, D, model.G_D(G, D), z_, y_, ema, {"itr": 1}, cfg, "cuda")
x = torch.randn(40, 1, 256, 768, device="cuda"); y = torch.arange(40, device="cuda")
for _ in range(2):
    train(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    train(x, y)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.self_device_time_total > 0:
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten self device time total {tot/1e3:.2f} ms")
for t, c, k, s in rows[:70]:
    print(f"{t/1e3:8.3f} ms {c:5d}  {k:32s} {s}")
