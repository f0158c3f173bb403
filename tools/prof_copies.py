"""Which aten::copy_ / contiguous / clone calls does one eager train step still issue, and from where?  (development aid)"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train(x, y)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::cat"):
        rows.append((e.count, e.key, str(e.input_shapes)[:90], [s for s in e.stack if "iea-gan_amd" in s or "tests" in s][:3]))
rows.sort(key=lambda r: -r[0])
for c, k, s, stck in rows[:60]:
    print(f"{c:5d} {k:18s} {s}\n        {stck}")
