"""Where do the device-to-device copies of one eager train step come from?  (development aid)"""
import os, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "iea-gan_amd"), root]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
import model, train_fns, utils

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
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy") and e.device_time_total >= 0:
        st = [s for s in (e.stack or []) if "iea-gan_amd" in s or "bench.py" in s]
        key = (e.name, str(e.input_shapes)[:60], st[0][-70:] if st else "?")
        cnt[key] += 1
for k, v in cnt.most_common(40):
    print(v, k)
