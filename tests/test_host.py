"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the product modules
honour the reference's state-dict contract, and the data-parallel gradient exchange is correct
(world_size 2, gloo)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_hip.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "ieagan_hip.h")).read()
    declared = set(re.findall(r"\b(ieagan_[A-Za-z0-9_]+)\s*\(", header))
    declared = {d for d in declared if not d.endswith("_desc") and not d.endswith("_rec")}
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/ieagan_hip.h but not exported"
    assert set(_hip.EXPORTS) <= declared
    lib.ieagan_abi_version.restype = ctypes.c_int
    assert lib.ieagan_abi_version() == int(re.search(r"#define IEAGAN_ABI_VERSION (\d+)", header).group(1)) == 2


def test_product_modules_match_reference_state_dict_contract(golden_dir, ref_cfg):
    import io, contextlib
    import model
    contract = json.load(open(os.path.join(golden_dir, "state_dict_contract.json")))
    for tag, over in (("256x768", {}), ("64x64", {"resolution": 64, "H_base": 1})):
        cfg = dict(ref_cfg, device="cpu", **over)
        with contextlib.redirect_stdout(io.StringIO()):
            G, D = model.Generator(**cfg), model.Discriminator(**cfg)
        for net, name in ((G, "G"), (D, "D")):
            ent = contract[f"{name}_{tag}"]
            assert {k: list(v.shape) for k, v in net.state_dict().items()} == ent["keys"]
            assert sorted(k for k, _ in net.named_parameters()) == ent["params"]
            assert sum(p.numel() for p in net.parameters()) == ent["n_params"]
        assert hasattr(G, "optim") and hasattr(D, "optim") and G.lr_sched is None and G.dim_z == 128
        for attr in ("shared", "RR_G", "linear_f", "linear", "blocks", "output_layer"):
            assert hasattr(G, attr)
        for attr in ("embed", "linear0", "linear1", "RR_D", "norm", "blocks", "input_conv"):
            assert hasattr(D, attr)


def test_product_path_fails_loudly_without_gpu(ref_cfg):
    """No silent CPU fallback: on a machine without a HIP device the forward raises."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import io, contextlib
    import model
    cfg = dict(ref_cfg, device="cpu", resolution=64, H_base=1)
    with contextlib.redirect_stdout(io.StringIO()):
        G = model.Generator(**cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G(torch.randn(40, 128), torch.arange(40))


def test_arena_flattens_and_keeps_state_dict(ref_cfg):
    import io, contextlib
    import model
    from arena import Arena
    cfg = dict(ref_cfg, device="cpu", resolution=64, H_base=1)
    with contextlib.redirect_stdout(io.StringIO()):
        D = model.Discriminator(**cfg)
    before = {k: v.clone() for k, v in D.state_dict().items()}
    ar = Arena(D)
    after = D.state_dict()
    assert list(before) == list(after)
    for k in before:
        assert torch.equal(before[k], after[k]) and ar.contains(after[k])
    g = ar.attach_grads()
    assert ar.grads_attached() and g.numel() == ar.n_param
    next(D.parameters()).grad.add_(1.0)
    assert float(g.sum()) == next(D.parameters()).numel()


def test_shard_events_is_balanced_partition():
    from parallel import shard_events
    for n, w in ((40, 8), (7, 3), (8, 8), (3, 8)):
        parts = [list(shard_events(n, r, w)) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%(root)r, "iea-gan_amd"))
import parallel
rank, world, _ = parallel.init_from_env("gloo")
torch.manual_seed(100 + rank)
flat = torch.randn(1000)            # stands for one network's state arena
parallel.broadcast_flat(flat)       # every rank starts from rank 0's state
grad = torch.full((1000,), float(rank + 1))
sync = parallel.GradSync(overlap=False)
stepped = []
sync.reduce_then("D", grad, lambda: stepped.append(grad.clone()))
expect = sum(range(1, world + 1)) / world
assert torch.allclose(stepped[0], torch.full((1000,), expect)), stepped[0][:4]
gathered = [torch.zeros(1000) for _ in range(world)]
dist.all_gather(gathered, flat)
assert all(torch.equal(g, gathered[0]) for g in gathered)
ev = list(parallel.shard_events(5, rank, world))
print("rank", rank, "ok", ev)
dist.destroy_process_group()
"""


def test_data_parallel_grad_sync_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2, r.stdout + r.stderr


def test_train_entry_point_config_merge(tmp_path):
    """defaults <- config.json <- command line, typed after the defaults (the reference's SUPPRESS-style merge)."""
    import json as _json
    import train
    cfgfile = tmp_path / "config.json"
    cfgfile.write_text(_json.dumps({"G_lr": 1e-4, "resolution": 128}))
    cfg = train.parse(["--config", str(cfgfile), "--synthetic", "3", "--resolution", "64", "--H_base", "1", "--clip_norm", "1e9",
                       "--ema", "false", "--max_iters", "2", "--device", "cuda"])
    assert cfg["resolution"] == 64 and cfg["H_base"] == 1 and cfg["G_lr"] == 1e-4 and cfg["clip_norm"] == 1e9
    assert cfg["ema"] is False and cfg["synthetic"] == 3 and cfg["D_lr"] == 5e-5
    ev = torch.from_numpy(train.synthetic_event(40, 58, 64, 1))
    assert ev.dtype == torch.uint8 and ev.shape == (40, 58, 64)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):      # ingestion is a HIP kernel: it must fail loudly here
            train.to_network_range(ev, 64)
    xf = train.to_network_range(torch.zeros(40, 64, 64), 64)            # float events are taken as already normalised
    assert xf.shape == (40, 1, 64, 64)
    with pytest.raises(SystemExit):
        train.parse(["--no_such_option", "1"])
