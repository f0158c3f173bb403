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
    assert lib.ieagan_abi_version() == int(re.search(r"#define IEAGAN_ABI_VERSION (\d+)", header).group(1)) == _hip.ABI_VERSION


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
# the epoch plan of train.run: 5 events on 2 ranks -> every rank runs the same number of iterations (one collective
# sequence each); the odd event is dropped instead of leaving one rank alone inside an all-reduce
per_rank = parallel.steps_per_rank(5, world)
mine = list(parallel.shard_events(5, rank, world))[:per_rank]
n_coll = 0
for _ in mine:
    t = torch.ones(4)
    dist.all_reduce(t)
    n_coll += 1
counts = [torch.zeros(1) for _ in range(world)]
dist.all_gather(counts, torch.tensor([float(n_coll)]))
assert all(float(c) == 2.0 for c in counts), counts
print("rank", rank, "ok", mine)
parallel.shutdown()                 # barrier + destroy: what train.run ends with
assert not dist.is_initialized()
"""


def test_every_rank_runs_the_same_number_of_steps():
    from parallel import shard_events, steps_per_rank
    for n, w in ((5, 2), (40, 8), (7, 3), (8, 8), (9, 8)):
        k = steps_per_rank(n, w)
        plans = [list(shard_events(n, r, w))[:k] for r in range(w)]
        assert {len(p) for p in plans} == {n // w} and len(set(sum(plans, []))) == w * k
    assert steps_per_rank(3, 8) == 0            # train.run refuses to start (before any collective)


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


def test_train_step_rejects_batches_that_are_not_whole_events(ref_cfg):
    """``events_per_step = E``: one step consumes E * batch_size images; a single event handed to an E = 2 train function (what
    train.py did before it grouped events) must raise instead of running the loss block on empty per-event slices."""
    import io, contextlib
    import model, train_fns, utils
    cfg = dict(ref_cfg, device="cpu", resolution=64, H_base=1, events_per_step=2, ema=False)
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = model.Generator(**cfg), model.Discriminator(**cfg)
    z_, y_ = utils.prepare_z_y(80, G.dim_z, cfg["n_classes"], device="cpu")
    train = train_fns.GAN_training_function(G, D, model.G_D(G, D), z_, y_, None, {"itr": 0}, cfg, "cpu")
    with pytest.raises(ValueError, match="multiple"):
        train(torch.zeros(40, 1, 64, 64), torch.arange(40))
    with pytest.raises(ValueError, match="labels"):
        train(torch.zeros(80, 1, 64, 64), torch.arange(40))
    z_small, _ = utils.prepare_z_y(40, G.dim_z, cfg["n_classes"], device="cpu")
    train2 = train_fns.GAN_training_function(G, D, model.G_D(G, D), z_small, y_, None, {"itr": 0}, cfg, "cpu")
    with pytest.raises(ValueError, match="latent rows"):
        train2(torch.zeros(80, 1, 64, 64), torch.arange(40).repeat(2))


def test_adam_state_interchanges_with_torch_optim_adam(ref_cfg):
    """G_optim.pth / D_optim.pth are in torch.optim.Adam's own state-dict format in both directions (the reference
    builds optim.Adam over the same parameter order, model.py:410-416, 858-864)."""
    import io, contextlib
    import model
    cfg = dict(ref_cfg, device="cpu", resolution=64, H_base=1)
    with contextlib.redirect_stdout(io.StringIO()):
        D = model.Discriminator(**cfg)
    ref_opt = torch.optim.Adam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), weight_decay=0, eps=cfg["adam_eps"])
    gen = torch.Generator().manual_seed(0)
    for p in D.parameters():
        p.grad = torch.randn(p.shape, generator=gen)
    ref_opt.step()
    ref_sd = ref_opt.state_dict()
    D.optim.load_state_dict(ref_sd)                       # reference-format checkpoint -> flat moment buffers
    ar = D.__dict__["_arena"]
    for i, (p, o, n) in enumerate(ar.param_slices):
        assert torch.equal(D.optim._m[o:o + n].view(p.shape), ref_sd["state"][i]["exp_avg"])
        assert torch.equal(D.optim._v[o:o + n].view(p.shape), ref_sd["state"][i]["exp_avg_sq"])
    ours = D.optim.state_dict()                           # ... and back: loadable by torch.optim.Adam
    assert set(ours) == {"state", "param_groups"} and ours["param_groups"][0]["params"] == ref_sd["param_groups"][0]["params"]
    fresh = torch.optim.Adam(D.parameters(), lr=1.0)
    fresh.load_state_dict(ours)
    st = fresh.state_dict()["state"]
    assert all(torch.equal(st[i]["exp_avg_sq"], ref_sd["state"][i]["exp_avg_sq"]) for i in st) and float(st[0]["step"]) == 1.0
    assert fresh.param_groups[0]["lr"] == cfg["D_lr"] and all(not v["exp_avg"].is_cuda for v in ours["state"].values())
    bad = {"state": {0: ref_sd["state"][0]}, "param_groups": ref_sd["param_groups"]}
    with pytest.raises(ValueError):                       # a mismatching checkpoint must raise, never silently re-zero
        D.optim.load_state_dict(bad)


def test_checkpoint_files_follow_reference_layout(tmp_path, ref_cfg, golden_dir):
    """save_weights / load_weights / write_metadata / get_singular_values with the reference's signatures, directory
    layout (<outputroot>/<run_name>/weights/<stem>[_suffix].pth) and key format (utils/__init__.py:572-726)."""
    import io, contextlib
    import model, utils
    cfg = dict(ref_cfg, device="cpu", resolution=64, H_base=1, outputroot=str(tmp_path), run_name="r")
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = model.Generator(**cfg), model.Discriminator(**cfg)
        G_ema = model.Generator(**dict(cfg, skip_init=True, no_optim=True))
        state = {"itr": 7, "epoch": 1, "save_num": 0, "save_best_num": 0, "best_FID": 999999}
        utils.save_weights(G, D, state, cfg, "copy7", G_ema)
        utils.write_metadata(cfg, state)
    wdir = tmp_path / "r" / "weights"
    assert sorted(os.listdir(wdir)) == ["D_copy7.pth", "D_optim_copy7.pth", "G_copy7.pth", "G_ema_copy7.pth", "G_optim_copy7.pth",
                                        "state_dict_copy7.pth"]
    assert (tmp_path / "r" / "logs" / "metalog.txt").read_text().startswith("datetime: ")
    with contextlib.redirect_stdout(io.StringIO()):
        G2, D2 = model.Generator(**cfg), model.Discriminator(**cfg)
        st2 = {"itr": 0, "epoch": 0, "save_num": 0, "save_best_num": 0, "best_FID": 0}
        utils.load_weights(G2, D2, st2, cfg, weight_name="copy7", G_ema=None, load_optim=True)
    assert st2 == state and all(torch.equal(v, G2.state_dict()[k]) for k, v in G.state_dict().items())
    # legacy key names of older reference checkpoints (transG / transcoder) still load
    legacy = {k.replace("RR_G", "transG"): v for k, v in torch.load(wdir / "G_copy7.pth").items()}
    torch.save(legacy, wdir / "G_copy7.pth")
    with contextlib.redirect_stdout(io.StringIO()):
        utils.load_weights(G2, None, {}, cfg, weight_name="copy7", load_optim=False)
    svs = utils.get_singular_values(G, "G")
    contract = json.load(open(os.path.join(golden_dir, "state_dict_contract.json")))["G_64x64"]["keys"]
    assert set(svs) == {f"G_{k}".replace(".", "_") for k in contract if "sv" in k} and "G_linear_sv0" in svs


def test_checkpoint_written_by_the_reference_loads(golden_dir, ref_cfg, tmp_path):
    """``tests/golden/ckpt_ref/run/weights/*.pth`` was written by the REFERENCE's ``utils.save_weights`` (reference
    utils/__init__.py:689-726; tests/golden/make_golden_r3.py, tiny G_ch = D_ch = 2 networks).  This package's
    ``utils.load_weights`` must read it: every state-dict entry equal to the reference's (per-key checksums from the same
    script), the iteration counters restored, the (empty) reference-format optimizer state accepted."""
    import io, contextlib, shutil
    import numpy as np
    import model, utils
    src = os.path.join(golden_dir, "ckpt_ref")
    shutil.copytree(src, tmp_path / "out")
    cfg = dict(ref_cfg, device="cpu", resolution=64, H_base=1, G_ch=2, D_ch=2, dim_z=8, hypersphere_dim=32, ema=False,
               outputroot=str(tmp_path / "out"), run_name="run")
    sums = np.load(os.path.join(golden_dir, "ckpt_ref.npz"))
    assert sorted(sums["files"].tolist()) == sorted(os.listdir(tmp_path / "out" / "run" / "weights"))
    with contextlib.redirect_stdout(io.StringIO()):
        G, D = model.Generator(**cfg), model.Discriminator(**cfg)
        st = {"itr": 0, "epoch": 0, "save_num": 0, "save_best_num": 0, "best_IS": 0, "best_FID": 0}
        utils.load_weights(G, D, st, cfg, weight_name=None, G_ema=None, load_optim=True)
    assert st["itr"] == 7 and st["epoch"] == 1 and st["best_FID"] == 999999
    n = 0
    for net, name in ((G, "G"), (D, "D")):
        for k, v in net.state_dict().items():
            ref = sums[f"{name}.{k}"]
            got = np.array([v.double().sum().item(), v.double().abs().sum().item()])
            assert np.allclose(got, ref, rtol=1e-12, atol=1e-12), (name, k, got, ref)
            n += 1
    assert n == len(sums.files) - 1           # same key set as the reference's modules (no extra / missing entries)
    raw = torch.load(tmp_path / "out" / "run" / "weights" / "G.pth", map_location="cpu")
    assert all(torch.equal(raw[k], v) for k, v in G.state_dict().items())


_STUBS = {"torchvision/__init__.py": "from . import transforms, datasets, utils\n",
          "torchvision/transforms.py": "class _T:\n    def __init__(self, *a, **k): pass\nCompose = Pad = ToTensor = Lambda = Normalize = Grayscale = _T\n",
          "torchvision/datasets.py": "", "torchvision/utils.py": "def save_image(*a, **k): pass\n",
          "seaborn.py": "", "boost_histogram.py": "", "cv2.py": ""}


@pytest.mark.skipif(not os.path.isfile("/root/reference/train.py"), reason="needs the reference checkout (this container only)")
def test_reference_train_script_imports_resolve_through_dropin(tmp_path):
    """``python iea-gan_amd/dropin.py <reference>/train.py ...``.  Every import statement of the
    reference's train.py (:12-19) resolves -- the accelerated modules to this package, the host-side bookkeeping
    sub-modules to the checkout -- and the names train.py uses exist with compatible signatures.  torchvision / seaborn
    are absent from this image and stubbed as EMPTY modules for the import (test infrastructure only)."""
    stubs = tmp_path / "stubs"
    for rel, body in _STUBS.items():
        f = stubs / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(body)
    env = dict(os.environ, PYTHONPATH=str(stubs), PYTHONDONTWRITEBYTECODE="1")
    dropin = os.path.join(ROOT, "iea-gan_amd", "dropin.py")
    r = subprocess.run([sys.executable, dropin, "/root/reference/train.py", "--help"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and "--outputroot" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    probe = tmp_path / "probe.py"
    probe.write_text(_PROBE % {"pkg": os.path.join(ROOT, "iea-gan_amd")})
    r = subprocess.run([sys.executable, str(probe)], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and "probe ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


_PROBE = r"""
import inspect, os, sys
sys.path[:0] = [%(pkg)r, "/root/reference"]
import layers, model, train_fns, utils                      # reference train.py:12-15
import utils.configuration as cf                            # :16
from utils.logging import MetricsLogger, Logger             # :17
from utils.dataloader import load_dataset                   # :18
from utils.plot import plot_sim_heatmap                     # :19
pkg = %(pkg)r
for m in (layers, model, train_fns, utils):
    assert os.path.dirname(os.path.abspath(m.__file__)).startswith(pkg), m.__file__
for m in (cf, sys.modules["utils.logging"], sys.modules["utils.dataloader"], sys.modules["utils.plot"]):
    assert m.__file__.startswith("/root/reference/utils/"), m.__file__
b = lambda f, *a, **k: inspect.signature(f).bind(*a, **k)
G = D = E = z = y = sd = cfg = log = object()
b(utils.seed_rng, 1)                                                                          # train.py:38
b(utils.apply_ema, G, E, 0.9999, 10000)                                                       # :51
b(utils.load_weights, G, D, sd, cfg, weight_name=None, G_ema=None, load_optim=True)           # :81
b(utils.write_metadata, cfg, sd)                                                              # :104
b(utils.prepare_z_y, 40, 128, 40, device="cuda", fp16=False, z_dist="normal", threshold=1.0, y_dist="permuted", ngd=False, fixed=True)  # :117
b(train_fns.GAN_training_function, G, D, object(), z, y, None, sd, cfg, "cuda")               # :151
b(utils.get_singular_values, G, "G")                                                          # :178
b(utils.save_and_sample, G, D, E, z, y, z, y, sd, cfg)                                        # :195
b(train_fns.test, G, D, E, sd, cfg, log)                                                      # :236
assert callable(train_fns.dummy_training_function) and hasattr(model, "G_D")
print("probe ok")
"""


def test_integration_doc_stub_matches_the_binding():
    """The ctypes stub printed in INTEGRATION.md declares the same fields, in the same order, as the binding the package uses
    (a struct that is too short would read past its end on the device side of the call)."""
    import re
    import _hip
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for cls, ref in (("ConvDesc", _hip.ConvDesc), ("SrcDesc", _hip.SrcDesc)):
        m = re.search(r"class %s\(C\.Structure\):\s*_fields_ = \[(.*?)\]\s" % cls, text, re.S)
        assert m, cls
        names = re.findall(r'\("(\w+)"', m.group(1))
        assert names == [f[0] for f in ref._fields_], (cls, names)


def test_register_spill_ratchet_of_the_built_library():
    """Scratch (register spills) per kernel family, read from the code-object metadata of the built library (tools/codeobj_notes.py, no GPU
    needed).  Two statements: (1) every instantiated variant of the round-4 kernel conv3x3_bwd is spill-free -- its two occupancy
    classes and the variant it does NOT offer (C = 32 with BatchNorm prologue + effgrad at the same resolution) exist for exactly that
    reason; (2) no OTHER family grows: the counts below are the state of this round (most are variants the step never launches; the
    launched ones are listed in DESIGN section 3), a new spill or a larger one fails here instead of showing up as a slower step."""
    import collections
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import codeobj_notes as cn
    so = os.path.join(root, "iea-gan_amd", "libieagan_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    rows = cn.kernels(so)
    names = cn.demangle([r["name"] for r in rows])
    assert len(rows) > 500, len(rows)
    fam = collections.defaultdict(lambda: [0, 0])        # family -> [kernels with scratch, largest scratch in bytes / lane]
    for r, n in zip(rows, names):
        f = re.sub(r"^void ", "", n).split("<")[0].split("(")[0]
        if r["scratch"]:
            fam[f][0] += 1
            fam[f][1] = max(fam[f][1], r["scratch"])
    for clean in ("conv3x3_bwd_kernel", "conv_wgrad_kernel", "conv3x3_lds_kernel"):         # spill-free families stay spill-free
        assert clean not in fam, (clean, fam[clean])
    allowed = {"conv1x1_bwd_kernel": (15, 320), "conv1x1_stream_kernel": (5, 340), "conv1x1_tile_kernel": (2, 12),
               "conv3x3_halo_kernel": (19, 104), "conv3x3_lds_fp8_kernel": (11, 96),
               "conv3x3_ws_kernel": (2, 84), "conv_gather_kernel": (7, 196)}
    for f, (cnt, worst) in fam.items():
        assert f in allowed, (f, cnt, worst)
        assert cnt <= allowed[f][0] and worst <= allowed[f][1], (f, cnt, worst, allowed[f])


def test_launched_kernels_are_spill_free_except_the_listed_ones():
    """The kernels the benchmark step really LAUNCHES (names from the committed rocprofv3 kernel trace of the newest round) against the
    code-object notes of the built library: scratch (spilled registers) only in the kernels listed here, and no more than listed."""
    import csv
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import codeobj_notes as cn
    so = os.path.join(root, "iea-gan_amd", "libieagan_hip.so")
    def newest_last(path):                  # rNN_ (end of round) after rNNa_ / rNN_mid_ (mid-round sets)
        m = re.match(r"r(\d+)([a-z]*)_(mid_|final_|baseline_)?kernel_stats", os.path.basename(path))
        return (int(m.group(1)), m.group(2) == "" and m.group(3) in (None, "final_"), m.group(2), m.group(3) or "") if m else (-1, False, "", "")
    traces = sorted(glob.glob(os.path.join(root, "profiles", "r0*_kernel_stats.csv")), key=newest_last)
    if not os.path.exists(so) or not traces:
        pytest.skip("library or kernel trace missing")
    rows = cn.kernels(so)
    names = cn.demangle([r["name"] for r in rows])
    launched = [r["Name"].strip() for r in csv.DictReader(open(traces[-1]))]
    allowed = {"conv1x1_stream_kernel<false, false, 4, 2, 2>": 12, "conv1x1_bwd_kernel<64, 16, 0, true, 2, 64>": 36,
               "conv1x1_bwd_kernel<16, 64, 0, true, 2, 64>": 20, "conv3x3_halo_kernel<true, true, 0, 2, 6, 32, false>": 44,
               "conv3x3_halo_kernel<false, false, 0, 2, 6, 32, true>": 36, "conv3x3_halo_kernel<true, true, 1, 2, 6, 32, false>": 44}
    matched, bad = 0, []
    for key in launched:                    # (the trace keeps the first 140 characters of a name: prefix match)
        ms = [r for n, r in zip(names, rows) if n.startswith(key) or r["name"].startswith(key)]
        if not ms:
            continue                        # not ours (ATen / rocBLAS / copy kernels)
        matched += 1
        worst = max(r["scratch"] for r in ms)
        if worst:
            short = re.sub(r"^void ", "", key).split("(")[0]
            if short not in allowed or worst > allowed[short]:
                bad.append((short, worst))
    assert matched >= 100, matched
    assert not bad, bad


def test_weight_gradient_pixel_splits_follow_the_traffic_rule():
    """The launch plan of ieagan_conv_wgrad (pure host code: runs without a GPU) at the production shapes: the partial dW slabs a launch writes
    and reads back never weigh more than twice its operands -- unless that would make a block walk more than 16 tiles -- and small dW x few
    splits stays with direct accumulation (no workspace).  DESIGN section 3, conv_wgrad (b)."""
    import _hip
    if not os.path.exists(_hip.LIB_PATH):
        pytest.skip("library not built")
    lib = ctypes.CDLL(_hip.LIB_PATH)
    lib.ieagan_conv_wgrad_workspace.restype = ctypes.c_long
    lib.ieagan_conv_wgrad_workspace.argtypes = [ctypes.POINTER(_hip.WgradDesc), ctypes.c_int]
    N = 40
    shapes = [(64, 64, 32, 96, 9, 0), (256, 64, 32, 96, 1, 0), (64, 64, 64, 192, 9, 0), (128, 128, 8, 24, 9, 0), (128, 128, 16, 48, 9, 0),
              (128, 32, 64, 192, 1, 0), (512, 128, 8, 24, 1, 0), (32, 128, 64, 192, 1, 2), (128, 512, 4, 12, 1, 0), (16, 16, 256, 768, 9, 0)]
    seen_direct = seen_slabs = 0
    for cin, cout, h, w, taps, rs in shapes:
        hs, ws = (2 * h, 2 * w) if rs == 2 else (h, w)
        kpad = ((taps * cin + 31) // 32) * 32
        src = _hip.SrcDesc(16, cin, hs, ws, rs, None, None, 0, 1)
        d = _hip.WgradDesc(N, h, w, cin, cout, taps, kpad, src, 16, cout, 16, 0, 0, None, None)
        elems = lib.ieagan_conv_wgrad_workspace(ctypes.byref(d), 1)
        d.partials = 16                                   # "a workspace exists": the plan of the two-stage form
        elems2 = lib.ieagan_conv_wgrad_workspace(ctypes.byref(d), 1)
        assert elems2 > 0 and elems2 % (cout * kpad) == 0, (cin, cout, h, w, elems2)
        splits = elems2 // (cout * kpad)
        tiles = N * ((h + 7) // 8) * ((w + 15) // 16)
        in_bytes = 2.0 * N * (hs * ws * cin + h * w * cout)
        slab_rw = 2.0 * splits * 4.0 * cout * kpad
        assert slab_rw <= 2.0 * in_bytes * 1.02 or splits <= max(2, -(-tiles // 16)), (cin, cout, h, w, splits, slab_rw / in_bytes)
        assert -(-tiles // splits) <= 64, (cin, cout, h, w, splits, tiles)            # no block walks the whole map either
        if elems == 0:
            seen_direct += 1
            assert splits * 4 * cout * taps * cin < 8 << 20, (cin, cout, h, w)          # direct accumulation only below 8 MB of adds
        else:
            seen_slabs += 1
    assert seen_direct >= 2 and seen_slabs >= 4, (seen_direct, seen_slabs)
