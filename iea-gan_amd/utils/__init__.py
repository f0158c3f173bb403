"""Train-step utilities behind the reference's ``utils`` surface (reference ``utils/__init__.py``:
Distribution 41-120, prepare_z_y 124-158, seed_rng 218-226, toggle_grad 261-263, make_mask 266-275,
apply_ema 809-837, ortho 843-859).  Host-side bookkeeping of the reference (sample sheets, plots,
Inception statistics) is out of scope of the MI355X hot path.  Checkpoint I/O, metadata and singular-value
logging keep the reference's signatures and on-disk layout (load_weights 592, write_metadata 671, save_weights 689,
save_and_sample 299, get_singular_values 572).
"""
from __future__ import annotations

import os as _os
import sys as _sys

import numpy as np
import torch

import _hip as H


def _fall_through_to_reference_checkout():
    """The reference's scripts also import the host-side bookkeeping sub-modules ``utils.configuration``,
    ``utils.logging``, ``utils.dataloader``, ``utils.plot``, ``utils.norm`` and ``utils.noise`` (reference
    train.py:16-19, utils/__init__.py:30).  They are outside the accelerated path and are not re-implemented here:
    when a reference checkout is on ``sys.path`` behind this package (``dropin.py``), its
    ``utils/`` directory is appended to this package's search path so those sub-modules resolve to the user's own
    files, while every name defined in this file keeps shadowing the reference's ``utils/__init__.py``."""
    here = _os.path.dirname(_os.path.abspath(__file__))
    for p in list(_sys.path):
        cand = _os.path.join(_os.path.abspath(p or "."), "utils")
        if cand != here and cand not in __path__ and _os.path.isfile(_os.path.join(cand, "configuration.py")):
            __path__.append(cand)


_fall_through_to_reference_checkout()


class Distribution(torch.Tensor):
    """Latent / label holder refilled in place by ``sample_()``."""

    def init_distribution(self, dist_type: str, **kwargs):
        self.dist_type, self.dist_kwargs = dist_type, kwargs
        if dist_type in ("normal", "censored_normal"):
            self.mean, self.var = kwargs["mean"], kwargs["var"]
        elif dist_type in ("categorical", "permuted"):
            self.num_categories = kwargs["num_categories"]
        elif dist_type != "bernoulli":
            raise NotImplementedError(f"Distribution '{dist_type}' is not implemented")

    def sample_(self):
        if self.dist_type == "normal":
            self.normal_(self.mean, self.var)
        elif self.dist_type == "censored_normal":
            self.normal_(self.mean, self.var)
            self.relu_()
        elif self.dist_type == "categorical":
            self.random_(0, self.num_categories)
        elif self.dist_type == "bernoulli":
            self.bernoulli_()
        elif self.dist_type == "permuted":
            self.copy_(torch.randperm(self.num_categories, device=self.device))
        else:
            raise NotImplementedError(f"Distribution '{self.dist_type}' is not implemented")

    def to(self, *args, **kwargs):
        new_obj = Distribution(self)
        new_obj.init_distribution(self.dist_type, **self.dist_kwargs)
        new_obj.data = super().to(*args, **kwargs)
        return new_obj


def prepare_z_y(G_batch_size, dim_z, nclasses, device="cuda", fp16=False, z_var=1.0, z_dist="normal", threshold=1,
                y_dist="permuted", ngd=False, fixed=False):
    if ngd or fp16:
        raise NotImplementedError("prepare_z_y: ngd / fp16 latents are not part of the MI355X path")
    z_ = Distribution(torch.randn(G_batch_size, dim_z, requires_grad=False))
    if z_dist in ("normal", "censored_normal"):
        z_.init_distribution(z_dist, mean=0, var=z_var)
    elif z_dist == "bernoulli":
        z_.init_distribution(z_dist)
    else:
        raise NotImplementedError(f"z_dist {z_dist}")
    z_ = z_.to(device, torch.float32)
    y_ = Distribution(torch.zeros(G_batch_size, requires_grad=False))
    y_.init_distribution("categorical" if y_dist == "categorical" else "permuted", num_categories=nclasses, device=device)
    y_ = y_.to(device, torch.int64)
    return z_, y_


def seed_rng(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)


def toggle_grad(model, on_or_off):
    for p in model.parameters():
        p.requires_grad = on_or_off


def make_mask(labels, n_cls, device):
    """[n_cls, n_samples] one-hot of the labels, built on the device (no host round trip)."""
    return (torch.arange(n_cls, device=labels.device)[:, None] == labels[None, :]).to(torch.long).to(device)


class apply_ema(object):
    """EMA of ``source`` into ``target`` over every state-dict entry (parameters AND buffers): one fused
    launch over the two flat arenas."""

    def __init__(self, source, target, decay=0.9999, start_itr=0):
        self.source, self.target, self.decay, self.start_itr = source, target, decay, start_itr
        self._decay_dev, self._decay_host = None, None
        print("Initializing EMA parameters to be source parameters...")
        with torch.no_grad():
            tgt = self.target.state_dict()
            for k, v in self.source.state_dict().items():
                tgt[k].copy_(v)

    def _arenas(self):
        from arena import Arena
        out = []
        for net in (self.source, self.target):
            a = net.__dict__.get("_arena")
            probe = next(net.parameters())
            if a is None or a.root is not net or not a.contains(probe):
                a = Arena(net)
                if hasattr(net, "_plan"):
                    net._plan = None
            out.append(a)
        return out

    def decay_dirty(self, itr=None):
        """True when ``push_decay(itr)`` would write the device scalar."""
        decay = 0.0 if (itr and itr < self.start_itr) else self.decay
        return self._decay_dev is None or decay != self._decay_host

    def push_decay(self, itr=None):
        """Mirror the decay for iteration ``itr`` into the device scalar the kernel reads (no-op when
        unchanged); called by ``update`` and once per replay when the step runs from a HIP graph."""
        decay = 0.0 if (itr and itr < self.start_itr) else self.decay
        if self._decay_dev is None:
            self._decay_dev = torch.tensor([decay], dtype=torch.float32, device=next(self.source.parameters()).device)
        elif decay != self._decay_host:
            self._decay_dev.fill_(decay)
        self._decay_host = decay

    def update(self, itr=None):
        H.require_gpu()
        src, tgt = self._arenas()
        assert src.flat.numel() == tgt.flat.numel(), "EMA source / target layouts differ"
        if not torch.cuda.is_current_stream_capturing():
            self.push_decay(itr)
        H.call("ieagan_ema_update", tgt.flat.data_ptr(), src.flat.data_ptr(), src.flat.numel(), self._decay_dev.data_ptr(),
               H.stream())


def _ortho_plan(arena, params):
    """Work lists of the batched ortho launch for the 2-D+ parameters ``params`` of one arena (cached per set)."""
    key = tuple(id(p) for p in params)
    plans = arena.__dict__.setdefault("_ortho_plans", {})
    if key in plans:
        return plans[key]
    offs = {id(p): o for p, o, _ in arena.param_slices}
    ksplit = H.lib().ieagan_ortho_ksplit()
    tile = 64
    table, gtiles, atiles, goff = [], [], [], 0
    for li, p in enumerate(params):
        R, K = p.shape[0], p.numel() // p.shape[0]
        M, red = (R, K) if R <= K else (K, R)
        table.append([offs[id(p)], R, K, goff])
        goff += (M * M + 7) // 8 * 8
        nt = (M + tile - 1) // tile
        for ti in range(nt):
            for tj in range(nt):
                for sp in range((red + ksplit - 1) // ksplit):
                    gtiles.append([li, ti, tj, sp])
        for ti in range((R + tile - 1) // tile):
            for tj in range((K + tile - 1) // tile):
                atiles.append([li, ti, tj, 0])
    dev = arena.flat.device
    plan = dict(table=torch.tensor(table, dtype=torch.int64, device=dev), ngt=len(gtiles), nat=len(atiles),
                gtiles=torch.tensor(gtiles, dtype=torch.int32, device=dev), atiles=torch.tensor(atiles, dtype=torch.int32, device=dev),
                gram=torch.empty(goff, dtype=torch.float32, device=dev))
    plans[key] = plan
    return plan


def ortho(model, strength=1e-4, blacklist=None):
    """Modified orthogonal regularisation added straight to ``param.grad`` (reference utils/__init__.py:843-859):
    grad += 2*strength * ((W W^T) (.) (1 - I)) W for every parameter with >= 2 dims outside ``blacklist`` -- ONE batched
    HIP call over the network's flat arena (csrc/ortho.hip).  Tall matrices are evaluated as
    W (W^T W) - diag(|w_i|^2) W so the Gram matrix is [in, in] (256x256 for G.linear, not 24576x24576)."""
    H.require_gpu()
    from arena import arena_of
    blacklist = blacklist or []
    arena = arena_of(model)
    if not arena.grads_attached():                      # stand-alone module: move its gradients into one flat buffer
        old = [(p, p.grad) for p, _, _ in arena.param_slices]
        arena.attach_grads()
        with torch.no_grad():
            for p, g in old:
                if g is not None:
                    p.grad.copy_(g)
    params = [p for p in model.parameters() if p.dim() >= 2 and not any(p is b for b in blacklist) and p.grad is not None]
    if not params:
        return
    plan = _ortho_plan(arena, params)
    H.call("ieagan_ortho_grad", arena.flat.data_ptr(), arena.grad.data_ptr(), plan["table"].data_ptr(), plan["gtiles"].data_ptr(),
           plan["ngt"], plan["atiles"].data_ptr(), plan["nat"], plan["gram"].data_ptr(), plan["gram"].numel(), float(strength),
           H.stream())


# ---------------------------------------------------------------------------------------------------------
# the steps either side of the train step
# ---------------------------------------------------------------------------------------------------------
def ingest_event(ev_u8, noise=None, scale=4e-3, pad=3, device=None):
    """uint8 sensor images of one event ``[N, Hin, W]`` (host or device) -> the network input fp32 ``[N, 1, Hin+2*pad, W]``
    on the GPU: the reference's ``Pad -> ToTensor -> fn_lognorm255 -> UniformNoise(4e-3) -> Normalize(0.5, 0.5)`` chain
    (utils/dataloader.py:66-77) as one HIP kernel.  Only the uint8 pixels cross PCIe (7.7 MB per 40x250x768 event
    instead of 31.5 MB of fp32).  ``noise``: explicit U[0,1) draws ``[N, Hin+2*pad, W]`` (parity tests); None = drawn on
    the device; False = no dequantisation noise."""
    H.require_gpu()
    dev = torch.device(device) if device is not None else (ev_u8.device if ev_u8.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    if ev_u8.dtype != torch.uint8 or ev_u8.dim() != 3:
        raise TypeError("ingest_event expects a uint8 tensor [N, H, W]")
    ev = ev_u8.to(dev, non_blocking=True).contiguous()
    N, Hin, W = ev.shape
    out = torch.empty(N, 1, Hin + 2 * pad, W, dtype=torch.float32, device=dev)
    if noise is None:
        noise = torch.rand(N, Hin + 2 * pad, W, device=dev)
    elif noise is False:
        noise = None
    else:
        noise = noise.to(dev, torch.float32).contiguous()
        if noise.numel() != out.numel():
            raise ValueError("noise must have the padded shape [N, Hin + 2*pad, W]")
    H.call("ieagan_event_ingest", ev.data_ptr(), H.ptr(noise), out.data_ptr(), N, Hin, W, pad, float(scale), H.stream())
    return out


def feature_statistics(feats):
    """Mean and unbiased covariance (np.cov(rowvar=False)) of a feature matrix [n, d], float64, on the features' device."""
    f = feats.to(torch.float64)
    mu = f.mean(0)
    fc = f - mu
    return mu, fc.t() @ fc / (f.shape[0] - 1)


def frechet_distance(mu1, sigma1, mu2, sigma2):
    """|mu1-mu2|^2 + Tr(S1) + Tr(S2) - 2 Tr((S1 S2)^(1/2)) evaluated on the device (reference: scipy ``sqrtm`` on the host,
    mycleanfid/fid.py:431-468).  For positive semi-definite S1, S2 the trace term equals sum(sqrt(eig(S1^(1/2) S2 S1^(1/2)))),
    two symmetric eigen-decompositions in float64 -- no complex arithmetic and no eps retry for singular products."""
    H.require_gpu()
    mu1, mu2, s1, s2 = (torch.as_tensor(t).to("cuda" if not torch.as_tensor(t).is_cuda else torch.as_tensor(t).device, torch.float64)
                        for t in (mu1, mu2, sigma1, sigma2))
    w1, v1 = torch.linalg.eigh(s1)
    r1 = (v1 * w1.clamp_min(0).sqrt()) @ v1.t()
    w = torch.linalg.eigvalsh(r1 @ s2 @ r1)
    diff = mu1 - mu2
    return float(diff.dot(diff) + torch.trace(s1) + torch.trace(s2) - 2.0 * w.clamp_min(0).sqrt().sum())


def count_parameters(module):
    print("Number of parameters: {}".format(sum(p.data.nelement() for p in module.parameters())))


# ---------------------------------------------------------------------------------------------------------
# checkpoint I/O, metadata and singular-value logging with the reference's signatures, file layout and key format
# (utils/__init__.py: join_strings 229, rename_weight_keys 242, save_and_sample 299, get_singular_values 572,
#  load_weights 592, write_metadata 671, save_weights 689): <outputroot>/<run_name>/{weights,logs,samples}/...
# ---------------------------------------------------------------------------------------------------------
def join_strings(delimiter, strings):
    return delimiter.join([s for s in strings if s])


def rename_weight_keys(state_dict, fragment, replacement):
    from collections import OrderedDict
    return OrderedDict((k.replace(fragment, replacement) if isinstance(k, str) else k, v) for k, v in state_dict.items())


def _run_dir(configuration, sub):
    import pathlib
    return pathlib.Path(configuration["outputroot"]).joinpath(configuration["run_name"]).joinpath(sub)


def _host_copy(sd):
    """State-dict entries are views of the network's flat arena: store detached host copies."""
    return type(sd)((k, v.detach().cpu().clone()) for k, v in sd.items())


def save_weights(G, D, state_dict, configuration, name_suffix=None, G_ema=None):
    """{G, G_optim, D, D_optim, state_dict, G_ema}[_suffix].pth under <outputroot>/<run_name>/weights; the optimizer
    files are in torch.optim.Adam's own state-dict format (``FusedAdam.state_dict``), so either implementation loads
    the other's checkpoint."""
    wdir = _run_dir(configuration, "weights")
    wdir.mkdir(parents=True, exist_ok=True)
    print("Saving weights to %s%s..." % (wdir.absolute(), "/" + name_suffix if name_suffix else ""))
    path = lambda stem: "%s/%s.pth" % (wdir.absolute(), join_strings("_", [stem, name_suffix]))
    torch.save(_host_copy(G.state_dict()), path("G"))
    torch.save(G.optim.state_dict(), path("G_optim"))
    torch.save(_host_copy(D.state_dict()), path("D"))
    torch.save(D.optim.state_dict(), path("D_optim"))
    torch.save({k: v for k, v in state_dict.items() if k != "config"}, path("state_dict"))
    if G_ema is not None:
        torch.save(_host_copy(G_ema.state_dict()), path("G_ema"))


def load_weights(G, D, state_dict, configuration, weight_name=None, G_ema=None, strict=True, load_optim=True):
    wdir = _run_dir(configuration, "weights")
    print(f"Loading {weight_name + ' ' if weight_name else ''}weights from {wdir.absolute()}...")
    path = lambda stem: wdir.joinpath(f"{join_strings('_', [stem, weight_name])}.pth").absolute()
    # always to the host first: the tensors are copied into the (already placed) flat arenas, and a data-parallel
    # rank must not create a context on GPU 0 by unpickling another rank's CUDA tensors
    read = lambda stem: torch.load(path(stem), map_location="cpu")

    def load_net(net, stem, old, new):
        sd = read(stem)
        try:
            net.load_state_dict(sd, strict=strict)
        except RuntimeError:
            print("Mismatch between file weight keys and model keys. Try renaming.")
            net.load_state_dict(rename_weight_keys(sd, old, new), strict=strict)

    if G is not None:
        load_net(G, "G", "transG", "RR_G")
        if load_optim:
            G.optim.load_state_dict(read("G_optim"))
    if D is not None:
        load_net(D, "D", "transcoder", "RR_D")
        if load_optim:
            D.optim.load_state_dict(read("D_optim"))
    saved = read("state_dict")
    for item in state_dict:
        if item in saved:
            state_dict[item] = saved[item]
    if G_ema is not None:
        load_net(G_ema, "G_ema", "transG", "RR_G")


def write_metadata(configuration, state_dict):
    import datetime
    ldir = _run_dir(configuration, "logs")
    ldir.mkdir(parents=True, exist_ok=True)
    with open(ldir.joinpath("metalog.txt").absolute(), "w") as f:
        f.write("datetime: %s\n" % str(datetime.datetime.now()))
        f.write("state: %s\n" % str(state_dict))


def get_singular_values(module, prefix):
    """{'<prefix>_<state-dict key with dots as underscores>': sigma} for every ``sv`` buffer (reference key format, e.g.
    ``G_linear_sv0``), read back with ONE device-to-host copy -- the reference does one ``.item()`` per layer, ~211 host
    synchronisations every sv_log_interval iterations."""
    names, vals = [], []
    for k, v in module.state_dict().items():
        if "sv" in k:
            names.append(k)
            vals.append(v.reshape(-1)[:1])
    if not names:
        return {}
    host = torch.cat(vals).float().cpu().tolist()
    return {f"{prefix}_{n}".replace(".", "_"): float(x) for n, x in zip(names, host)}


def denorm(x, crop=True):
    """[-1, 1] network range -> detector units [0, 255] (+ the 3-row crop): reference utils/norm.py:34-46."""
    x = torch.pow(256.0, x.float() * 0.5 + 0.5) - 1.0
    x = x.clamp(0, 255)
    return x[..., 3:-3, :] if crop else x


def save_and_sample(G, D, G_ema, z_, y_, fixed_z, fixed_y, state_dict, config):
    """Checkpoint copy + fixed-latent sample of the current generator (reference utils/__init__.py:299-365).  The
    sample is written as ``samples/fixed_samples<itr>.npy`` (detector units, [N, 1, H-6, W]); the reference's JPEG
    sheet additionally needs torchvision and is written only when that is importable."""
    save_weights(G, D, state_dict, config, "copy%d" % state_dict["itr"], G_ema if config["ema"] else None)
    if config["num_save_copies"] > 0:
        state_dict["save_num"] = (state_dict["save_num"] + 1) % config["num_save_copies"]
    which_G = G_ema if (config["ema"] and config["use_ema"]) else G
    with torch.no_grad():
        imgs = denorm(which_G(fixed_z, fixed_y).float()).cpu()
    sdir = _run_dir(config, "samples")
    sdir.mkdir(parents=True, exist_ok=True)
    np.save(str(sdir.joinpath(f"fixed_samples{state_dict['itr']}.npy").absolute()), imgs.numpy())
    try:
        from torchvision.utils import save_image
    except Exception:
        return
    save_image(imgs, sdir.joinpath(f"fixed_samples{state_dict['itr']}.jpg").absolute(), nrow=int(imgs.shape[0] ** 0.5),
               normalize=False)
