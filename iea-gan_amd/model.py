"""IEA-GAN generator / discriminator for MI355X, behind the reference's ``model`` module surface.

Public surface kept (reference ``model.py``): ``GBlock`` 16-71, ``G_arch`` 74-136, ``Generator`` 139-487,
``DBlock`` 490-557, ``D_arch`` 561-621, ``Discriminator`` 624-944, ``G_D`` 949-1121, ``Model`` 1124-1127,
``generate`` 1130-1148 -- constructor keywords, attribute names, forward signatures / return arities and
state-dict keys.  Inside, a forward pass is a short chain of fused HIP operators on bf16 NHWC activations
(``ops.py``): one batched spectral-norm launch for every layer of the network, then per block four fused
convolutions whose prologue applies the previous BatchNorm + ReLU (+ upsample / pool) and whose epilogue
adds bias + shortcut and accumulates the statistics of the next BatchNorm.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

import _hip as H
import layers
import ops
import RRM
from arena import Arena, arena_of
from diff_aug import DiffAugment
from optim import FusedAdam


# =====================================================================================================
# architecture tables (channel multipliers of ch per stage; every G stage up-samples, every listed
# D stage down-samples)
# =====================================================================================================
_G_TABLE = {512: ([16, 16, 8, 8, 4, 2, 1], [16, 8, 8, 4, 2, 1, 1]), 256: ([16, 16, 8, 8, 4, 2], [16, 8, 8, 4, 2, 1]),
            128: ([16, 16, 8, 4, 2], [16, 8, 4, 2, 1]), 64: ([16, 16, 8, 4], [16, 8, 4, 2]), 32: ([4, 4, 4], [4, 4, 4])}
_D_TABLE = {512: ([1, 1, 2, 4, 8, 8, 16], [1, 2, 4, 8, 8, 16, 16]), 256: ([1, 2, 4, 8, 8, 16], [2, 4, 8, 8, 16, 16]),
            128: ([1, 2, 4, 8, 16], [2, 4, 8, 16, 16]), 64: ([1, 2, 4, 8], [2, 4, 8, 16]), 32: ([4, 4, 4], [4, 4, 4])}


def _attn_set(attention):
    return {int(a) for a in str(attention).split("_")}


def G_arch(ch=64, attention="64", ksize="333333", dilation="111111"):
    arch = {}
    for res, (ins, outs) in _G_TABLE.items():
        n = len(ins)
        resolutions = [res >> (n - 1 - i) for i in range(n)]
        arch[res] = {"in_channels": [ch * m for m in ins], "out_channels": [ch * m for m in outs], "upsample": [True] * n,
                     "resolution": resolutions, "attention": {r: (r in _attn_set(attention)) for r in resolutions}}
    return arch


def D_arch(ch=64, attention="64", ksize="333333", dilation="111111"):
    arch = {}
    for res, (ins, outs) in _D_TABLE.items():
        n = len(ins)
        resolutions = [max(res >> (i + 1), 4) for i in range(n + 1)]
        down = [True] * n + [False] if res != 32 else [True, True, False, False]
        if res == 32:
            resolutions = [16, 16, 16, 16]
        arch[res] = {"in_channels": [ch * m for m in ins], "out_channels": [ch * m for m in outs], "downsample": down,
                     "resolution": resolutions, "attention": {r: (r in _attn_set(attention)) for r in set(resolutions)}}
    return arch


def _activation(name):
    if name in ("inplace_relu", "relu"):
        return nn.ReLU(inplace=(name == "inplace_relu"))
    raise NotImplementedError(f"activation function {name} not implemented on the MI355X path (ReLU is fused into the convs)")


def _set_conv_dtype(net, conv_dtype):
    """``conv_dtype`` of a network ('bf16' | 'fp8', BASELINE configs[4]): the operand precision of the MFMAs of its C = 64 / 128 3x3
    layers, carried per layer in the conv descriptor flags (no process-wide switch: two networks / train functions with different
    settings coexist)."""
    if conv_dtype not in ("bf16", "fp8"):
        raise NotImplementedError("conv_dtype must be 'bf16' or 'fp8'")
    net.conv_dtype = conv_dtype
    for m in net.modules():
        if isinstance(m, layers.SNConv2d):
            m.conv_flags = H.CONV_FP8 if conv_dtype == "fp8" else 0


def _n_events(net, n_images):
    """Number of independent events in a batch of ``n_images``: ``events_per_step`` when the batch is exactly that many
    events of ``event_size`` images, else 1 (the whole batch is one event, as in the reference)."""
    E = net.events_per_step
    return E if (E > 1 and n_images == E * net.event_size) else 1


def _sn_children(module, prefix):
    """(state-dict prefix, layer) for every spectrally normalised layer below ``module``."""
    out = []
    for n, m in module.named_modules():
        if isinstance(m, layers.SN):
            out.append((f"{prefix}.{n}" if prefix and n else (prefix or n), m))
    return out


def _sn_biases(module):
    """name -> bias parameter of every spectrally normalised conv below ``module`` (batched SN backward)."""
    return {n: m.bias for n, m in _sn_children(module, "") if isinstance(m, layers.SNConv2d) and getattr(m, "bias", None) is not None}


# =====================================================================================================
# generator
# =====================================================================================================
class GBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=layers.SNConv2d, which_bn=layers.bn, activation=None,
                 upsample=None, channel_ratio=4):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.hidden_channels = in_channels // channel_ratio
        self.which_conv, self.which_bn, self.activation, self.upsample = which_conv, which_bn, activation, upsample
        hid = self.hidden_channels
        self.conv1 = which_conv(in_channels, hid, kernel_size=1, padding=0)
        self.conv2 = which_conv(hid, hid)
        self.conv3 = which_conv(hid, hid)
        self.conv4 = which_conv(hid, out_channels, kernel_size=1, padding=0)
        self.bn1, self.bn2, self.bn3, self.bn4 = which_bn(in_channels), which_bn(hid), which_bn(hid), which_bn(hid)

    def fused(self, xa, xstats, bank, cols, recs, prefix, want_stats=True, events=1):
        """xa bf16 [N,H,W,Cin] with its per-event (sum, sumsq) statistics -> (out bf16, out statistics).  ``events``: the
        batch holds that many events of N / events images; every BatchNorm normalises within its event."""
        N, Hh, Ww, _ = xa.shape
        up = 1 if self.upsample else 0
        cnt = (N // events) * Hh * Ww
        tr = self.training
        E = events
        link = ops.ResLink() if (torch.is_grad_enabled() and ops.opts_of(recs[prefix + ".conv1"]).fuse_shortcut_grad) else None   # conv4 -> conv1, in-kernel
        s, t = self.bn1.scale_shift(xstats, bank, *cols["bn1"], cnt, E)
        h, st = self.conv1.fused(xa, recs[prefix + ".conv1"], scale=s, shift=t, relu=True, want_stats=tr, res_in=link, events=E)
        s, t = self.bn2.scale_shift(st, bank, *cols["bn2"], cnt, E)
        h, st = self.conv2.fused(h, recs[prefix + ".conv2"], scale=s, shift=t, relu=True, rs=up, want_stats=tr, events=E)
        cnt2 = cnt * (4 if up else 1)
        s, t = self.bn3.scale_shift(st, bank, *cols["bn3"], cnt2, E)
        h, st = self.conv3.fused(h, recs[prefix + ".conv3"], scale=s, shift=t, relu=True, want_stats=tr, events=E)
        s, t = self.bn4.scale_shift(st, bank, *cols["bn4"], cnt2, E)
        return self.conv4.fused(h, recs[prefix + ".conv4"], scale=s, shift=t, relu=True, ra=xa, Ca=self.out_channels,
                                ra_rs=up, want_stats=want_stats and tr, res_out=link, events=E)

    def forward(self, x, y):
        """Stand-alone block on NCHW fp32 ``x`` with the conditioning vector ``y`` [N, cond]."""
        H.require_gpu()
        arena_of(self)
        recs = {"b." + n: m._record() for n, m in _sn_children(self, "") if isinstance(m, layers.SNConv2d)}
        parts, cols, c0 = [], {}, 0
        for name in ("bn1", "bn2", "bn3", "bn4"):
            b = getattr(self, name)
            parts += [b.gain(y), b.bias(y)]
            cols[name] = (c0, c0 + b.output_size)
            c0 += 2 * b.output_size
        bank = ops.GainBank(torch.cat(parts, 1), 4)
        xa, st = ops.ToNHWCFn.apply(x, self.training)
        out, _ = self.fused(xa, st, bank, cols, recs, "b", want_stats=False)
        return ops.ToNCHWFn.apply(out)


class Generator(nn.Module):
    def __init__(self, G_ch=64, G_depth=2, dim_z=128, bottom_width=4, resolution=256, G_kernel_size=3, G_attn="64",
                 n_classes=40, H_base=1, num_G_SVs=1, num_G_SV_itrs=1, attn_type="sa", G_shared=True, shared_dim=128,
                 rdof_dim=4, hier=True, cross_replica=False, mybn=False, G_activation="relu", G_lr=5e-5, G_B1=0.0,
                 G_B2=0.999, adam_eps=1e-8, BN_eps=1e-5, SN_eps=1e-12, G_init="ortho", G_mixed_precision=False,
                 G_fp16=False, skip_init=False, no_optim=False, sched_version="default", RRM_prx_G=True,
                 prior_embed=False, n_head_G=2, G_param="SN", norm_style="bn", device="cuda", events_per_step=1, **kwargs):
        super().__init__()
        # E events per forward batch: a batch of exactly E * event_size images is E independent events
        self.events_per_step = max(int(events_per_step or 1), 1)
        self.event_size = int(kwargs.get("batch_size") or n_classes)
        if G_param != "SN" or not G_shared or not hier or not RRM_prx_G or prior_embed:
            raise NotImplementedError("MI355X Generator: G_param='SN', G_shared, hier, RRM_prx_G, no prior_embed "
                                      "(the configuration the reference ships)")
        self.ch, self.G_depth, self.dim_z, self.bottom_width, self.H_base = G_ch, G_depth, dim_z, bottom_width, H_base
        self.resolution, self.kernel_size, self.attention, self.n_classes = resolution, G_kernel_size, G_attn, n_classes
        self.G_shared, self.shared_dim, self.hier = G_shared, (shared_dim if shared_dim > 0 else dim_z), hier
        self.cross_replica, self.mybn, self.init, self.G_param, self.norm_style = cross_replica, mybn, G_init, G_param, norm_style
        self.BN_eps, self.SN_eps, self.fp16 = BN_eps, SN_eps, G_fp16
        self.RRM_prx_G, self.n_head_G, self.prior_embed, self.rdof_dim, self.device = RRM_prx_G, n_head_G, prior_embed, rdof_dim, device
        self.activation = _activation(G_activation)
        self.arch = G_arch(self.ch, self.attention)[resolution]
        if any(self.arch["attention"].values()):
            raise NotImplementedError("self-attention inside G (G_attn) is not part of the MI355X path (reference ships G_attn='0')")

        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, num_svs=num_G_SVs,
                                            num_itrs=num_G_SV_itrs, eps=SN_eps)
        self.which_linear = functools.partial(layers.SNLinear, num_svs=num_G_SVs, num_itrs=num_G_SV_itrs, eps=SN_eps)
        self.which_embedding = nn.Embedding
        cond = self.shared_dim + self.dim_z
        self.which_bn = functools.partial(layers.ccbn, which_linear=functools.partial(self.which_linear, bias=False),
                                          cross_replica=cross_replica, mybn=mybn, input_size=cond, norm_style=norm_style,
                                          eps=BN_eps)
        self.shared = nn.Embedding(n_classes, self.shared_dim)
        self.linear_f = self.which_linear(self.shared_dim + rdof_dim, 128)
        self.RR_G = RRM.RelationalReasoning(num_layers=1, input_dim=128, dim_feedforward=128, which_linear=nn.Linear,
                                            num_heads=n_head_G, dropout=0.0, hidden_dim=128)
        self.linear = self.which_linear(cond, self.arch["in_channels"][0] * (bottom_width ** 2) * H_base)
        blocks = []
        for idx, (cin, cout) in enumerate(zip(self.arch["in_channels"], self.arch["out_channels"])):
            for g in range(G_depth):
                last = g == G_depth - 1
                blocks.append(nn.ModuleList([GBlock(cin, cout if last else cin, self.which_conv, self.which_bn, self.activation,
                                                    functools.partial(F.interpolate, scale_factor=2)
                                                    if (self.arch["upsample"][idx] and last) else None)]))
        self.blocks = nn.ModuleList(blocks)
        c_last = self.arch["out_channels"][-1]
        self.output_layer = nn.Sequential(layers.bn(c_last, cross_replica=cross_replica, mybn=mybn), self.activation,
                                          self.which_conv(c_last, 1))
        _set_conv_dtype(self, kwargs.get("conv_dtype", "bf16"))
        if not skip_init:
            self.init_weights()
        self._plan = None
        if no_optim:
            return
        self.lr, self.B1, self.B2, self.adam_eps = G_lr, G_B1, G_B2, adam_eps
        self.optim = FusedAdam(self.parameters(), lr=G_lr, betas=(G_B1, G_B2), weight_decay=0, eps=adam_eps, owner=self)
        self.lr_sched = _scheduler(self.optim, sched_version, G_lr, kwargs)

    def init_weights(self):
        self.param_count = _init_weights(self, self.init)
        print("Param count for Gs initialized parameters: %d" % self.param_count)

    def _apply(self, fn, *a, **k):
        self._plan = None
        return super()._apply(fn, *a, **k)

    # ---- one-time plan: arena, SN table, gain-bank columns -----------------------------------------
    def _prepare(self):
        H.require_gpu()
        ar = self.__dict__.get("_arena")
        if ar is None or ar.root is not self or not ar.contains(self.shared.weight):
            ar, self._plan = Arena(self), None
        if self._plan is not None and self._plan["arena"] is ar:
            return self._plan
        entries, stack, cols, c0 = [], [], [], 0
        for name, m in _sn_children(self, ""):
            entries.append((name, m._sn_kind, m.weight, m.u0, m.sv0))
        for bi, bl in enumerate(self.blocks):
            cmap = {}
            for bn_name in ("bn1", "bn2", "bn3", "bn4"):
                b = getattr(bl[0], bn_name)
                stack += [f"blocks.{bi}.0.{bn_name}.gain", f"blocks.{bi}.0.{bn_name}.bias"]
                cmap[bn_name] = (c0, c0 + b.output_size)
                c0 += 2 * b.output_size
            cols.append(cmap)
        bank = ops.SNBank(ar.flat, entries, stack=stack, owner=ar, biases=_sn_biases(self))
        modmap = dict(_sn_children(self, ""))
        sw = [modmap[n].weight for n in stack]
        offs = {id(p): o for p, o, _ in ar.param_slices}
        dev = ar.flat.device
        rows, r0 = [], 0
        for w in sw:
            rows.append(r0)
            r0 += w.shape[0]
        self._plan = dict(arena=ar, bank=bank, stack=stack, cols=cols, stack_weights=sw, n_bn=4 * len(self.blocks),
                          stack_layers=torch.tensor([bank.index[n] for n in stack], dtype=torch.int32, device=dev),
                          stack_row0=torch.tensor(rows, dtype=torch.int64, device=dev),
                          stack_dst_arena=torch.tensor([offs[id(w)] for w in sw], dtype=torch.int64, device=dev),
                          stack_dst_flat=torch.tensor([r * sw[0].shape[1] for r in rows], dtype=torch.int64, device=dev))
        return self._plan

    def forward(self, z, y, rdof=None, export=False):
        """``export=True`` (inference): returns detector units [N, H-6, W] straight from the last kernel (see ``generate``)."""
        plan = self._prepare()
        recs = plan["bank"].run(self.training, self.SN_eps)
        N = y.size(0)
        E = _n_events(self, N)
        ye = self.shared(y)
        if rdof is None:
            rdof = self.__dict__.pop("_next_rdof", None)    # explicit draw injected by a parity test
        if rdof is None:   # the reference draws a fixed 40 rows here (model.py:466); we follow the batch
            rdof = torch.randn(N, self.rdof_dim, device=z.device)
        ye = self.linear_f.fused(torch.cat([ye, rdof], 1), recs["linear_f"])
        ye = self.RR_G(ye.view(E, N // E, -1)).reshape(N, -1)          # the sensors of ONE event are the RRM's tokens
        zc = torch.cat([ye, z], 1)
        gb = ops.StackedSNLinearFn.apply(zc, recs["__stack__"], recs, plan, *plan["stack_weights"])
        bank = ops.GainBank(gb, plan["n_bn"])
        h = self.linear.fused(zc, recs["linear"])
        h = h.view(N, -1, self.bottom_width, self.bottom_width * self.H_base)
        xa, st = ops.ToNHWCFn.apply(h, self.training, E)
        for bi, bl in enumerate(self.blocks):
            xa, st = bl[0].fused(xa, st, bank, plan["cols"][bi], recs, f"blocks.{bi}.0", events=E)
        bn_out, conv_out = self.output_layer[0], self.output_layer[2]
        Nn, Hh, Ww, _ = xa.shape
        s, t = bn_out.scale_shift(st, (Nn // E) * Hh * Ww, E, Nn)
        if export:
            if torch.is_grad_enabled() and any(p.requires_grad for p in (conv_out.weight,)) and xa.requires_grad:
                raise RuntimeError("Generator(export=True) is an inference path: call it under torch.no_grad()")
            return ops.output_conv_export(xa, s, t, recs["output_layer.2"], conv_out.bias)
        return ops.OutputConvFn.apply(xa, s, t, conv_out.weight, conv_out.bias, recs["output_layer.2"])


# =====================================================================================================
# discriminator
# =====================================================================================================
class DBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=layers.SNConv2d, wide=True, preactivation=True, activation=None,
                 downsample=None, channel_ratio=4):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.hidden_channels = out_channels // channel_ratio
        self.which_conv, self.preactivation, self.activation, self.downsample = which_conv, preactivation, activation, downsample
        hid = self.hidden_channels
        self.conv1 = which_conv(in_channels, hid, kernel_size=1, padding=0)
        self.conv2 = which_conv(hid, hid)
        self.conv3 = which_conv(hid, hid)
        self.conv4 = which_conv(hid, out_channels, kernel_size=1, padding=0)
        self.learnable_sc = in_channels != out_channels
        if self.learnable_sc:
            self.conv_sc = which_conv(in_channels, out_channels - in_channels, kernel_size=1, padding=0)

    def stem_ok(self, input_conv, x):
        """Can ``ops.DStemFn`` take input_conv + this block's conv1 / conv_sc / pooled shortcut?  (the first block of the shipped
        ch = 32 discriminator on a map whose height / width are multiples of 8 / 64: the stem's backward runs conv_sc through the fused
        1x1 backward at the POOLED resolution, whose tiles need (W / 2) % 32 == 0)"""
        return (x.is_cuda and x.dim() == 4 and x.shape[1] == 1 and x.shape[2] % 8 == 0 and x.shape[3] % 64 == 0
                and input_conv.out_channels == 32 and self.in_channels == 32 and self.out_channels == 64 and self.hidden_channels == 16
                and self.downsample is not None and not self.preactivation and self.learnable_sc
                and all(c.bias is not None for c in (input_conv, self.conv1, self.conv_sc)))

    def fused_stem(self, x, input_conv, recs, prefix):
        """The block with its input side fused into one launch (ops.DStemFn): x is the fp32 image [N, 1, H, W]."""
        link = ops.ResLink() if (torch.is_grad_enabled() and ops.opts_of(recs[prefix + ".conv1"]).fuse_shortcut_grad) else None    # conv4 -> stem backward
        h, p0, sc = ops.DStemFn.apply(x, input_conv.weight, input_conv.bias, self.conv1.weight, self.conv1.bias, self.conv_sc.weight,
                                      self.conv_sc.bias, recs["input_conv"], recs[prefix + ".conv1"], recs[prefix + ".conv_sc"], link)
        h, _ = self.conv2.fused(h, recs[prefix + ".conv2"], relu=True)
        h, _ = self.conv3.fused(h, recs[prefix + ".conv3"], relu=True)
        out, _ = self.conv4.fused(h, recs[prefix + ".conv4"], relu=True, rs=2, ra=p0, Ca=self.in_channels, ra_rs=0, rb=sc, res_out=link)
        return out

    def fused(self, xa, recs, prefix):
        rs = 2 if self.downsample else 0
        link = ops.ResLink() if (torch.is_grad_enabled() and ops.opts_of(recs[prefix + ".conv1"]).fuse_shortcut_grad) else None   # conv4 (-> conv_sc) -> conv1
        h, _ = self.conv1.fused(xa, recs[prefix + ".conv1"], relu=self.preactivation, res_in=link)
        h, _ = self.conv2.fused(h, recs[prefix + ".conv2"], relu=True)
        h, _ = self.conv3.fused(h, recs[prefix + ".conv3"], relu=True)
        sc = None
        if self.learnable_sc:   # conv_sc sees the pooled, un-activated block input (model.py:534-539)
            sc, _ = self.conv_sc.fused(xa, recs[prefix + ".conv_sc"], rs=rs, res_in=link if rs == 2 else None)
        out, _ = self.conv4.fused(h, recs[prefix + ".conv4"], relu=True, rs=rs, ra=xa, Ca=self.in_channels, ra_rs=rs, rb=sc,
                                  res_out=link)
        return out

    def forward(self, x):
        H.require_gpu()
        arena_of(self)
        recs = {"b." + n: m._record() for n, m in _sn_children(self, "")}
        xa, _ = ops.ToNHWCFn.apply(x, False)
        return ops.ToNCHWFn.apply(self.fused(xa, recs, "b"))


class Discriminator(nn.Module):
    def __init__(self, D_ch=64, D_wide=True, D_depth=2, resolution=256, D_kernel_size=3, D_attn="64", n_classes=40,
                 attn_type="sa", num_D_SVs=1, num_D_SV_itrs=1, D_activation="relu", conditional_strategy="Proj", D_lr=2e-4,
                 D_B1=0.0, D_B2=0.999, adam_eps=1e-8, SN_eps=1e-12, output_dim=1, D_init="ortho", D_mixed_precision=False,
                 D_fp16=False, sched_version="default", skip_init=False, D_param="SN", hypersphere_dim=512,
                 nonlinear_embed=False, normalize_embed=True, prior_embed=False, RRM_prx_D=False, RRM_embed=False,
                 n_head_D=4, events_per_step=1, **kwargs):
        super().__init__()
        self.events_per_step = max(int(events_per_step or 1), 1)
        self.event_size = int(kwargs.get("batch_size") or n_classes)
        if D_param != "SN" or prior_embed or RRM_prx_D or nonlinear_embed or attn_type != "sa":
            raise NotImplementedError("MI355X Discriminator: D_param='SN', attn_type='sa', no prior_embed / RRM_prx_D / "
                                      "nonlinear_embed (the configuration the reference ships)")
        self.ch, self.D_wide, self.D_depth, self.resolution = D_ch, D_wide, D_depth, resolution
        self.kernel_size, self.attention, self.n_classes = D_kernel_size, D_attn, n_classes
        self.activation = _activation(D_activation)
        self.init, self.D_param, self.SN_eps, self.fp16 = D_init, D_param, SN_eps, D_fp16
        self.RRM_prx_D, self.RRM_embed, self.prior_embed = RRM_prx_D, RRM_embed, prior_embed
        self.conditional_strategy, self.nonlinear_embed, self.normalize_embed, self.n_head_D = \
            conditional_strategy, nonlinear_embed, normalize_embed, n_head_D
        self.arch = D_arch(self.ch, self.attention)[resolution]
        sn = dict(num_svs=num_D_SVs, num_itrs=num_D_SV_itrs, eps=SN_eps)
        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, **sn)
        self.which_linear = functools.partial(layers.SNLinear, **sn)
        self.which_embedding = functools.partial(layers.SNEmbedding, **sn)

        self.input_conv = self.which_conv(1, self.arch["in_channels"][0])
        blocks = []
        for idx, (cin, cout) in enumerate(zip(self.arch["in_channels"], self.arch["out_channels"])):
            stage = [DBlock(cin if d == 0 else cout, cout, self.which_conv, D_wide, (idx > 0 or d > 0), self.activation,
                            nn.AvgPool2d(2) if (self.arch["downsample"][idx] and d == 0) else None) for d in range(D_depth)]
            if self.arch["attention"][self.arch["resolution"][idx]]:
                print("Adding attention layer in D at resolution %d" % self.arch["resolution"][idx])
                stage.append(layers.Attention(cout, self.which_conv))
            blocks.append(nn.ModuleList(stage))
        self.blocks = nn.ModuleList(blocks)
        c_top = self.arch["out_channels"][-1]
        self.linear0 = self.which_linear(c_top, output_dim)
        if RRM_embed:
            self.RR_D = RRM.RelationalReasoning(num_layers=1, input_dim=c_top, dim_feedforward=512, num_heads=n_head_D,
                                                dropout=0.0, hidden_dim=512, which_linear=self.which_linear)
            self.norm = nn.LayerNorm(hypersphere_dim)
        if conditional_strategy == "Proj":
            self.embed = self.which_embedding(n_classes, c_top)
        elif conditional_strategy == "Contra":
            self.linear1 = self.which_linear(c_top, hypersphere_dim)
            self.embed = self.which_embedding(n_classes, hypersphere_dim)
        else:
            raise NotImplementedError(f"conditional_strategy {conditional_strategy}")
        _set_conv_dtype(self, kwargs.get("conv_dtype", "bf16"))
        if not skip_init:
            self.init_weights()
        self._plan = None
        self.lr, self.B1, self.B2, self.adam_eps = D_lr, D_B1, D_B2, adam_eps
        self.optim = FusedAdam(self.parameters(), lr=D_lr, betas=(D_B1, D_B2), weight_decay=0, eps=adam_eps, owner=self)
        self.lr_sched = _scheduler(self.optim, sched_version, D_lr, kwargs)

    def init_weights(self):
        self.param_count = _init_weights(self, self.init)
        print("Param count for Ds initialized parameters: %d" % self.param_count)

    def _apply(self, fn, *a, **k):
        self._plan = None
        return super()._apply(fn, *a, **k)

    def _prepare(self):
        H.require_gpu()
        ar = self.__dict__.get("_arena")
        if ar is None or ar.root is not self or not ar.contains(self.input_conv.weight):
            ar, self._plan = Arena(self), None
        if self._plan is not None and self._plan["arena"] is ar:
            return self._plan
        # only the layers a forward pass evaluates take part in the power iteration: the projection head never touches RR_D
        # (constructed whenever RRM_embed, reference model.py:788-798, 939-944), whose u0 / sv0 must then stay put
        used = lambda n: self.conditional_strategy == "Contra" or not n.startswith("RR_D.")
        entries = [(n, m._sn_kind, m.weight, m.u0, m.sv0) for n, m in _sn_children(self, "") if used(n)]
        self._plan = dict(arena=ar, bank=ops.SNBank(ar.flat, entries, owner=ar, biases=_sn_biases(self)))
        return self._plan

    def forward(self, x, y=None):
        plan = self._prepare()
        recs = plan["bank"].run(self.training, self.SN_eps)
        first = self.blocks[0][0]
        stem = plan["bank"].opts.fuse_d_stem and isinstance(first, DBlock) and first.stem_ok(self.input_conv, x)
        if stem:            # input_conv + the first block's three reads of its output: one launch (h0 never reaches HBM)
            h = first.fused_stem(x, self.input_conv, recs, "blocks.0.0")
        else:
            h = ops.InputConvFn.apply(x, self.input_conv.weight, self.input_conv.bias, recs["input_conv"])
        for si, stage in enumerate(self.blocks):
            for bi, blk in enumerate(stage):
                if stem and si == 0 and bi == 0:
                    continue
                p = f"blocks.{si}.{bi}"
                h = blk.fused(h, recs, p)
        h = ops.ReluSumPoolFn.apply(h)                                             # global sum pool of relu -> [N, C] fp32
        if self.conditional_strategy == "Contra":
            out = torch.squeeze(self.linear0.fused(h, recs["linear0"]))
            nrm = self.normalize_embed
            # unit-sphere class proxies: embedding lookup + F.normalize in one launch (model.py:916, 933)
            proxy = ops.EmbedNormFn.apply(y, self.embed.weight, recs["embed"]) if nrm else self.embed.fused(y, recs["embed"])
            if self.RRM_embed:
                N = h.shape[0]
                E = _n_events(self, N)
                h = self.RR_D(h.view(E, N // E, -1), recs=recs, prefix="RR_D").reshape(N, -1)
                # LayerNorm + F.normalize of the embedding in one launch (model.py:920-921, 935)
                emb = ops.LayerNormFn.apply(self.linear1.fused(h, recs["linear1"]), self.norm.weight, self.norm.bias, self.norm.eps, nrm)
            else:
                emb = self.linear1.fused(h, recs["linear1"])
                if nrm:
                    emb = F.normalize(emb, dim=1)
            return proxy, emb, out
        out = self.linear0.fused(h, recs["linear0"])
        return out + torch.sum(self.embed.fused(y, recs["embed"]) * h, 1, keepdim=True)


def _init_weights(net, style):
    count = 0
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear, nn.Embedding)):
            if style == "ortho":
                init.orthogonal_(m.weight)
            elif style == "N02":
                init.normal_(m.weight, 0, 0.02)
            elif style in ("glorot", "xavier"):
                init.xavier_uniform_(m.weight)
            else:
                print("Init style not recognized...")
            count += sum(p.data.nelement() for p in m.parameters())
    return count


def _scheduler(opt, version, lr, kwargs):
    if version == "CosAnnealLR":
        return torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=kwargs["num_epochs"], eta_min=lr / 4, last_epoch=-1)
    if version == "CosAnnealWarmRes":
        return torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=lr / 4)
    return None


# =====================================================================================================
# G o D composite
# =====================================================================================================
class G_D(nn.Module):
    """G(z) -> DiffAugment (fake only) -> D, with the reference's branch structure and return arities
    (reference model.py:956-1121)."""

    def __init__(self, G, D):
        super().__init__()
        self.G, self.D = G, D

    def forward(self, z, gy, x=None, dy=None, x_aug=None, contra=True, train_G=False, return_G_z=False, split_D=False,
                diff_aug=True, pixel_reg=False, real_first=False, real_out=None):
        """``real_first`` (split_D, D phase): evaluate D(x) BEFORE G(z) -> D(G(z)) -- data-parallel runs hide G's gradient exchange +
        update behind the real pass this way (the reference order is fake, then real: model.py:987, 1002; swapping it changes which
        spectral-norm iterate each pass sees, a tolerance-level deviation).  ``real_out``: D(x, dy) already evaluated
        by the caller (segmented graph replay)."""
        if real_first and real_out is None and split_D and x is not None and not train_G:
            real_out = self.D(x, dy)
        with torch.set_grad_enabled(train_G):
            G_z = self.G(z, gy)
            if diff_aug:
                G_z = DiffAugment(G_z, policy="color,translation,cutout")
            G_reg = F.threshold(G_z, -0.25, -1) if pixel_reg else None
        if return_G_z and not pixel_reg:
            raise RuntimeError("return_G_z needs pixel_reg=True (G_reg is otherwise undefined, as in the reference)")
        if split_D:
            fake = self.D(G_z, gy)
            if contra:
                if train_G:
                    return (*fake, G_z, G_reg) if return_G_z else fake
                return (*fake, *(real_out if real_out is not None else self.D(x, dy)))
            if x is not None:
                return fake, (real_out if real_out is not None else self.D(x, dy))
            return (fake, G_z, G_reg) if return_G_z else fake
        # joint pass over the concatenated batch
        parts, labels = [G_z], [gy]
        if x is not None:
            parts.append(x)
            labels.append(dy)
            if x_aug is not None:
                parts.append(x_aug)
                labels.append(dy)
        D_in = torch.cat(parts, 0) if len(parts) > 1 else G_z
        D_cl = torch.cat(labels, 0) if (dy is not None and len(parts) > 1) else gy
        sizes = [p.shape[0] for p in parts]
        if contra:
            proxy, emb, out = self.D(D_in, D_cl)
            if x is None:
                return (proxy, emb, out, G_z, G_reg) if return_G_z else (proxy, emb, out)
            outs, embs, proxies = torch.split(out, sizes), torch.split(emb, sizes), torch.split(proxy, sizes)
            res = (proxies[0], embs[0], outs[0], proxies[1], embs[1], outs[1])
            return res + (embs[2], outs[2]) if x_aug is not None else res
        out = self.D(D_in, D_cl)
        if x is None:
            return (out, G_z, G_reg) if return_G_z else out
        return tuple(torch.split(out, sizes))


class Model(Generator):
    def __init__(self, config: dict):
        assert isinstance(config, dict), "Expected configuration dictionary"
        super().__init__(**config)


def generate(model):
    """One event of 40 sensor images in detector units: [40, 250, 768] (reference model.py:1130-1148).  With this
    package's Generator the threshold / 256^x / clamp / crop run inside the last conv kernel and only the finished
    event crosses PCIe; any other callable takes the reference's host-side post-processing."""
    device = next(model.parameters()).device
    with torch.no_grad():
        latents = torch.randn(40, 128, device=device)
        labels = torch.arange(40, dtype=torch.long, device=device)
        if isinstance(model, Generator):
            return model(latents, labels, export=True).cpu()
        imgs = model(latents, labels).detach().cpu()
        imgs = F.threshold(imgs, -0.26, -1)            # cut the noise below 7 ADU
        imgs = imgs.mul_(0.5).add_(0.5)
        imgs = torch.pow(256, imgs).add_(-1).clamp_(0, 255)
        return imgs[:, 0, 3:-3, :]
