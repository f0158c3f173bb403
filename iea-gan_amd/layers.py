"""Operator layer of the MI355X IEA-GAN path, behind the reference's ``layers`` module surface.

Same class names, constructor signatures and state-dict keys as the reference's operator-injection
API (reference ``layers.py``: SN 121-165, SNConv2d 169-206, SNLinear 210-224, SNEmbedding 230-259,
Attention 262-300, ccbn 622-694, bn 698-742), so ``functools.partial(layers.SNConv2d, ...)`` factories
keep working.  ``forward`` of each class is the stand-alone (module-boundary, NCHW fp32) entry; the
networks in ``model.py`` call the fused internal entry points with bf16 NHWC activations.

All arithmetic runs in libieagan_hip.so (see ``ops.py``); there is no eager fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter as P

import _hip as H
import ops
from arena import Arena, arena_of


class identity(nn.Module):
    def forward(self, tensor):
        return tensor


# -----------------------------------------------------------------------------------------------------
# spectral norm
# -----------------------------------------------------------------------------------------------------
class SN(object):
    """Mixin holding the power-iteration state (``u0``, ``sv0``) of a spectrally normalised layer.
    One singular value / one iteration per forward (what every shipped config uses)."""

    def __init__(self, num_svs, num_itrs, num_outputs, transpose=False, eps=1e-12):
        if num_svs != 1 or num_itrs != 1 or transpose:
            raise NotImplementedError("the MI355X path implements num_svs=1, num_itrs=1, transpose=False")
        self.num_itrs, self.num_svs, self.transpose, self.eps = num_itrs, num_svs, transpose, eps
        self.register_buffer("u0", torch.randn(1, num_outputs))
        self.register_buffer("sv0", torch.ones(1))
        self._bank = None

    @property
    def u(self):
        return [self.u0]

    @property
    def sv(self):
        return [self.sv0]

    _sn_kind = ops.KIND_PLAIN

    def _record(self):
        """Run this layer's own power iteration (stand-alone use) and return its SNRecord."""
        H.require_gpu()
        ar = arena_of(self)
        if self._bank is None or self._bank.arena is not ar.flat:
            self._bank = ops.SNBank(ar.flat, [("l", self._sn_kind, self.weight, self.u0, self.sv0)])
        return self._bank.run(self.training, self.eps)["l"]

    def W_(self):
        """The spectrally normalised weight W / sigma (differentiable w.r.t. ``weight``)."""
        rec = self._record()
        if self._sn_kind == ops.KIND_PLAIN:
            return ops.SNWeightFn.apply(self.weight, rec)
        sigma = rec.ctx[0]
        return self.weight / sigma.detach() if not self.weight.requires_grad else _DivSigma.apply(self.weight, rec)


class _DivSigma(torch.autograd.Function):
    """weight / sigma in the parameter's own layout, for API users of ``W_()`` on conv layers."""

    @staticmethod
    def forward(ctx, weight, rec):
        ctx.rec = rec
        ctx.save_for_backward(weight)
        return weight / rec.ctx[0]

    @staticmethod
    def backward(ctx, g):
        (weight,) = ctx.saved_tensors
        rec = ctx.rec
        plain = ops.SNRecord()
        for k in ops.SNRecord.__slots__:
            setattr(plain, k, getattr(rec, k))
        plain.kind = ops.KIND_PLAIN
        return ops.sn_backward(g.contiguous().view(rec.out, rec.inn), weight, plain)[0], None


class SNConv2d(nn.Conv2d, SN):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 num_svs=1, num_itrs=1, eps=1e-12):
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        SN.__init__(self, num_svs, num_itrs, out_channels, eps=eps)
        k = self.kernel_size
        if k not in ((1, 1), (3, 3)) or self.stride != (1, 1) or self.dilation != (1, 1) or self.groups != 1 \
                or self.padding != ((k[0] - 1) // 2, (k[1] - 1) // 2):
            raise NotImplementedError("MI355X SNConv2d: 1x1 (pad 0) or 3x3 (pad 1), stride 1, no dilation/groups")
        self.taps = k[0] * k[1]
        if in_channels == 1 and self.taps == 9:
            self._sn_kind = ops.KIND_C1_IN
        elif out_channels == 1 and self.taps == 9:
            self._sn_kind = ops.KIND_C1_OUT
        else:
            self._sn_kind = ops.KIND_CONV

    conv_flags = 0          # ieagan_conv_desc.flags of this layer; set by the owning network (conv_dtype='fp8' -> H.CONV_FP8)

    # fused internal entry (bf16 NHWC in/out); kwargs are those of ops.conv
    def fused(self, xa, rec, **kw):
        return ops.conv(xa, self.weight, self.bias, rec, self.taps, flags=self.conv_flags, **kw)

    def forward(self, x):
        rec = self._record()
        if self._sn_kind == ops.KIND_C1_IN:
            return ops.ToNCHWFn.apply(ops.InputConvFn.apply(x, self.weight, self.bias, rec))
        if self._sn_kind == ops.KIND_C1_OUT:
            raise NotImplementedError("C->1 SNConv2d is only available fused (Generator.output_layer)")
        xa, _ = ops.ToNHWCFn.apply(x, False)
        out, _ = self.fused(xa, rec)
        return ops.ToNCHWFn.apply(out)


class SNLinear(nn.Linear, SN):
    def __init__(self, in_features, out_features, bias=True, num_svs=1, num_itrs=1, eps=1e-12):
        nn.Linear.__init__(self, in_features, out_features, bias)
        SN.__init__(self, num_svs, num_itrs, out_features, eps=eps)

    def fused(self, x, rec):
        if x.is_cuda and x.dim() == 2 and self.in_features % 4 == 0 and self.in_features <= 2048:
            return ops.LinearFn.apply(x, self.weight, self.bias, rec)       # one HIP launch each way (csrc/rrm_fused.hip)
        return F.linear(x, ops.SNWeightFn.apply(self.weight, rec), self.bias)

    def forward(self, x):
        return self.fused(x, self._record())


class SNEmbedding(nn.Embedding, SN):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2,
                 scale_grad_by_freq=False, sparse=False, _weight=None, num_svs=1, num_itrs=1, eps=1e-12):
        nn.Embedding.__init__(self, num_embeddings, embedding_dim, padding_idx, max_norm, norm_type, scale_grad_by_freq,
                              sparse, _weight)
        SN.__init__(self, num_svs, num_itrs, num_embeddings, eps=eps)

    def fused(self, idx, rec):
        return F.embedding(idx, ops.SNWeightFn.apply(self.weight, rec))

    def forward(self, x):
        return self.fused(x, self._record())


# -----------------------------------------------------------------------------------------------------
# non-local block (SAGAN style)
# -----------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, ch, which_conv=SNConv2d, name="attention"):
        super().__init__()
        self.ch, self.which_conv = ch, which_conv
        self.theta = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.phi = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.g = which_conv(ch, ch // 2, kernel_size=1, padding=0, bias=False)
        self.o = which_conv(ch // 2, ch, kernel_size=1, padding=0, bias=False)
        self.gamma = P(torch.tensor(0.0), requires_grad=True)

    def sn_layers(self, prefix):
        return [(f"{prefix}.{n}", getattr(self, n)) for n in ("theta", "phi", "g", "o")]

    def fused(self, xa, recs, prefix):
        """xa: bf16 [N,H,W,C].  theta/phi/g/o 1x1 convs and the streaming-softmax affinity are HIP kernels;
        the 2x2 max-pool of phi / g and the final gamma*o + x are two small HIP element-wise kernels."""
        N, Hh, Ww, C = xa.shape
        # xa feeds theta, phi, g and the residual: their four gradients are summed inside the dgrad kernels (ops.SumLink)
        link = ops.SumLink(4) if (torch.is_grad_enabled() and xa.requires_grad and ops.opts_of(recs[prefix + ".theta"]).fuse_shortcut_grad) else None
        theta, _ = self.theta.fused(xa, recs[prefix + ".theta"], res_in=link)
        phi, _ = self.phi.fused(xa, recs[prefix + ".phi"], res_in=link)
        g, _ = self.g.fused(xa, recs[prefix + ".g"], res_in=link)

        phi, g = ops.MaxPool2Fn.apply(phi), ops.MaxPool2Fn.apply(g)
        q = theta.view(N, Hh * Ww, C // 8)
        k = phi.view(N, Hh * Ww // 4, C // 8)
        v = g.view(N, Hh * Ww // 4, C // 2)
        o_pre = ops.NLAttentionFn.apply(q, k, v).view(N, Hh, Ww, C // 2)
        o, _ = self.o.fused(o_pre, recs[prefix + ".o"])
        return ops.GammaResidualFn.apply(o, xa, self.gamma, link)

    def forward(self, x, y=None):
        H.require_gpu()
        recs = {f"a.{n}": m._record() for n, m in (("theta", self.theta), ("phi", self.phi), ("g", self.g), ("o", self.o))}
        xa, _ = ops.ToNHWCFn.apply(x, False)
        return ops.ToNCHWFn.apply(self.fused(xa, recs, "a"))


# -----------------------------------------------------------------------------------------------------
# normalisation
# -----------------------------------------------------------------------------------------------------
class ccbn(nn.Module):
    """Class-conditional BatchNorm: batch_norm without affine, then ``out*(1+gain(y)) + bias(y)``."""

    def __init__(self, output_size, input_size, which_linear, eps=1e-5, momentum=0.1, cross_replica=False, mybn=False,
                 norm_style="bn"):
        super().__init__()
        if mybn or norm_style != "bn":
            raise NotImplementedError("MI355X ccbn: norm_style='bn', mybn=False (the shipped configuration)")
        self.output_size, self.input_size = output_size, input_size
        self.gain = which_linear(input_size, output_size)
        self.bias = which_linear(input_size, output_size)
        self.eps, self.momentum, self.cross_replica, self.mybn, self.norm_style = eps, momentum, cross_replica, mybn, norm_style
        self.register_buffer("stored_mean", torch.zeros(output_size))
        self.register_buffer("stored_var", torch.ones(output_size))

    def scale_shift(self, stats, bank, col_gain, col_bias, count, events=1):
        """Per-(n,c) scale / shift from the producer's per-event statistics and the generator-wide gain bank
        (``count``: elements per channel of one event)."""
        link = ops.BNLink() if torch.is_grad_enabled() else None
        s, t = ops.BNFinalizeFn.apply(stats, bank.gb, bank, col_gain, col_bias, self.output_size, self.stored_mean,
                                      self.stored_var, count, self.eps, 0.1, self.training, events, link)
        if link is not None:
            s._bn_link = link          # read by the consuming conv's backward (ops.ConvFn): dgrad-fused BatchNorm backward
        return s, t

    def forward(self, x, y):
        """Stand-alone ccbn (NCHW fp32 in/out): statistics + apply as two HIP passes."""
        H.require_gpu()
        xa, st = ops.ToNHWCFn.apply(x, self.training)
        gb = torch.cat([self.gain(y), self.bias(y)], 1)
        bank = ops.GainBank(gb, 1)
        C = self.output_size
        N, Hh, Ww, _ = xa.shape
        scale, shift = self.scale_shift(st, bank, 0, C, N * Hh * Ww)
        return ops.ToNCHWFn.apply(_affine_act(xa, scale, shift, relu=False))

    def extra_repr(self):
        return f"out: {self.output_size}, in: {self.input_size}, cross_replica={self.cross_replica}"


class bn(nn.Module):
    def __init__(self, output_size, eps=1e-5, momentum=0.1, cross_replica=False, mybn=False):
        super().__init__()
        if mybn:
            raise NotImplementedError("MI355X bn: mybn=False")
        self.output_size = output_size
        self.gain = P(torch.ones(output_size), requires_grad=True)
        self.bias = P(torch.zeros(output_size), requires_grad=True)
        self.eps, self.momentum, self.cross_replica, self.mybn = eps, momentum, cross_replica, mybn
        self.register_buffer("stored_mean", torch.zeros(output_size))
        self.register_buffer("stored_var", torch.ones(output_size))

    def scale_shift(self, stats, count, events=1, n_images=1):
        return ops.BNFinalizePlainFn.apply(stats, self.gain, self.bias, self.stored_mean, self.stored_var, count, self.eps,
                                           self.momentum, self.training, events, n_images)

    def forward(self, x, y=None):
        H.require_gpu()
        xa, st = ops.ToNHWCFn.apply(x, self.training)
        N, Hh, Ww, _ = xa.shape
        scale, shift = self.scale_shift(st, N * Hh * Ww)
        return ops.ToNCHWFn.apply(_affine_act(xa, scale, shift, relu=False))


class _IdentityConvCache:
    """bf16 identity 1x1 'weights' used to run a bare BN apply through the fused conv kernel."""
    cache = {}

    @classmethod
    def get(cls, C, device):
        key = (C, str(device))
        if key not in cls.cache:
            rec = ops.SNRecord()
            rec.kind, rec.out, rec.inn, rec.taps, rec.cin = ops.KIND_CONV, C, C, 1, C
            rec.kpad = rec.kpad2 = ops._kpad(C)
            w = torch.zeros(C, rec.kpad, dtype=torch.bfloat16, device=device)
            w[:, :C] = torch.eye(C, dtype=torch.bfloat16, device=device)
            rec.w_fwd = rec.w_bwd = w
            rec.w_plain = rec.ctx = None
            cls.cache[key] = (rec, torch.empty(0, device=device))
        return cls.cache[key]


def _affine_act(xa, scale, shift, relu):
    """Stand-alone BN apply: the conv kernel's prologue with an identity 1x1 weight (frozen)."""
    rec, dummy_w = _IdentityConvCache.get(xa.shape[-1], xa.device)
    out, _ = ops.conv(xa, dummy_w, None, rec, 1, scale=scale, shift=shift, relu=relu)
    return out
