"""Relational Reasoning Module: pre-LN transformer encoder over the 40 sensor tokens of one event.

Module surface and state-dict keys follow reference ``RRM.py`` (MultiheadAttention 19-63, EncoderBlock
66-109, RelationalReasoning 112-133).  qkv is packed head-interleaved ([B,S,H,3*hd] then chunk), dropout
is 0 in every shipped configuration.  When ``which_linear`` is a spectrally normalised layer the
normalised weights come from the owning network's batched SN launch (``recs``).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


FUSED = True        # stage-wise fused HIP kernels for the encoder block (False: op by op, the path the > 64-token joint pass takes)


def _lin(layer, x, recs, name):
    if recs is not None and name in recs:
        return layer.fused(x, recs[name])
    return layer(x)


def scaled_dot_product(q, k, v):
    att = F.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size(-1)), dim=-1)
    return torch.matmul(att, v), att


class MultiheadAttention(nn.Module):
    def __init__(self, input_dim, embed_dim, num_heads, which_linear):
        super().__init__()
        assert embed_dim % num_heads == 0, "Embedding dimension must be 0 modulo number of heads."
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.which_linear = which_linear
        self.qkv_proj = which_linear(input_dim, 3 * embed_dim)
        self.o_proj = which_linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.qkv_proj.weight)
        self.qkv_proj.bias.data.fill_(0)
        nn.init.xavier_uniform_(self.o_proj.weight)
        self.o_proj.bias.data.fill_(0)

    def forward(self, x, return_attention=False, recs=None, prefix=""):
        B, S, _ = x.shape
        qkv = _lin(self.qkv_proj, x, recs, prefix + ".qkv_proj")          # [B, S, H * 3*hd], head-interleaved
        if not qkv.is_cuda:
            import _hip
            _hip.require_gpu()
            raise RuntimeError("RRM of the MI355X path takes HIP tensors (no CPU fallback)")
        if S <= 64:
            import ops                                                    # fused HIP core: S x S affinity in LDS
            vals, att = ops.RRMAttentionFn.apply(qkv, self.num_heads)
        else:                                                             # > 64 tokens (joint fake+real pass): library path
            q, k, v = qkv.reshape(B, S, self.num_heads, 3 * self.head_dim).permute(0, 2, 1, 3).chunk(3, dim=-1)
            vals, att = scaled_dot_product(q, k, v)
            vals = vals.permute(0, 2, 1, 3).reshape(B, S, self.embed_dim)
        o = _lin(self.o_proj, vals, recs, prefix + ".o_proj")
        return (o, att) if return_attention else o


class EncoderBlock(nn.Module):
    def __init__(self, input_dim, num_heads, dim_feedforward, dropout, which_linear):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("RRM dropout is 0.0 in the IEA-GAN configurations")
        self.which_linear = which_linear
        self.self_attn = MultiheadAttention(input_dim, input_dim, num_heads, which_linear)
        self.linear_net = nn.Sequential(which_linear(input_dim, dim_feedforward), nn.Dropout(dropout),
                                        nn.ReLU(inplace=True), which_linear(dim_feedforward, input_dim))
        self.norm1 = nn.LayerNorm(input_dim)
        self.norm2 = nn.LayerNorm(input_dim)
        self.dropout = nn.Dropout(dropout)

    def _fusable(self, x):
        """The stage-wise fused HIP block (ops.RRMBlockFn) takes <= 64 tokens per event (attention affinity in LDS) and channel
        counts that are multiples of 16; anything else (e.g. the 80 / 120 tokens of a joint fake + real pass) runs op by op."""
        at = self.self_attn
        dims = (at.qkv_proj.in_features, at.qkv_proj.out_features, self.linear_net[0].out_features)
        return x.is_cuda and x.dim() == 3 and x.shape[1] <= 64 and all(d % 16 == 0 for d in dims) and dims[1] == 3 * dims[0] \
            and at.o_proj.in_features == at.o_proj.out_features == dims[0] and self.linear_net[3].out_features == dims[0]

    def forward(self, x, recs=None, prefix=""):
        if FUSED and self._fusable(x):
            import ops
            at, ln = self.self_attn, self.linear_net
            names = [prefix + ".self_attn.qkv_proj", prefix + ".self_attn.o_proj", prefix + ".linear_net.0", prefix + ".linear_net.3"]
            layers_ = (at.qkv_proj, at.o_proj, ln[0], ln[3])
            sn = hasattr(at.qkv_proj, "_record")                       # spectrally normalised flavour (D): normalised weights of this pass
            if sn:
                rl = [recs[n] if (recs is not None and n in recs) else l._record() for n, l in zip(names, layers_)]
            else:
                rl = None
            return ops.RRMBlockFn.apply(x, at.num_heads, self.norm1.eps, rl, self.norm1.weight, self.norm1.bias, at.qkv_proj.weight,
                                        at.qkv_proj.bias, at.o_proj.weight, at.o_proj.bias, self.norm2.weight, self.norm2.bias,
                                        ln[0].weight, ln[0].bias, ln[3].weight, ln[3].bias)
        x = x + self.self_attn(self.norm1(x), recs=recs, prefix=prefix + ".self_attn")
        h = _lin(self.linear_net[0], self.norm2(x), recs, prefix + ".linear_net.0")
        h = _lin(self.linear_net[3], F.relu(h), recs, prefix + ".linear_net.3")
        return x + h


class RelationalReasoning(nn.Module):
    def __init__(self, num_layers, hidden_dim, **block_args):
        super().__init__()
        self.layers = nn.ModuleList([EncoderBlock(**block_args) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_dim)

    def forward(self, x, recs=None, prefix=""):
        for i, l in enumerate(self.layers):
            x = l(x, recs=recs, prefix=f"{prefix}.layers.{i}")
        if FUSED and x.is_cuda:
            import ops
            return ops.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return self.norm(x)

    def get_attention_maps(self, x):
        maps = []
        for l in self.layers:
            _, a = l.self_attn(x, return_attention=True)      # reference feeds the un-normalised x here (RRM.py:129)
            maps.append(a)
            x = l(x)
        return maps
