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

    def forward(self, x, recs=None, prefix=""):
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
        return self.norm(x)

    def get_attention_maps(self, x):
        maps = []
        for l in self.layers:
            _, a = l.self_attn(x, return_attention=True)      # reference feeds the un-normalised x here (RRM.py:129)
            maps.append(a)
            x = l(x)
        return maps
