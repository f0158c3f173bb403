"""DiffAugment 'color,translation,cutout' for single-channel PXD events, one fused HIP kernel per pass.

Surface of reference ``diff_aug.py`` (``DiffAugment(x, policy, channels_first)``).  The seven random
draws are taken from the global generator on ``x.device`` in the reference's order (brightness,
saturation, contrast: ``rand(N,1,1,1)``; translation: two ``randint``; cutout: two ``randint``), or can be
passed explicitly through ``draws`` so that a CPU checker consumes identical numbers.  Saturation is an
exact identity for one channel ((x - x) * r + x) and only consumes its draw.
"""
from __future__ import annotations

import torch

import _hip as H
import ops

FULL_POLICY = "color,translation,cutout"
NEXT_DRAWS = []      # FIFO of explicit draw dicts consumed by the next DiffAugment calls (tests)


def draw(n, h, w, device, generator=None):
    kw = dict(device=device, generator=generator)
    sh_x, sh_y = int(h * 0.125 + 0.5), int(w * 0.125 + 0.5)
    ch, cw = int(h * 0.5 + 0.5), int(w * 0.5 + 0.5)
    d = {"brightness": torch.rand(n, 1, 1, 1, **kw), "saturation": torch.rand(n, 1, 1, 1, **kw),
         "contrast": torch.rand(n, 1, 1, 1, **kw)}
    d["tx"] = torch.randint(-sh_x, sh_x + 1, size=[n, 1, 1], **kw)
    d["ty"] = torch.randint(-sh_y, sh_y + 1, size=[n, 1, 1], **kw)
    d["ox"] = torch.randint(0, h + (1 - ch % 2), size=[n, 1, 1], **kw)
    d["oy"] = torch.randint(0, w + (1 - cw % 2), size=[n, 1, 1], **kw)
    return d


def DiffAugment(x, policy="", channels_first=True, draws=None):
    if not policy:
        return x
    if policy != FULL_POLICY or not channels_first or x.dim() != 4 or x.shape[1] != 1:
        raise NotImplementedError("MI355X DiffAugment: policy 'color,translation,cutout' on [N,1,H,W] events "
                                  "(what model.G_D applies, reference model.py:971-978)")
    H.require_gpu()
    n, _, h, w = x.shape
    if draws is None and NEXT_DRAWS:
        draws = NEXT_DRAWS.pop(0)                          # explicit draws injected by a parity test
    d = draws if draws is not None else draw(n, h, w, x.device)
    f = lambda t: t.reshape(n).to(device=x.device, dtype=torch.float32).contiguous()
    g = lambda t: t.reshape(n).to(device=x.device, dtype=torch.int64).contiguous()
    return ops.DiffAugFn.apply(x.float(), f(d["brightness"]), f(d["contrast"]), g(d["tx"]), g(d["ty"]), g(d["ox"]), g(d["oy"]))
