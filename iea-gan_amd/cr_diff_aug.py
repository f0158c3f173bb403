"""Consistency-regularisation augmentation of real events: per-image horizontal flip (p = 0.5) followed
by a reflect-padded translation of up to 1/8 of each side -- one gather kernel.

Surface of reference ``cr_diff_aug.py`` (``CR_DiffAug(x, flip, translation)``).  Draw order as the
reference: a uniform [N,1] for the flip, then ``randint`` t_x and t_y, all on ``x.device``.
"""
from __future__ import annotations

import torch

import _hip as H
import ops


def draw(n, h, w, device, generator=None):
    # The reference draws the flip uniforms from the CPU generator (cr_diff_aug.py:24) and copies them over; here all three draws
    # come from the generator of ``device``: no host-to-device copy inside the step (a captured HIP graph cannot contain one),
    # same distribution.  Bit-level replay of the reference's stream is the job of the explicit ``draws`` argument.
    d = {"flip": torch.rand(n, 1, device=device, generator=generator)}
    d["tx"] = torch.randint(-int(h / 8), int(h / 8) + 1, size=[n, 1, 1], device=device, generator=generator)
    d["ty"] = torch.randint(-int(w / 8), int(w / 8) + 1, size=[n, 1, 1], device=device, generator=generator)
    return d


def CR_DiffAug(x, flip=True, translation=True, draws=None):
    if not (flip or translation):
        return x
    if x.dim() != 4 or x.shape[1] != 1:
        raise NotImplementedError("MI355X CR_DiffAug: single-channel [N,1,H,W] events")
    H.require_gpu()
    n, _, h, w = x.shape
    d = draws if draws is not None else draw(n, h, w, x.device)
    flip_u = d["flip"].reshape(n).to(device=x.device, dtype=torch.float32) if flip else torch.ones(n, device=x.device)
    zero = torch.zeros(n, dtype=torch.int64, device=x.device)
    tx = d["tx"].reshape(n).to(device=x.device, dtype=torch.int64) if translation else zero
    ty = d["ty"].reshape(n).to(device=x.device, dtype=torch.int64) if translation else zero
    return ops.cr_diffaug(x.float(), flip_u.contiguous(), tx.contiguous(), ty.contiguous())
