"""Flat fp32 arena for all parameters and buffers of a network.

MI355X-first memory layout: one contiguous HBM allocation per network, ``[parameters | buffers]``,
each tensor 32-byte aligned.  ``nn.Parameter`` objects and registered buffers become views into it, so
``state_dict()`` keeps the reference's key names and shapes while

* the batched spectral-norm kernels address every layer by offset (one launch for all layers),
* Adam and EMA are one fused launch over the whole arena,
* data-parallel training all-reduces ONE gradient buffer per network over RCCL/xGMI
  (``Arena.grad`` -- the ``.grad`` of every parameter is a view into it).
"""
from __future__ import annotations

import torch
import torch.nn as nn

ALIGN = 8  # floats


class Arena:
    def __init__(self, root: nn.Module):
        params = [(n, p) for n, p in root.named_parameters()]
        owners = []
        for mn, m in root.named_modules():
            for bn, b in m._buffers.items():
                if b is not None:
                    owners.append((m, bn, b))
        tensors = [p for _, p in params] + [b for _, _, b in owners]
        if not tensors:
            raise ValueError("module has no state")
        dev = tensors[0].device
        for t in tensors:
            if t.dtype != torch.float32 or t.device != dev:
                raise TypeError("arena holds fp32 state on one device (master weights stay fp32; bf16 is an activation type)")
        offs, cur = [], 0
        for t in tensors:
            offs.append(cur)
            cur += (max(t.numel(), 1) + ALIGN - 1) // ALIGN * ALIGN
            if len(offs) == len(params):
                self.n_param = cur
        if not params:
            self.n_param = 0
        self.flat = torch.zeros(cur, dtype=torch.float32, device=dev)
        self.param_slices = []
        with torch.no_grad():
            for (n, p), o in zip(params, offs[:len(params)]):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.param_slices.append((p, o, p.numel()))
            for (m, bn, b), o in zip(owners, offs[len(params):]):
                v = self.flat[o:o + b.numel()].view(b.shape)
                v.copy_(b)
                m._buffers[bn] = v
        self.grad = None
        self.root = root
        for m in root.modules():
            m.__dict__["_arena"] = self

    # ---- gradients as one buffer ------------------------------------------------------------------
    def attach_grads(self):
        """Make every parameter's ``.grad`` a view into one flat buffer (zeroed)."""
        if self.grad is None:
            self.grad = torch.zeros(self.n_param, dtype=torch.float32, device=self.flat.device)
        else:
            self.grad.zero_()
        for p, o, n in self.param_slices:
            p.grad = self.grad[o:o + n].view(p.shape)
        return self.grad

    def grads_attached(self) -> bool:
        if self.grad is None:
            return False
        base, end = self.grad.data_ptr(), self.grad.data_ptr() + self.grad.numel() * 4
        return all(p.grad is not None and base <= p.grad.data_ptr() < end for p, _, _ in self.param_slices)

    def contains(self, t: torch.Tensor) -> bool:
        base = self.flat.data_ptr()
        return t.device == self.flat.device and base <= t.data_ptr() < base + self.flat.numel() * 4


def arena_of(module: nn.Module) -> Arena:
    """The arena a module's state lives in; flattens the module (stand-alone use) when it has none
    or when ``.to()`` / ``.cuda()`` has moved its tensors out of it."""
    a = module.__dict__.get("_arena")
    probe = next(iter(module.parameters()), None)
    if probe is None:
        probe = next(iter(module.buffers()))
    if a is None or not a.contains(probe):
        a = Arena(module)
    return a
