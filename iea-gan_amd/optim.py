"""Fused Adam over a network's flat parameter arena (one HIP launch per step).

Drop-in for the ``torch.optim.Adam(params, lr, betas, weight_decay=0, eps)`` the reference builds in
``model.py:410-416, 858-864``: same constructor arguments, ``step`` / ``zero_grad`` / ``state_dict``.
Gradients live in ``Arena.grad`` (every ``p.grad`` is a view), which is also the single buffer the
data-parallel path all-reduces.
"""
from __future__ import annotations

import torch

import _hip as H
from arena import arena_of


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, owner=None):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam: weight_decay=0, amsgrad=False (the reference's settings)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.owner = owner          # the nn.Module whose arena holds exactly these parameters
        self._m = self._v = self._hp = None
        self._hp_host = None
        self._step = 0

    def _arena(self):
        a = self.owner.__dict__.get("_arena")
        probe = next(self.owner.parameters())
        if a is None or a.root is not self.owner or not a.contains(probe):
            from arena import Arena
            a = Arena(self.owner)
            if hasattr(self.owner, "_plan"):
                self.owner._plan = None
        return a

    def zero_grad(self, set_to_none: bool = False):
        if self.owner is None or not next(self.owner.parameters()).is_cuda:
            return super().zero_grad(set_to_none=set_to_none)
        self._arena().attach_grads()

    def push_hyper(self):
        """Mirror lr / betas / eps into the device block the kernel reads (no-op when unchanged); called
        by ``step`` and, when the train step is replayed from a HIP graph, once per replay."""
        g = self.param_groups[0]
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
        if self._hp is None:
            a = self._arena()
            self._hp = torch.tensor([*cur, float(self._step), 1.0, 0.0, 0.0], dtype=torch.float32, device=a.flat.device)
            self._hp_host = cur
        elif cur != self._hp_host:
            self._hp[0:4] = torch.tensor(cur, dtype=torch.float32)
            self._hp_host = cur

    @torch.no_grad()
    def step(self, closure=None):
        H.require_gpu()
        a = self._arena()
        if not a.grads_attached():
            # gradients were produced without zero_grad(): gather them into the flat buffer once
            old = [(p, p.grad) for p, _, _ in a.param_slices]
            a.attach_grads()
            for p, g in old:
                if g is not None:
                    p.grad.copy_(g)
        n = a.n_param
        if self._m is None or self._m.numel() != n or self._m.device != a.flat.device:
            self._m = torch.zeros(n, dtype=torch.float32, device=a.flat.device)
            self._v = torch.zeros(n, dtype=torch.float32, device=a.flat.device)
        if not torch.cuda.is_current_stream_capturing():
            self.push_hyper()
        self._step += 1          # host mirror (informational; the device counter hp[4] is authoritative)
        H.call("ieagan_adam_step", a.flat.data_ptr(), a.grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), n,
               self._hp.data_ptr(), H.stream())

    # checkpoint format: flat moments + step (the reference's per-parameter dict does not survive the
    # arena layout; utils.load_weights accepts both)
    def state_dict(self):
        step = int(self._hp[4].item()) if self._hp is not None else self._step
        return {"fused": True, "step": step, "exp_avg": self._m, "exp_avg_sq": self._v,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if not sd.get("fused"):
            raise ValueError("FusedAdam.load_state_dict expects a FusedAdam checkpoint")
        self._step, self._m, self._v = sd["step"], sd["exp_avg"], sd["exp_avg_sq"]
        self._hp = None
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
