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

    def hyper_dirty(self):
        """True when ``push_hyper`` would write the device block (first call, or lr / betas / eps changed on the host)."""
        g = self.param_groups[0]
        return self._hp is None or (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) != self._hp_host

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
        if self._m is None:
            self._m = torch.zeros(n, dtype=torch.float32, device=a.flat.device)
            self._v = torch.zeros(n, dtype=torch.float32, device=a.flat.device)
        elif self._m.numel() != n:
            raise RuntimeError(f"FusedAdam: optimizer state holds {self._m.numel()} elements, the network's arena {n}")
        elif self._m.device != a.flat.device:       # network moved after the state was created / loaded: follow it
            if self._hp is not None:                # the device counter hp[4] is authoritative (graph replays never touch the mirror)
                self._step = int(self._hp[4].item())
            self._m, self._v, self._hp = self._m.to(a.flat.device), self._v.to(a.flat.device), None
            self.push_hyper()
        if not torch.cuda.is_current_stream_capturing():
            self.push_hyper()
        self._step += 1          # host mirror (informational; the device counter hp[4] is authoritative)
        H.call("ieagan_adam_step", a.flat.data_ptr(), a.grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), n,
               self._hp.data_ptr(), H.stream())

    # Checkpoint format = torch.optim.Adam's own state dict ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}},
    # 'param_groups': [...]}, parameters indexed in ``parameters()`` order, which equals the reference's order -- see
    # tests/golden/state_dict_contract.json): the reference loads our G_optim.pth / D_optim.pth and we load its files.
    # The flat moment buffers are split into per-parameter HOST copies on save and gathered back on load.
    def state_dict(self):
        a = self._arena() if (self.owner is not None and self._m is not None) else None
        step = float(self._hp[4].item()) if self._hp is not None else float(self._step)
        state = {}
        if a is not None:
            for i, (p, o, n) in enumerate(a.param_slices):
                state[i] = {"step": torch.tensor(step), "exp_avg": self._m[o:o + n].view(p.shape).detach().cpu().clone(),
                            "exp_avg_sq": self._v[o:o + n].view(p.shape).detach().cpu().clone()}
        groups, k = [], 0
        for g in self.param_groups:
            d = {key: v for key, v in g.items() if key != "params"}
            d["params"] = list(range(k, k + len(g["params"])))
            k += len(g["params"])
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if sd.get("fused"):         # round-1 files: flat moments
            flat_m, flat_v, step = sd["exp_avg"], sd["exp_avg_sq"], float(sd["step"])
            per_param = None
        else:
            per_param, flat_m, flat_v = sd["state"], None, None
            step = max([float(torch.as_tensor(s["step"]).item()) for s in per_param.values()] or [0.0])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})
        self._step, self._hp = int(step), None
        self._gather_moments(per_param, flat_m, flat_v)

    def _gather_moments(self, per_param, flat_m, flat_v):
        """Moments onto the arena's device, in arena order; a size mismatch raises (it must never silently re-zero:
        that desynchronises data-parallel replicas on resume)."""
        a = self._arena()
        dev, n = a.flat.device, a.n_param
        m = torch.zeros(n, dtype=torch.float32, device=dev)
        v = torch.zeros(n, dtype=torch.float32, device=dev)
        if per_param is None:
            if flat_m.numel() != n or flat_v.numel() != n:
                raise ValueError(f"optimizer checkpoint holds {flat_m.numel()} moment elements, the network has {n}")
            m.copy_(flat_m.to(dev))
            v.copy_(flat_v.to(dev))
        else:
            if per_param and len(per_param) != len(a.param_slices):
                raise ValueError(f"optimizer checkpoint holds {len(per_param)} parameters, the network has {len(a.param_slices)}")
            for i, (p, o, cnt) in enumerate(a.param_slices):
                s = per_param.get(i)
                if s is None:
                    continue
                if s["exp_avg"].numel() != cnt:
                    raise ValueError(f"optimizer checkpoint: parameter {i} has {s['exp_avg'].numel()} elements, expected {cnt}")
                m[o:o + cnt].copy_(s["exp_avg"].reshape(-1).to(dev, torch.float32))
                v[o:o + cnt].copy_(s["exp_avg_sq"].reshape(-1).to(dev, torch.float32))
        self._m, self._v = m, v
