"""Losses of the IEA-GAN train step on the 40-token event embeddings (reference ``loss.py``: unif_loss 8-9,
IEA_loss 14-27, loss_hinge_dis 30-33, loss_hinge_gen 36-38, l2_loss 41-44, Conditional_Contrastive_loss 79-132).

Everything here acts on [40] logits and [40, 1024] unit-norm embeddings, i.e. on one 40x40 Gram matrix.
The masks the reference rebuilds on the host with numpy at every call are built once on the device.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def unif_loss(x, t=2):
    """log mean exp(-t * ||xi - xj||^2) over the i<j pairs (what torch.pdist enumerates)."""
    n = x.shape[0]
    sq = (x * x).sum(1)
    d2 = (sq[:, None] + sq[None, :] - 2.0 * (x @ x.t())).clamp_min(0.0)
    iu = torch.triu_indices(n, n, 1, device=x.device)
    return d2[iu[0], iu[1]].mul(-t).exp().mean().log()


def IEA_loss(k_f, k_r):
    with torch.no_grad():
        target = F.softmax(k_r @ k_r.t(), dim=-1)
    logp = F.log_softmax(k_f @ k_f.t(), dim=-1)
    return F.kl_div(logp, target, reduction="batchmean")


def loss_hinge_dis(dis_fake, dis_real):
    return torch.mean(F.relu(1.0 - dis_real)), torch.mean(F.relu(1.0 + dis_fake))


def loss_hinge_gen(dis_fake):
    return -torch.mean(dis_fake)


def l2_loss(dis_real, dis_aug_real):
    return F.mse_loss(dis_real, dis_aug_real)


class Conditional_Contrastive_loss(torch.nn.Module):
    def __init__(self, device, batch_size, pos_collected_numerator):
        super().__init__()
        self.device, self.batch_size, self.pos_collected_numerator = device, batch_size, pos_collected_numerator
        self._offdiag = {}

    def _mask(self, n, device):
        key = (n, str(device))
        if key not in self._offdiag:
            self._offdiag[key] = ~torch.eye(n, dtype=torch.bool, device=device)
        return self._offdiag[key]

    def remove_diag(self, M):
        h, w = M.shape
        assert h == w, "h and w should be same"
        return M[self._mask(h, M.device)].view(h, -1)

    def forward(self, inst_embed, proxy, negative_mask, labels, temperature, margin):
        sim = F.cosine_similarity(inst_embed.unsqueeze(1), inst_embed.unsqueeze(0), dim=-1)
        zone = torch.exp((self.remove_diag(sim) - margin) / temperature)
        pos = torch.exp((F.cosine_similarity(inst_embed, proxy, dim=-1) - margin) / temperature)
        if self.pos_collected_numerator:
            keep = self.remove_diag(negative_mask[labels])
            numerator = pos + (zone * keep).sum(dim=1)
        else:
            numerator = pos
        denominator = pos + zone.sum(dim=1)
        return -torch.log(temperature * (numerator / denominator)).mean()
