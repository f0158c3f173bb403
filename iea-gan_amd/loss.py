"""Losses of the IEA-GAN train step on the 40-sample logits / unit-sphere embeddings of one event.

Function surface of reference ``loss.py`` (unif_loss 8-9, IEA_loss 14-27, loss_hinge_dis 30-33, loss_hinge_gen
36-38, l2_loss 41-44, Conditional_Contrastive_loss 79-132).  Every function is one launch of the fused HIP
``loss_block`` kernel (value + gradient; the 40x40 Gram matrix the embedding losses share lives in LDS); the
train step (``train_fns.py``) evaluates ALL terms of a phase with a single call of that kernel.  No numpy
masks, no host round trips.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

import _hip as H
import ops


def _gpu(*ts):
    if not all(t is None or t.is_cuda for t in ts):
        H.require_gpu()
        raise RuntimeError("loss functions of the MI355X path take HIP tensors (no CPU fallback)")


def unif_loss(x, t=2):
    """log mean_{i<j} exp(-t |xi - xj|^2), t = 2."""
    if t != 2:
        raise NotImplementedError("unif_loss: t = 2 (the only value the reference uses)")
    _gpu(x)
    return ops.loss_block(e=x, w_unif=1.0)[0]


def IEA_loss(k_f, k_r):
    """KL_batchmean(softmax(k_r k_r^T) || softmax(k_f k_f^T)); no gradient to the real embeddings."""
    _gpu(k_f, k_r)
    return ops.loss_block(e=k_f, er=k_r.detach(), w_iea=1.0)[0]


def loss_hinge_dis(dis_fake, dis_real):
    _gpu(dis_fake, dis_real)
    return ops.loss_block(dreal=dis_real, w_hinge_real=1.0)[0], ops.loss_block(dfake=dis_fake, w_hinge_fake=1.0)[0]


def loss_hinge_gen(dis_fake):
    _gpu(dis_fake)
    return ops.loss_block(dfake=dis_fake, w_hinge_gen=1.0)[0]


def l2_loss(dis_real, dis_aug_real):
    return F.mse_loss(dis_real, dis_aug_real)


class Conditional_Contrastive_loss(torch.nn.Module):
    """2C loss with ``pos_collected_numerator=False`` (the shipped configuration): numerator exp(cos(e_i,p_i)/t),
    denominator numerator + sum_{j != i} exp(cos(e_i,e_j)/t)."""

    def __init__(self, device, batch_size, pos_collected_numerator):
        super().__init__()
        if pos_collected_numerator:
            raise NotImplementedError("MI355X contrastive loss: pos_collected_numerator=False")
        self.device, self.batch_size, self.pos_collected_numerator = device, batch_size, pos_collected_numerator

    def forward(self, inst_embed, proxy, negative_mask, labels, temperature, margin):
        if margin != 0:
            raise NotImplementedError("MI355X contrastive loss: margin = 0 (as the reference's train step passes)")
        _gpu(inst_embed, proxy)
        return ops.loss_block(e=inst_embed, p=proxy, w_contra=1.0, temperature=float(temperature))[0]
