"""Reconstruction losses of the generator's training step (latentfusion/losses.py:33-103,
latentfusion/trainutils.py:102-133): hard-pixel mining over a base loss, the reductions, the Beta prior
on the predicted mask, and the criterion / optimiser factories with the reference's names and defaults."""
import torch
from torch import nn, optim


def reduce_loss(loss, reduction='mean', dim=None):
    """losses.py:59-72."""
    if reduction is None:
        return loss
    if reduction == 'mean':
        return loss.mean() if dim is None else loss.mean(dim=dim)
    if reduction == 'sum':
        return loss.sum() if dim is None else loss.sum(dim=dim)
    raise ValueError(f'Unknown reduction {reduction!r}')


class HardPixelLoss(nn.Module):
    """Mean (or sum) of the k largest per-pixel losses of every image (losses.py:33-56): the base loss is
    evaluated per element, reduced over channels, and the k hardest pixels of each (B*V) image are kept."""

    def __init__(self, base_loss, k, reduction='mean', **kwargs):
        super().__init__()
        self.base_loss = base_loss(reduction='none', **kwargs)
        self.k = k
        self.reduction = reduction

    def forward(self, x, y):
        if x.dim() > 4:
            x = x.reshape(-1, *x.shape[-3:])
        if y.dim() > 4:
            y = y.reshape(-1, *y.shape[-3:])
        if x.dim() != 4:
            raise ValueError('x must be in BCHW format')
        if y.dim() != 4:
            raise ValueError('y must be in BCHW format')
        loss = self.base_loss(x, y)
        loss = reduce_loss(loss, dim=1, reduction=self.reduction).reshape(x.size(0), -1)
        loss, _ = torch.topk(loss, k=self.k, dim=1, largest=True)
        return reduce_loss(loss, self.reduction)


def lsgan_loss(input, target, reduction='mean'):
    """Least-squares GAN loss (losses.py:75-77)."""
    return reduce_loss((input.squeeze() - target) ** 2, reduction=reduction)


def multiscale_lsgan_loss(inputs, target, reduction='mean'):
    loss = 0
    for x in inputs:
        loss = loss + lsgan_loss(x, target, reduction)
    return loss


def _log_beta(alpha, beta):
    alpha, beta = torch.tensor(alpha), torch.tensor(beta)
    return torch.lgamma(alpha) + torch.lgamma(beta) - torch.lgamma(alpha + beta)


def beta_prior_loss(tensor, alpha, beta, reduction='mean', eps=1e-4):
    """Negative log-density of Beta(alpha, beta) clamped at 0 (losses.py:94-100)."""
    loss = ((alpha - 1.0) * torch.log(tensor.clamp(min=eps)) + (beta - 1.0) * torch.log((1.0 - tensor).clamp(min=eps))
            - _log_beta(alpha, beta).to(tensor.device))
    return reduce_loss((-loss).clamp(min=0), reduction=reduction)


def get_recon_criterion(loss_type, k=2000):
    """trainutils.py:114-133 (the VGG perceptual variant needs torchvision weights and is not provided)."""
    if loss_type == 'smooth_l1':
        return nn.SmoothL1Loss()
    if loss_type == 'l1':
        return nn.L1Loss()
    if loss_type == 'hard_l1':
        return HardPixelLoss(nn.L1Loss, k=k)
    if loss_type == 'hard_smooth_l1':
        return HardPixelLoss(nn.SmoothL1Loss, k=k)
    if loss_type == 'binary_cross_entropy':
        return nn.BCEWithLogitsLoss(reduction='none')
    raise ValueError(f'Unknown recon_loss_type {loss_type!r}.')


def get_optimizer(parameters, name, lr):
    """trainutils.py:102-111: note the GAN-style betas (0.0, 0.99)."""
    if name == 'adam':
        return optim.Adam(parameters, lr=lr, betas=(0.0, 0.99))
    if name == 'adamw':
        return optim.AdamW(parameters, lr=lr, betas=(0.0, 0.99))
    if name == 'sgd':
        return optim.SGD(parameters, lr=lr)
    raise ValueError(f'Unknown optimizer {name!r}')
