"""(batch, view) <-> flat batch reshapes (API mirror of latentfusion/three/batchview.py)."""
import torch


def bv2b(x):
    return x.reshape(-1, *x.shape[2:])


def b2bv(x, num_view=-1, batch_size=-1):
    if num_view == -1 and batch_size == -1:
        raise ValueError('One of num_view or batch_size must be non-negative.')
    return x.reshape(batch_size, num_view, *x.shape[1:])


def vcat(tensors, batch_size):
    """Concatenate (B*V_i, ...) tensors along the view axis (reference batchview.py:31-33)."""
    return bv2b(torch.cat([b2bv(t, batch_size=batch_size) for t in tensors], dim=1))


def vsplit(tensor, sections):
    views = b2bv(tensor, num_view=sum(sections))
    return tuple(bv2b(t) for t in torch.split(views, sections, dim=1))
