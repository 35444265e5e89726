"""Vector distances of the reference (latentfusion/distances.py:5-42): row-wise (`cosine_distance`,
`pairwise_distance`, `distance`) and all-pairs (`outer_distance`) variants over the metrics
cosine / euclidean / inner / ols_coef."""
import torch
import torch.nn.functional as F


def _cos_dist(x1, x2, dim, eps):
    return 1.0 - F.cosine_similarity(x1, x2, dim=dim, eps=eps)


def cosine_distance(x1, x2, dim=1, eps=1e-8):
    """1 - cos(x1, x2) along `dim` (plain vectors: along their only axis)."""
    return _cos_dist(x1, x2, 0 if x1.dim() == 1 else dim, eps)


def pairwise_distance(x1, x2, metric='cosine', p=2, eps=1e-8):
    table = {'cosine': lambda: _cos_dist(x1, x2, 1, eps),
             'euclidean': lambda: F.pairwise_distance(x1, x2, p=p, eps=eps)}
    if metric not in table:
        raise ValueError(f'Unknown type {metric!r}')
    return table[metric]()


def distance(x1, x2, metric='cosine', p=2, eps=1e-8, dim=0):
    """Cosine distance, or the p-norm of the difference for every other metric name."""
    return _cos_dist(x1, x2, dim, eps) if metric == 'cosine' else (x1 - x2).norm(p=p, dim=dim)


def outer_distance(x1, x2, metric='cosine', p=2, eps=1e-8):
    """(len(x1), len(x2)) matrix of distances between the rows of x1 and the rows of x2."""
    if metric == 'euclidean':
        return torch.cdist(x1, x2)
    gram = x1 @ x2.t()
    n1 = x1.norm(dim=1, keepdim=True)
    if metric == 'inner':
        return -gram
    if metric == 'cosine':
        return 1.0 - gram / (n1 @ x2.norm(dim=1, keepdim=True).t()).clamp(min=eps)
    if metric == 'ols_coef':
        return -(gram / n1.pow(2).clamp(min=eps))
    raise ValueError(f'Unknown type {metric!r}')
