"""Signed max-magnitude pooling (latentfusion/functional.py:47-49), the one helper of that module on the
reconstruct-and-render path (PoolFuser 'abs_max')."""


def absolute_max_pool(tensor, dim):
    """Along `dim` (kept), the entry of largest magnitude with its sign."""
    return tensor.take_along_dim(tensor.abs().argmax(dim=dim, keepdim=True), dim=dim)
