"""(batch, view) <-> flat batch reshapes (API mirror of latentfusion/three/batchview.py)."""


def bv2b(x):
    return x.reshape(-1, *x.shape[2:])


def b2bv(x, num_view=-1, batch_size=-1):
    if num_view == -1 and batch_size == -1:
        raise ValueError('One of num_view or batch_size must be non-negative.')
    return x.reshape(batch_size, num_view, *x.shape[1:])
