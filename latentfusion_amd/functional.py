"""Small tensor helpers of the reference (latentfusion/functional.py:4-49): feature taps of a sequential
module, per-channel (de)normalisation of CHW / NCHW images, unit normalisation, signed max-magnitude pooling."""
import torch


def extract_features(x, submodule, layers):
    """Runs the children of `submodule` in order and collects the outputs of those named in `layers`."""
    taps = []
    for name, child in submodule.named_children():
        x = child(x)
        if name in layers:
            taps.append(x)
    return taps


def _per_channel(tensor, values):
    """`values` (one per channel) shaped to broadcast against a CHW or NCHW tensor."""
    if tensor.dim() not in (3, 4):
        raise ValueError(f'Unsupported number of dimensions ({tensor.dim()}.')
    v = torch.as_tensor(values, dtype=torch.float32, device=tensor.device)
    return v.view(*([1] * (tensor.dim() - 3)), -1, 1, 1)


def normalize(tensor, mean, std):
    return (tensor - _per_channel(tensor, mean)) / _per_channel(tensor, std)


def denormalize(tensor, mean, std):
    return tensor * _per_channel(tensor, std) + _per_channel(tensor, mean)


def unit_normalize(tensor, dim, eps=1e-3):
    return tensor / (tensor.norm(dim=dim, keepdim=True) + eps)


def absolute_max_pool(tensor, dim):
    """Along `dim` (kept), the entry of largest magnitude with its sign."""
    return tensor.take_along_dim(tensor.abs().argmax(dim=dim, keepdim=True), dim=dim)
