"""Small tensor helpers (latentfusion/functional.py:4-49)."""
import torch


def extract_features(x, submodule, layers):
    outputs = []
    for name, module in submodule.named_children():
        x = module(x)
        if name in layers:
            outputs.append(x)
    return outputs


def _stats(tensor, mean, std):
    mean = torch.as_tensor(mean, dtype=torch.float32, device=tensor.device)
    std = torch.as_tensor(std, dtype=torch.float32, device=tensor.device)
    if tensor.dim() == 4:
        return mean[None, :, None, None], std[None, :, None, None]
    if tensor.dim() == 3:
        return mean[:, None, None], std[:, None, None]
    raise ValueError(f'Unsupported number of dimensions ({tensor.dim()}.')


def normalize(tensor, mean, std):
    mean, std = _stats(tensor, mean, std)
    return (tensor - mean) / std


def denormalize(tensor, mean, std):
    mean, std = _stats(tensor, mean, std)
    return (tensor * std) + mean


def unit_normalize(tensor, dim, eps=1e-3):
    return tensor / (eps + torch.norm(tensor, dim=dim, keepdim=True))


def absolute_max_pool(tensor, dim):
    """Signed value of the largest magnitude along `dim` (kept dimension)."""
    _, index = tensor.abs().max(dim=dim, keepdim=True)
    return torch.gather(tensor, dim, index)
