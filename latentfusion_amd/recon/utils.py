"""Coordinate channels and the canonical camera distance (API mirror of the hot-path part of
latentfusion/recon/utils.py:13-65)."""
import math

import torch


def optimal_camera_dist(focal_length, size, radius, slack=1.5):
    """Distance at which a sphere of `radius` fills a `size`-pixel viewport (reference :13-22)."""
    theta = math.atan2(size / 2.0, focal_length)
    x = radius * math.cos(theta) / math.sin(theta)
    return math.sqrt(x ** 2 + radius ** 2 - 2 * x * radius * math.cos(math.pi / 2.0 - theta)) + slack


def get_normalized_voxel_coords(tensor):
    """(.., 3, D, H, W) channels ordered (z, y, x), each linspace(-1, 1) (reference :35-43)."""
    D, H, W = tensor.shape[-3:]
    dev = tensor.device
    z, y, x = torch.meshgrid(torch.linspace(-1.0, 1.0, D, device=dev), torch.linspace(-1.0, 1.0, H, device=dev),
                             torch.linspace(-1.0, 1.0, W, device=dev), indexing='ij')
    coords = torch.stack((z, y, x), dim=0)
    return coords.expand(*tensor.shape[:-4], -1, -1, -1, -1)


def get_normalized_pixel_coords(tensor):
    H, W = tensor.shape[-2:]
    dev = tensor.device
    y, x = torch.meshgrid(torch.linspace(-1.0, 1.0, H, device=dev), torch.linspace(-1.0, 1.0, W, device=dev),
                          indexing='ij')
    return torch.stack((y, x), dim=0).expand(*tensor.shape[:-3], -1, -1, -1)


def get_normalized_voxel_depth(tensor):
    B, _, D, H, W = tensor.shape
    return torch.linspace(-1.0, 1.0, D, device=tensor.device).view(1, 1, D, 1, 1).expand(B, 1, D, H, W)


def mask_normalized_depth(depth, mask):
    return ((depth / 2.0 + 0.5) * mask) * 2.0 - 1.0
