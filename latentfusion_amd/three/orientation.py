"""View sampling (API mirror of the used part of latentfusion/three/orientation.py)."""
import math

import torch

from . import quaternion as q
from .core import normalize, uniform_unit_vector


def evenly_distributed_points(n, hemisphere=False, pole=(0.0, 0.0, 1.0)):
    """Sunflower (golden-angle) lattice on the sphere (reference :126-158)."""
    k = torch.arange(0, n, dtype=torch.float32) + 0.5
    phi = torch.acos(1 - 2 * k / n / 2) if hemisphere else torch.acos(1 - 2 * k / n)
    theta = math.pi * (1 + 5 ** 0.5) * k
    pts = torch.stack((torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)), dim=1)
    if hemisphere:
        up = torch.tensor((0.0, 0.0, 1.0))
        pole_t = torch.tensor(pole)
        if (up + pole_t).abs().sum() < 1e-5:
            pts = -pts
        elif (up - pole_t).abs().sum() >= 1e-5:
            axis = torch.cross(pole_t, up, dim=0).expand(n, 3)
            angle = torch.acos((pole_t * up).sum()).expand(n)
            pts = q.rotate_vector(q.from_axis_angle(axis, angle), pts)
    return pts


def random_quat_from_ray(forward, up=None):
    """Orientation whose +z axis is `forward`; roll is random unless `up` is given
    (reference :69-92 -- consumes one randn(n,3) from the global RNG when up is None)."""
    n = forward.shape[0]
    if up is None:
        down = uniform_unit_vector(n)
    else:
        down = -(torch.tensor(up).unsqueeze(0).expand(n, 3) + forward)
    right = normalize(torch.cross(down, forward, dim=-1))
    down = normalize(torch.cross(forward, right, dim=-1))
    return q.mat_to_quat(torch.stack([right, down, forward], dim=1))


def evenly_distributed_quats(n, hemisphere=False, hemisphere_pole=(0.0, 0.0, 1.0), upright=False,
                             upright_up=(0.0, 0.0, 1.0)):
    rays = evenly_distributed_points(n, hemisphere, hemisphere_pole)
    return random_quat_from_ray(-rays, upright_up if upright else None)
