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


def spiral_orbit(n, c=16):
    """n pure quaternions spiralling from pole to pole (reference orientation.py:9-13)."""
    phi = torch.linspace(0, math.pi, n)
    return q.from_spherical(phi, c * phi)


def _check_up(up, n):
    if not torch.is_tensor(up):
        up = torch.tensor(up, dtype=torch.float32)
    if up.dim() == 1:
        up = up.expand(n, -1)
    return normalize(up)


def _is_ray_in_segment(ray, up, min_angle, max_angle):
    angle = torch.acos((up * ray).sum(dim=-1))
    return (min_angle <= angle) & (angle <= max_angle)


def sample_segment_rays(n, up, min_angle, max_angle):
    """Rejection-samples unit rays whose angle to `up` lies in [min_angle, max_angle] (reference :29-40)."""
    up = _check_up(up, n)
    rays = normalize(torch.randn(n, 3))
    num_invalid = n
    while num_invalid > 0:
        valid = _is_ray_in_segment(rays, up, min_angle, max_angle)
        num_invalid = int((~valid).sum().item())
        rays[~valid] = normalize(torch.randn(num_invalid, 3))
    return normalize(rays)


def sample_hemisphere_rays(n, up):
    """Uniform rays reflected into the hemisphere around `up` (reference :43-67)."""
    up = _check_up(up, n)
    rays = normalize(torch.randn(n, 3))
    dot = (up * rays).sum(dim=-1)
    rays[dot < 0] = rays[dot < 0] - 2 * dot[dot < 0, None] * up[dot < 0]
    return rays


def sample_segment_quats(n, up, min_angle, max_angle):
    """Random yaw about `up`, then tilt into the sphere segment (reference :95-123)."""
    up = _check_up(up, n)
    yaw_quat = q.from_axis_angle(up, torch.rand(n) * math.pi * 2.0)
    rays = sample_segment_rays(n, up, min_angle, max_angle)
    pivot = torch.cross(up, rays, dim=-1)
    angles = torch.acos((up * rays).sum(dim=-1))
    return q.qmul(q.from_axis_angle(pivot, angles), yaw_quat)
