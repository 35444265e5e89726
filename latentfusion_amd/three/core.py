"""Small homogeneous-coordinate helpers (API mirror of latentfusion/three/core.py)."""
import torch


def acos_safe(t, eps=1e-7):
    return torch.acos(t.clamp(-1.0 + eps, 1.0 - eps))


def ensure_batch_dim(tensor, num_dims):
    if tensor.dim() == num_dims:
        return tensor.unsqueeze(0), True
    return tensor, False


def normalize(vector, dim=-1):
    return vector / vector.norm(p=2.0, dim=dim, keepdim=True)


def uniform(n, min_val, max_val):
    return (max_val - min_val) * torch.rand(n) + min_val


def uniform_unit_vector(n):
    return normalize(torch.randn(n, 3), dim=1)


def inner_product(a, b):
    return (a * b).sum(dim=-1)


def homogenize(coords):
    return torch.cat((coords, torch.ones_like(coords[..., :1])), dim=-1)


def dehomogenize(coords):
    return coords[..., :-1] / coords[..., -1:]


def transform_coords(coords, transform):
    """Applies (B,4,4)/(B,3,4) transforms to (B,M,3) points (reference core.py:70-81)."""
    coords, squeezed = ensure_batch_dim(coords, 2)
    out = dehomogenize((transform @ homogenize(coords).transpose(1, 2)).transpose(1, 2))
    return out.squeeze(0) if squeezed else out


def transform_coord_grid(grid, transform):
    if transform.size(0) != grid.size(0):
        raise ValueError('Batch dimensions must match.')
    flat = homogenize(grid).view(grid.size(0), -1, grid.size(-1) + 1)
    out = (transform @ flat.transpose(1, 2)).transpose(1, 2)
    return dehomogenize(out.view(*grid.shape[:-1], transform.size(1)))


def grid_to_coords(grid):
    return grid.view(grid.size(0), -1, grid.size(-1))


def spherical_to_cartesian(theta, phi, r=1.0):
    """reference core.py:89-93 (z uses theta, as there)."""
    return torch.stack((r * torch.cos(theta) * torch.sin(phi), r * torch.sin(theta) * torch.sin(phi), r * torch.cos(theta)), dim=-1)


def points_bound(points):
    return torch.stack((torch.min(points, dim=0)[0], torch.max(points, dim=0)[0]), dim=1)


def points_radius(points):
    centroid = points_bound(points).mean(dim=1).unsqueeze(0)
    return torch.norm(points - centroid, dim=1).max()


def points_diameter(points):
    return 2 * points_radius(points)


def points_centroid(points):
    return points_bound(points).mean(dim=1)


def points_bounding_size(points):
    bounds = points_bound(points)
    return torch.norm(bounds[:, 1] - bounds[:, 0])
