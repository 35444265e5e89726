"""Initial pose from the target's depth and mask: translation from the mask centroid and the mid-range
of the (outlier-rejected) masked depth, identity rotation (latentfusion/pose/initialization.py:8-99).

Reproduced quirk (DESIGN Q18): `_erode_mask` builds the eroded mask and then returns the ORIGINAL one,
because its guard `len(eroded) < 10` looks at the leading (size-1) axis (initialization.py:35-43).  The
erosion itself (skimage.morphology.binary_erosion with a disk, border treated as foreground) is kept so
that the function changes behaviour exactly when the reference's would."""
import torch
import torch.nn.functional as F

from .. import three
from ..modules.geometry import Camera


def _masks_to_viewports(masks, pad: float = 10):
    """Tight (xmin, ymin, xmax, ymax) of every mask, grown by `pad` (initialization.py:8-24)."""
    out = []
    padding = torch.tensor([-pad, -pad, pad, pad], dtype=torch.float32, device=masks.device)
    for mask in masks:
        coords = torch.nonzero(mask.squeeze()).float()
        out.append(torch.stack([coords[:, 1].min(), coords[:, 0].min(), coords[:, 1].max(), coords[:, 0].max()]) + padding)
    return torch.stack(out, dim=0)


def _masks_to_centroids(masks):
    vp = _masks_to_viewports(masks, 0.0)
    return torch.stack(((vp[:, 2] + vp[:, 0]) / 2.0, (vp[:, 3] + vp[:, 1]) / 2.0), dim=-1)


def _disk(radius, device):
    r = torch.arange(-radius, radius + 1, device=device)
    return ((r[:, None] ** 2 + r[None, :] ** 2) <= radius ** 2).float()


def _erode_mask(mask, size=5):
    """mask: (1,H,W) bool.  See the module docstring for why this returns `mask`."""
    k = _disk(size, mask.device)
    m = F.pad(mask.float().unsqueeze(0), (size, size, size, size), value=1.0)          # border counts as foreground
    eroded = (F.conv2d(m, k[None, None]) >= k.sum() - 0.5).squeeze(0)                  # (1,H,W)
    if len(eroded) < 10:
        return mask
    return eroded


def _reject_outliers(data, m=1.5):
    keep = torch.abs(data - torch.median(data)) < m * torch.std(data)
    return data[keep], int((~keep).sum().item())


def _reject_outliers_mad(data, m=2.0):
    median = data.median()
    mad = torch.median(torch.abs(data - median))
    keep = torch.abs(data - median) / mad < m
    return data[keep], int((~keep).sum().item())


def _estimate_camera_dist(depth, mask):
    zs = torch.zeros(depth.shape[0], device=depth.device)
    mask = mask.bool()
    for i in range(depth.shape[0]):
        m = _erode_mask(mask[i], size=3)
        vals = depth[i][m & (depth[i] > 0.0)]
        vals, _ = _reject_outliers_mad(vals, m=3.0)
        zs[i] = (vals.min() + vals.max()) / 2.0
    return zs


def estimate_translation(depth, mask, intrinsic):
    z_cam = _estimate_camera_dist(depth, mask)
    uv = _masks_to_centroids(mask)
    u0, v0 = intrinsic[..., 0, 2], intrinsic[..., 1, 2]
    fu, fv = intrinsic[..., 0, 0], intrinsic[..., 1, 1]
    return (uv[:, 0] - u0) / fu * z_cam, (uv[:, 1] - v0) / fv * z_cam, z_cam


def estimate_initial_pose(depth, mask, intrinsic, width, height) -> Camera:
    """Camera with the estimated translation and identity rotation (initialization.py:89-99)."""
    translation = torch.stack(estimate_translation(depth, mask, intrinsic), dim=-1)
    rotation = three.quaternion.identity(intrinsic.shape[0], intrinsic.device)
    extrinsic = three.to_extrinsic_matrix(translation, rotation)
    return Camera(intrinsic, extrinsic, height=height, width=width)
