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


def repeat_tensor_as(tensor, shape_ref, num_shape_dims=3):
    """Expands a (C, *shape) tensor over the leading batch dims of `shape_ref` (reference :25-32)."""
    shape_dims = shape_ref.shape[-num_shape_dims:]
    num_batch_dims = shape_ref.dim() - num_shape_dims - 1
    for _ in range(num_batch_dims):
        tensor = tensor.unsqueeze(0)
    return tensor.expand(*shape_ref.shape[:num_batch_dims], -1, *shape_dims)


def _process_batch(batch, rotation, cube_size, camera_dist, input_size, device, is_gt):
    """(B,V,...) renders -> zoomed, normalised (B,V,...) tensors + Camera (reference :68-103)."""
    from ..modules.geometry import Camera
    from ..three.batchview import b2bv, bv2b
    batch_size = batch['mask'].shape[0]
    extrinsic = bv2b(batch['extrinsic'].to(device))
    intrinsic = bv2b(batch['intrinsic'].to(device))
    mask = bv2b(batch['mask'].unsqueeze(2).float().to(device))
    image = bv2b(batch['render'].to(device) * 2.0 - 1.0)                 # gan_normalize
    depth = bv2b(batch['depth'].unsqueeze(2).to(device)) if 'depth' in batch else None
    camera = Camera(intrinsic, extrinsic, z_span=cube_size / 2.0, height=image.size(2), width=image.size(3)).to(device)
    if rotation is not None:
        camera.rotate(rotation.expand(camera.length, -1))
    out = {}
    out['image'], out['camera'] = camera.zoom(image, target_size=input_size, target_dist=camera_dist, scale_mode='bilinear')
    out['mask'] = camera.zoom(mask, target_size=input_size, target_dist=camera_dist, scale_mode='nearest')[0]
    if depth is not None:
        out['depth'] = camera.normalize_depth(
            camera.zoom(depth, target_size=input_size, target_dist=camera_dist, scale_mode='nearest')[0])
    if is_gt:
        out['image'] = out['image'] * out['mask']
        out['depth'] = mask_normalized_depth(out['depth'], out['mask'])
    for k in ('image', 'depth', 'mask'):
        if k in out:
            out[k] = b2bv(out[k], batch_size=batch_size)
    return out


def process_batch(batch, cube_size, camera_dist, input_size, device, random_orientation=True):
    """Moves a training batch to the device, folds views into the batch axis, zooms everything to the
    canonical camera and optionally applies ONE random rotation to all cameras (reference :106-127)."""
    from ..three import quaternion
    rand_rot = quaternion.random(1).to(device) if random_orientation else None
    return {k: _process_batch(v, rand_rot, cube_size, camera_dist, input_size, device, is_gt='gt' in k)
            for k, v in batch.items()}
