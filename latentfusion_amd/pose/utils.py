"""Camera sampling / (de)parameterisation and loss reductions of the pose estimators
(API mirror of latentfusion/pose/utils.py:13-117)."""
import math

import torch
from torch import nn

from .. import three
from ..modules.geometry import Camera


def perturb_camera(camera, translation_std, quaternion_std):
    camera = camera.clone()
    camera.translation = camera.translation + torch.randn_like(camera.translation) * translation_std
    camera.log_quaternion = camera.log_quaternion + torch.randn_like(camera.log_quaternion) * quaternion_std
    return camera


def sample_cameras_with_estimate(n, camera_est, translation_std=0.0, hemisphere=False, upright=False) -> Camera:
    """n cameras at the estimated translation with evenly distributed orientations
    (reference :28-45; consumes the global RNG in the same order: randn_like, then the rolls)."""
    device = camera_est.device
    translation = camera_est.translation.expand(n, -1)
    translation = translation + torch.randn_like(translation) * translation_std
    quaternion = three.orientation.evenly_distributed_quats(n, hemisphere=hemisphere, upright=upright)
    extrinsic = three.to_extrinsic_matrix(translation.cpu(), quaternion).to(device)
    return Camera(camera_est.intrinsic.expand(n, -1, -1), extrinsic, camera_est.z_span, width=camera_est.width,
                  height=camera_est.height, viewport=camera_est.viewport.expand(n, -1))


def parameterize_camera(camera, optimize_rotation=True, optimize_translation=True, optimize_viewport=False):
    cam = camera.clone()
    if optimize_rotation:
        cam.log_quaternion = nn.Parameter(cam.log_quaternion)
    if optimize_translation:
        cam.translation = nn.Parameter(cam.translation)
    if optimize_viewport:
        cam.viewport = nn.Parameter(cam.viewport)
    return cam


def deparameterize_camera(camera):
    return camera.detach().clone()


def flip_camera(camera, axis=(0.0, 0.0, 1.0)):
    ax = torch.tensor([axis], dtype=torch.float32, device=camera.device).expand(len(camera), -1)
    return camera.clone().rotate(three.quaternion.from_axis_angle(ax, math.pi))


def zero_invalid_pixels(tensor, invalid_mask):
    return tensor * (~invalid_mask).float()


def iou_loss(input_mask, target_mask, eps=1e-4):
    """log(union) - log(intersection) with clamps (reference :99-108)."""
    inter = torch.sum(input_mask * target_mask, dim=(1, 2, 3))
    union = torch.sum(input_mask, dim=(1, 2, 3)) + torch.sum(target_mask, dim=(1, 2, 3)) - inter
    return torch.log(union.clamp(min=eps)) - torch.log(inter.clamp(min=eps))


def reduce_loss_mask(loss, mask, eps=1e-4):
    if loss.dim() == 4:
        loss = loss.squeeze(1)
    if mask.dim() == 4:
        mask = mask.squeeze(1)
    return (loss * mask).sum(dim=(-2, -1)).clamp(min=eps / 10) / mask.sum(dim=(-2, -1)).clamp(min=eps)
