"""Pose-accuracy metrics of the reference's evaluation code: rotation / translation distance, ADD,
ADD-S, ADD with the z-180 symmetry, 2-D projection error (latentfusion/pose/metrics.py:11-109).

Same names, argument order and return types as the reference.  ADD-S's nearest-neighbour search runs
in row blocks of `batch_size` like the reference's `best_distance`, so the peak memory is
batch_size x P instead of P x P."""
import collections
import math

import torch

from .. import three


def camera_rotation_dist(camera1, camera2):
    return three.quaternion.angular_distance(camera1.quaternion, camera2.quaternion)


def camera_translation_dist(camera1, camera2):
    return torch.norm(camera1.translation - camera2.translation, dim=-1)


def compute_point_add(extrinsic_gt, extrinsic_eval, points):
    """Mean distance between the model points under the two poses (metrics.py:76-80)."""
    p_gt = three.transform_coords(points, extrinsic_gt)
    p_ev = three.transform_coords(points, extrinsic_eval)
    return torch.mean(torch.norm(p_gt - p_ev, dim=-1))


def best_distance(x1, x2, batch_size: int = 1000):
    """For every row of x1 the distance to its nearest row of x2 (metrics.py:90-99)."""
    out = []
    for i in range(0, x1.shape[0], batch_size):
        out.append(torch.cdist(x1[i:i + batch_size], x2).min(dim=1).values)
    return torch.cat(out, dim=0) if out else x1.new_zeros(0)


def compute_point_add_s(extrinsic_gt, extrinsic_eval, points):
    """ADD-S: mean closest-point distance (metrics.py:83-87)."""
    p_gt = three.transform_coords(points, extrinsic_gt)
    p_ev = three.transform_coords(points, extrinsic_eval)
    if p_gt.dim() == 3:                                   # (1,P,3) from a batched extrinsic
        p_gt, p_ev = p_gt.reshape(-1, 3), p_ev.reshape(-1, 3)
    return torch.mean(best_distance(p_gt, p_ev))


def compute_point_add_sym(extrinsic_gt, extrinsic_eval, points):
    """min(ADD, ADD with the ground truth turned 180 degrees about z) (metrics.py:64-73)."""
    z_axis = torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float32)
    rot = three.rotation_to_4x4(three.quaternion.quat_to_mat(three.quaternion.from_axis_angle(z_axis, math.pi)))
    a0 = compute_point_add(extrinsic_gt, extrinsic_eval, points)
    a1 = compute_point_add(extrinsic_gt @ rot.to(extrinsic_gt), extrinsic_eval, points)
    return torch.min(a0, a1)


def compute_point_proj2d(proj_gt, proj_eval, points):
    """Mean image-plane distance of the projected model points (metrics.py:102-106)."""
    p_gt = three.transform_coords(points, proj_gt)
    p_ev = three.transform_coords(points, proj_eval)
    return torch.mean(torch.norm(p_gt - p_ev, dim=-1))


def camera_metrics(camera_gt, camera_eval, points, scale_to_meters, use_add=True, use_add_sym=True, use_add_s=True,
                   use_proj2d=True, **kwargs):
    """Dictionary of metrics for one camera pair, a list of them for batched cameras (metrics.py:19-61;
    like the reference, the per-pair recursion uses the default switches)."""
    if len(camera_gt) > 1:
        return [camera_metrics(c1, c2, points, scale_to_meters) for c1, c2 in zip(camera_gt, camera_eval)]
    camera_gt, camera_eval = camera_gt.clone().cpu(), camera_eval.clone().cpu()
    metrics = {
        'rotation_dist': camera_rotation_dist(camera_gt, camera_eval).squeeze().item(),
        'translation_dist': (camera_translation_dist(camera_gt, camera_eval) * scale_to_meters).squeeze().item(),
    }
    if points is not None:
        if use_add:
            metrics['add'] = compute_point_add(camera_gt.obj_to_cam, camera_eval.obj_to_cam, points) * scale_to_meters
        if use_add_s:
            metrics['add_s'] = compute_point_add_s(camera_gt.obj_to_cam, camera_eval.obj_to_cam, points) * scale_to_meters
        if use_add_sym:
            metrics['add_sym'] = (compute_point_add_sym(camera_gt.obj_to_cam, camera_eval.obj_to_cam, points)
                                  * scale_to_meters)
        if use_proj2d:
            metrics['proj2d'] = compute_point_proj2d(camera_gt.obj_to_image, camera_eval.obj_to_image, points)
    return metrics


def concat_camera_metrics(metrics_list):
    out = collections.defaultdict(list)
    for key in metrics_list[0].keys():
        for m in metrics_list:
            out[key].append(m[key])
    return out
