"""Pose fitness loss (API mirror of default_pose_loss / weigh_losses,
latentfusion/pose/estimation.py:70-126 and pose/utils.py:81-117).

    depth     mean |D^ - D_t| over the frame (invalid target pixels zeroed)
    ov_depth  the same, averaged over the predicted-and-target mask overlap
    iou       log(union) - log(intersection) of soft masks
    mask      mean BCE-with-logits
    latent    cosine distance of 2-D latents (when both are given)

D^ = uncrop_nearest(depth crop) * sigmoid(uncrop_bilinear(mask-logit crop)).
"""
import torch
import torch.nn.functional as F

from . import utils as pu


def cosine_distance(x1, x2, dim=1, eps=1e-8):
    return 1.0 - torch.cosine_similarity(x1, x2, dim if x1.dim() > 1 else 0, eps)


def default_pose_loss(target, z_pred_depth, z_pred_mask_logits, z_pred_camera, z_pred_latent=None,
                      z_target_latent=None):
    pred_depth, _ = z_pred_camera.uncrop(z_pred_depth, scale_mode='nearest')
    pred_mask_logits, _ = z_pred_camera.uncrop(z_pred_mask_logits, scale_mode='bilinear')
    pred_mask = torch.sigmoid(pred_mask_logits)
    pred_depth = pred_depth * pred_mask
    invalid = (target.depth == 0) & (target.mask > 0.1)
    target_mask = target.mask
    target_depth = target.depth * target.mask                      # target.prepare()

    out = {}
    overlap = pred_mask * target_mask
    depth_loss = pu.zero_invalid_pixels(F.l1_loss(pred_depth, target_depth.expand_as(pred_depth), reduction='none'),
                                        invalid)
    out['ov_depth'] = pu.reduce_loss_mask(depth_loss, overlap)
    out['depth'] = depth_loss.mean(dim=(1, 2, 3))
    out['iou'] = pu.iou_loss(pred_mask, pu.zero_invalid_pixels(target_mask, invalid))
    out['mask'] = F.binary_cross_entropy_with_logits(pred_mask_logits, target_mask.expand_as(pred_mask),
                                                     reduction='none').mean(dim=(1, 2, 3))
    if z_pred_latent is not None and z_target_latent is not None:
        zp = z_pred_latent.reshape(z_pred_latent.shape[0], -1)
        zt = z_target_latent.reshape(z_target_latent.shape[0], -1)
        out['latent'] = cosine_distance(zp, zt.expand_as(zp))
    return out


def weigh_losses(loss_dict, weight_dict):
    return {k: weight_dict.get(k, 0.0) * v for k, v in loss_dict.items()}
