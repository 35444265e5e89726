"""Pose fitness loss (API mirror of default_pose_loss / weigh_losses,
latentfusion/pose/estimation.py:70-126 and pose/utils.py:81-117).

    depth     mean |D^ - D_t| over the frame (invalid target pixels zeroed)
    ov_depth  the same, averaged over the predicted-and-target mask overlap
    iou       log(union) - log(intersection) of soft masks
    mask      mean BCE-with-logits
    latent    cosine distance of 2-D latents (when both are given)

D^ = uncrop_nearest(depth crop) * sigmoid(uncrop_bilinear(mask-logit crop)).

Device tensors take the fused HIP kernels (lf_pose_loss_fwd_depth / _bwd_depth through `_PoseLossTerms`, round 5): uncrop, the
four frame reductions and their adjoints are four launches each way, fixed-order sums, no ATen reduction over the frame --
the same kernels the render-loop engine uses, in the form that takes the metric depth crop.  Host tensors (CPU tests of the
estimators' logic) evaluate the expressions below.
"""
import torch
import torch.nn.functional as F

from . import utils as pu


def cosine_distance(x1, x2, dim=1, eps=1e-8):
    return 1.0 - torch.cosine_similarity(x1, x2, dim if x1.dim() > 1 else 0, eps)


class _PoseLossTerms(torch.autograd.Function):
    """(crop (N,2,h,w) = [metric depth, mask logit], coefs (N,24) from engine.camera_coefs, target depth / mask (H*W))
    -> (depth, ov_depth, iou, mask), each (N,), differentiable w.r.t. the crop and the coefficient block (entries 18..21: the
    uncrop map, i.e. the camera's viewport)."""

    @staticmethod
    def forward(ctx, crop, coefs, tdepth, tmask, H, W):
        from .. import _lib, ops
        L = _lib.lib()
        lg = ops.cl(crop)                                          # channels-last == [N][h*w][2]
        n, _, h, w = lg.shape
        nbytes = L.lf_pose_loss_scratch_bytes(n, h, w, H, W)
        scratch = torch.empty(nbytes // 4 + 1, device=lg.device, dtype=torch.float32)
        sums = torch.empty(n, 8, device=lg.device, dtype=torch.float32)
        losses = torch.empty(n, 8, device=lg.device, dtype=torch.float32)
        gsums = torch.empty(n, 8, device=lg.device, dtype=torch.float32)
        ones = torch.ones(4, device=lg.device, dtype=torch.float32)
        cf = coefs.detach().contiguous()
        with ops._timed('pose_loss_terms'):
            _lib.check(L.lf_pose_loss_fwd_depth(lg.data_ptr(), cf.data_ptr(), tdepth.data_ptr(), tmask.data_ptr(), ones.data_ptr(),
                                                sums.data_ptr(), losses.data_ptr(), gsums.data_ptr(), scratch.data_ptr(),
                                                scratch.numel() * 4, n, h, w, H, W, torch.cuda.current_stream().cuda_stream),
                       'lf_pose_loss_fwd_depth')
        ctx.save_for_backward(lg, cf, tdepth, tmask, sums)
        ctx.dims = (n, h, w, H, W)
        ctx.set_materialize_grads(False)
        return losses[:, 0].clone(), losses[:, 1].clone(), losses[:, 2].clone(), losses[:, 3].clone()

    @staticmethod
    def backward(ctx, g_depth, g_ov, g_iou, g_mask):
        from .. import _lib
        L = _lib.lib()
        lg, cf, tdepth, tmask, S = ctx.saved_tensors
        n, h, w, H, W = ctx.dims
        z = torch.zeros(n, device=lg.device, dtype=torch.float32)
        g_depth, g_ov, g_iou, g_mask = (z if g is None else g.float() for g in (g_depth, g_ov, g_iou, g_mask))
        # d(terms)/d(sums), the algebra of pose_loss_finish_kernel with the incoming per-sample gradients as weights
        inv_hw = 1.0 / float(H * W)
        num, den = S[:, 1].clamp_min(1e-5), S[:, 2].clamp_min(1e-4)
        uni = S[:, 3] + S[:, 5] - S[:, 4]
        d_uni = torch.where(uni > 1e-4, 1.0 / uni, torch.zeros_like(uni))
        gs = torch.stack((g_depth * inv_hw,
                          g_ov * torch.where(S[:, 1] > 1e-5, 1.0 / den, torch.zeros_like(den)),
                          g_ov * torch.where(S[:, 2] > 1e-4, -num / (den * den), torch.zeros_like(den)),
                          g_iou * d_uni,
                          g_iou * (-d_uni - torch.where(S[:, 4] > 1e-4, 1.0 / S[:, 4], torch.zeros_like(uni))),
                          z, g_mask * inv_hw, z), dim=1).contiguous()
        nbytes = L.lf_pose_loss_scratch_bytes(n, h, w, H, W)
        scratch = torch.empty(nbytes // 4 + 1, device=lg.device, dtype=torch.float32)
        gcrop = torch.empty_like(lg)
        gcoefs = torch.zeros(n, 24, device=lg.device, dtype=torch.float32)
        _lib.check(L.lf_pose_loss_bwd_depth(lg.data_ptr(), cf.data_ptr(), tdepth.data_ptr(), tmask.data_ptr(), gs.data_ptr(),
                                            gcrop.data_ptr(), gcoefs.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                            n, h, w, H, W, torch.cuda.current_stream().cuda_stream), 'lf_pose_loss_bwd_depth')
        return gcrop, gcoefs, None, None, None, None


def _fused_ok(target, z_pred_depth, z_pred_mask_logits, cam):
    d, m = target.depth, target.mask
    return (z_pred_depth.is_cuda and z_pred_mask_logits.is_cuda and d is not None and m is not None and d.is_cuda and m.is_cuda
            and z_pred_depth.dim() == 4 and z_pred_depth.shape[1] == 1 and z_pred_depth.shape == z_pred_mask_logits.shape
            and d.dim() == 4 and d.shape[0] == 1 and d.shape[1] == 1 and m.shape == d.shape
            and tuple(d.shape[-2:]) == (int(cam.height), int(cam.width)) and min(z_pred_depth.shape[-2:]) > 1
            and z_pred_depth.dtype == torch.float32 and z_pred_mask_logits.dtype == torch.float32)


def default_pose_loss(target, z_pred_depth, z_pred_mask_logits, z_pred_camera, z_pred_latent=None,
                      z_target_latent=None):
    if _fused_ok(target, z_pred_depth, z_pred_mask_logits, z_pred_camera):
        from ..engine import camera_coefs
        h, w = z_pred_depth.shape[-2:]
        H, W = target.depth.shape[-2:]
        coefs = camera_coefs(z_pred_camera, 1.0, h, w)             # (only the uncrop entries 18..21 are read: cube size is moot)
        crop = torch.cat((z_pred_depth, z_pred_mask_logits), dim=1)
        dl, ov, iou, bce = _PoseLossTerms.apply(crop, coefs, target.depth.reshape(-1).float().contiguous(),
                                                target.mask.reshape(-1).float().contiguous(), int(H), int(W))
        out = {'ov_depth': ov, 'depth': dl, 'iou': iou, 'mask': bce}
    else:
        out = _pose_loss_expressions(target, z_pred_depth, z_pred_mask_logits, z_pred_camera)
    if z_pred_latent is not None and z_target_latent is not None:
        zp = z_pred_latent.reshape(z_pred_latent.shape[0], -1)
        zt = z_target_latent.reshape(z_target_latent.shape[0], -1)
        out['latent'] = cosine_distance(zp, zt.expand_as(zp))
    return out


def _pose_loss_expressions(target, z_pred_depth, z_pred_mask_logits, z_pred_camera):
    """The reference's expressions, term by term (host tensors; also the yardstick of the fused form in tests)."""
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
    return out


def weigh_losses(loss_dict, weight_dict):
    return {k: weight_dict.get(k, 0.0) * v for k, v in loss_dict.items()}
