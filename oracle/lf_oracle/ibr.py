"""ORACLE (test infrastructure, not product): image-based rendering on CPU
(restates latentfusion/ibr.py:11-222 with the oracle camera record)."""
import math

import torch
import torch.nn.functional as F

from . import nets
from .camera import Cam


def _uv(cam, h, w):
    v, u = torch.meshgrid(torch.linspace(0.0, 1.0, h), torch.linspace(0.0, 1.0, w), indexing='ij')
    u = u.unsqueeze(0) * cam.vw.view(-1, 1, 1) + cam.viewport[:, 0].view(-1, 1, 1)
    v = v.unsqueeze(0) * cam.vh.view(-1, 1, 1) + cam.viewport[:, 1].view(-1, 1, 1)
    return u, v


def _cam_points(cam, depth):
    """(B,1,H,W) metric depth -> (B, H*W, 3) camera-space points (geometry.py:533-545)."""
    h, w = depth.shape[-2:]
    u, v = _uv(cam, h, w)
    z = depth.view_as(u)
    x = (u - cam.u0.view(-1, 1, 1)) / cam.fu.view(-1, 1, 1) * z
    y = (v - cam.v0.view(-1, 1, 1)) / cam.fv.view(-1, 1, 1) * z
    return torch.stack((x, y, z), dim=-1).view(len(cam), -1, 3)


def _apply(T, pts):
    hom = torch.cat((pts, torch.ones_like(pts[..., :1])), dim=-1)
    out = (T @ hom.transpose(1, 2)).transpose(1, 2)
    return out[..., :-1] / out[..., -1:]


def reproject_views(image_in, depth_in, depth_out, cam_in, cam_out):
    """ibr.py:52-93.  image_in (Vi,C,H,W), depth_* normalised (V,1,H,W)."""
    vi, vo = len(cam_in), len(cam_out)
    h, w = depth_out.shape[-2:]
    obj = _apply(cam_out.cam_to_obj, _cam_points(cam_out, cam_out.denormalize_depth(depth_out)))     # (Vo,HW,3)
    obj = obj[:, None].expand(-1, vi, -1, -1).reshape(vo * vi, h * w, 3)
    P = (cam_in.K @ cam_in.obj_to_cam)[None].expand(vo, -1, -1, -1).reshape(vo * vi, 3, 4)
    pix = _apply(P, obj)
    vp = cam_in.viewport.repeat(vo, 1)
    gx = ((pix[..., 0] - vp[:, 0, None]) / (vp[:, 2] - vp[:, 0])[:, None]) * 2 - 1
    gy = ((pix[..., 1] - vp[:, 1, None]) / (vp[:, 3] - vp[:, 1])[:, None]) * 2 - 1
    grid = torch.stack((gx, gy), dim=-1).view(vo * vi, h, w, 2)
    img = image_in[None].expand(vo, -1, -1, -1, -1).reshape(vo * vi, *image_in.shape[1:])
    # NOTE: the reference feeds the NORMALISED input depth to depth_object_coords (ibr.py:73)
    obj_in = _apply(cam_in.cam_to_obj, _cam_points(cam_in, depth_in)).view(vi, h, w, 3)
    obj_in = obj_in[None].expand(vo, -1, -1, -1, -1).reshape(vo * vi, h * w, 3)
    rep = Cam(cam_out.K.repeat_interleave(vi, 0), cam_out.log_q.repeat_interleave(vi, 0), cam_out.t.repeat_interleave(vi, 0),
              viewport=cam_out.viewport.repeat_interleave(vi, 0))
    z_tf = _apply(rep.obj_to_cam, obj_in)[..., 2].view(vo * vi, 1, h, w)
    z_tf = rep.normalize_depth(z_tf)
    img_re = F.grid_sample(img, grid, mode='bilinear', align_corners=False)
    dep_re = F.grid_sample(z_tf, grid, mode='bilinear', align_corners=False)
    return img_re.view(vo, vi, *img_re.shape[1:]), dep_re.view(vo, vi, *dep_re.shape[1:])


def render_latent_ibr2(pck, z_obj, cam_in, cam_out, image_in, p=0.5, eps=1e-4, apply_mask=False):
    """ibr.py:157-222 (weight_type 'cam_dist')."""
    y_in, _, _ = nets.decode(pck, z_obj, cam_in, apply_mask=apply_mask)
    y_out, lat, _ = nets.decode(pck, z_obj, cam_out, apply_mask=apply_mask)
    img_re, _ = reproject_views(image_in[0], y_in['depth'][0], y_out['depth'][0], cam_in, cam_out)
    a, b = cam_out.position, cam_in.position
    d = (1.0 - (a @ b.t()) / (a.norm(dim=1, keepdim=True) @ b.norm(dim=1, keepdim=True).t()).clamp(min=eps)) / 2.0
    wgt = torch.softmax(1.0 / (d[..., None, None] ** p).clamp(min=eps), dim=1)
    color = (wgt.unsqueeze(2) * img_re).sum(dim=1).unsqueeze(0)
    if apply_mask:
        color = color * (y_out['mask'] > 0.5)
    y_out['color'] = color
    return y_out, lat


def reproject_views_batch(image_in, depth_in, depth_out, cam_in, cam_out):
    """ibr.py:96-138.  image_in (B,Vi,C,H,W), depth_in (B,Vi,1,H,W), depth_out (B,Vo,1,H,W); cameras object-major."""
    from . import quat
    nb, vi, vo = image_in.shape[0], image_in.shape[1], depth_out.shape[1]
    imgs, deps, dr, dt = [], [], [], []
    for i in range(nb):
        ci, co = cam_in[i * vi:(i + 1) * vi], cam_out[i * vo:(i + 1) * vo]
        dr.append(quat.angular_distance(co.quaternion, ci.quaternion, eps=1e-2) / math.pi)
        a, b = co.position, ci.position
        dt.append((1.0 - (a @ b.t()) / (a.norm(dim=1, keepdim=True) @ b.norm(dim=1, keepdim=True).t()).clamp(min=1e-8)) / 2.0)
        im, de = reproject_views(image_in[i], depth_in[i], depth_out[i], ci, co)
        imgs.append(im)
        deps.append(de)
    return torch.stack(imgs), torch.stack(deps), torch.stack(dr), torch.stack(dt)


def blend_logits(logits, image_reproj):
    """ibr.py:225-228."""
    w = torch.softmax(logits, dim=1).unsqueeze(2)
    return (w * image_reproj).sum(dim=1), w


def warp_blend_logits(logits, image_reproj, flow_size):
    """ibr.py:231-249."""
    vi = image_reproj.shape[1]
    h, w = image_reproj.shape[-2:]
    bl, fx, fy = torch.split(logits, vi, dim=1)
    wgt = torch.softmax(bl, dim=1).unsqueeze(2)
    dx = flow_size / w * torch.tanh(fx)
    dy = flow_size / h * torch.tanh(fy)
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing='ij')
    grid = torch.stack((gx[None, None] + dx, gy[None, None] + dy), dim=-1).clamp(-1, 1)
    flat = F.grid_sample(image_reproj.reshape(-1, *image_reproj.shape[2:]), grid.reshape(-1, h, w, 2), mode='bilinear',
                         align_corners=False)
    img = flat.view(-1, vi, *flat.shape[1:])
    return (wgt * img).sum(dim=1), wgt, dx, dy


def render_latent_ibr(pck, z_obj, cam_in, cam_out, image_in, p=0.5, eps=1e-4):
    """ibr.py:141-154 (weight_type 'cam_dist'): (colour, depth_out, mask_out, reprojections)."""
    y_in, _, _ = nets.decode(pck, z_obj, cam_in)
    y_out, _, _ = nets.decode(pck, z_obj, cam_out)
    img_re, _ = reproject_views(image_in[0], y_in['depth'][0], y_out['depth'][0], cam_in, cam_out)
    a, b = cam_out.position, cam_in.position
    d = (1.0 - (a @ b.t()) / (a.norm(dim=1, keepdim=True) @ b.norm(dim=1, keepdim=True).t()).clamp(min=eps)) / 2.0
    wgt = torch.softmax(1.0 / (d[..., None, None] ** p).clamp(min=eps), dim=1)
    return (wgt.unsqueeze(2) * img_re).sum(dim=1).unsqueeze(0), y_out['depth'], y_out['mask'], img_re.unsqueeze(0)


def render_ibr_generator(pck, gck, z_obj, color_in, cam_in, cam_out):
    """LatentFusionModel.render_ibr on PREPROCESSED inputs (recon/inference.py:151-217): the generator U-Net `gck`
    sees [depth_out | per input view: reprojected colour (3), reprojected depth (1), camera similarity (1)]."""
    y_in, _, _ = nets.decode(pck, z_obj, cam_in)
    y_out, lat, _ = nets.decode(pck, z_obj, cam_out)
    mask_out, depth_out = y_out['mask'], y_out['depth']
    img_re, dep_re, _dr, dt = reproject_views_batch(color_in.unsqueeze(0), y_in['depth'], y_out['depth'], cam_in, cam_out)
    img_re = (img_re * mask_out.unsqueeze(2)).flatten(0, 1)
    dep_re = ((dep_re + 1.0) * mask_out.unsqueeze(2) - 1.0).flatten(0, 1)
    sims = 1.0 - dt.flatten(0, 1) * 2
    x = torch.cat((img_re, dep_re, sims[:, :, None, None, None].expand(-1, -1, -1, *img_re.shape[-2:])), dim=2)
    x = x.reshape(-1, x.shape[1] * x.shape[2], x.shape[3], x.shape[4])
    x = torch.cat((depth_out.flatten(0, 1), x), dim=1)
    ga = gck['args']
    gsd = {'g.' + k: v for k, v in gck['state_dict'].items()}
    logits = nets.unet(x, gsd, 'g', ga['block_config'], ga['in_channels'], ga['out_channels'])
    color, wgt, dx, dy = warp_blend_logits(logits, img_re, 5)
    out = dict(y_out)
    out['color'] = color
    return {k: v.squeeze(0) for k, v in out.items()}, lat.squeeze(0), {'logits': logits, 'image_reproj': img_re,
                                                                       'depth_reproj': dep_re, 'flow_dx': dx, 'flow_dy': dy}
