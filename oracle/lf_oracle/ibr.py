"""ORACLE (test infrastructure, not product): image-based rendering on CPU
(restates latentfusion/ibr.py:11-222 with the oracle camera record)."""
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
