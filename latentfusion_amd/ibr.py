"""Image-based rendering of colour (API mirror of latentfusion/ibr.py:11-222): depth-based
reprojection of the input views into the output views and a camera-distance softmax blend.
Colour only -- not part of the pose loop; the depth maps come from the HIP
renderer, the 2-D warps are served by the device grid sampler.
"""
import math

import torch

from . import image_ops, three
from .three.batchview import b2bv, bv2b


def outer_distance(x1, x2, metric='cosine', eps=1e-8):
    """Pairwise distance matrix (reference distances.py:27-42; cosine only is used here)."""
    if metric != 'cosine':
        raise ValueError(f'Unknown type {metric!r}')
    w1 = x1.norm(dim=1, keepdim=True)
    w2 = x2.norm(dim=1, keepdim=True)
    return 1.0 - (x1 @ x2.t()) / (w1 @ w2.t()).clamp(min=eps)


def pixel_coords_uv(camera, size):
    """Viewport pixel lattice (reference modules/geometry.py:495-513)."""
    h, w = size
    dev = camera.device
    v, u = torch.meshgrid(torch.linspace(0.0, 1.0, h, device=dev), torch.linspace(0.0, 1.0, w, device=dev), indexing='ij')
    u = u.unsqueeze(0) * camera.viewport_width.view(-1, 1, 1) + camera.viewport[:, 0].view(-1, 1, 1)
    v = v.unsqueeze(0) * camera.viewport_height.view(-1, 1, 1) + camera.viewport[:, 1].view(-1, 1, 1)
    return u, v


def depth_camera_coords(camera, depth):
    u, v = pixel_coords_uv(camera, depth.shape[-2:])
    z = depth.view_as(u)
    x = (u - camera.u0.view(-1, 1, 1)) / camera.fu.view(-1, 1, 1) * z
    y = (v - camera.v0.view(-1, 1, 1)) / camera.fv.view(-1, 1, 1) * z
    return x, y, z


def depth_object_coords(camera, depth):
    x, y, z = depth_camera_coords(camera, depth)
    grid = torch.stack((x, y, z), dim=-1)
    return three.transform_coords(three.grid_to_coords(grid), camera.cam_to_obj).view_as(grid)


def depth_to_warp_field(source_cam, target_cam, target_depth):
    """Sampling grid that pulls source-view pixels into the target views, (V_o, V_i, H, W, 2)
    (reference ibr.py:11-49)."""
    h, w = target_depth.shape[-2:]
    x, y, z = depth_camera_coords(target_cam, target_cam.denormalize_depth(target_depth))
    cam_pts = three.grid_to_coords(torch.stack((x, y, z), dim=-1))
    obj_pts = three.transform_coords(cam_pts, target_cam.cam_to_obj)
    obj_pts = bv2b(obj_pts[:, None].expand(-1, len(source_cam), -1, -1))
    to_pix = bv2b(source_cam.obj_to_image[None].expand(len(target_cam), -1, -1, -1))
    pix = three.transform_coords(obj_pts, to_pix)
    vp = source_cam.viewport.repeat(len(target_cam), 1)
    gx = ((pix[..., 0] - vp[:, 0, None]) / (vp[:, 2] - vp[:, 0])[:, None]) * 2 - 1
    gy = ((pix[..., 1] - vp[:, 1, None]) / (vp[:, 3] - vp[:, 1])[:, None]) * 2 - 1
    return torch.stack((gx, gy), dim=-1).view(len(target_cam), len(source_cam), h, w, 2)


def reproject_views(image_in, depth_in, depth_out, camera_in, camera_out):
    """(V_i,C,H,W) inputs -> (V_o,V_i,C,H,W) reprojected images and depths (reference ibr.py:52-93)."""
    grid = bv2b(depth_to_warp_field(camera_in, camera_out, depth_out))
    v_i, v_o = len(camera_in), len(camera_out)
    image = bv2b(image_in.unsqueeze(0).expand(v_o, -1, -1, -1, -1))
    obj = depth_object_coords(camera_in, depth_in)
    obj = bv2b(obj.unsqueeze(0).expand(v_o, -1, -1, -1, -1))
    cam_rep = camera_out.repeat_interleave(v_i)
    depth_tf = three.transform_coord_grid(obj, cam_rep.obj_to_cam)[..., 2].unsqueeze(1)
    depth_tf = cam_rep.normalize_depth(depth_tf)
    image_re = image_ops._sample(image, grid, 'bilinear', 'zeros')
    depth_re = image_ops._sample(depth_tf, grid, 'bilinear', 'zeros')
    return b2bv(image_re, v_i), b2bv(depth_re, v_i)


def reproject_views_batch(image_in, depth_in, depth_out, camera_in, camera_out):
    """reproject_views per object of a batch (reference ibr.py:96-138).

    image_in (B,V_i,C,H,W), depth_in (B,V_i,1,H,W), depth_out (B,V_o,1,H,W); cameras flat, object-major.
    Returns reprojected images and depths (B,V_o,V_i,.,H,W) and the rotation / position distances between every
    output and input camera, each (B,V_o,V_i): angular distance / pi (acos clamp 1e-2) and cosine distance / 2."""
    nb, n_in, n_out = image_in.shape[0], image_in.shape[1], depth_out.shape[1]
    imgs, deps, dist_r, dist_t = [], [], [], []
    for i in range(nb):
        cin, cout = camera_in[i * n_in:(i + 1) * n_in], camera_out[i * n_out:(i + 1) * n_out]
        dist_r.append(three.quaternion.angular_distance(cout.quaternion, cin.quaternion, eps=1e-2) / math.pi)
        dist_t.append(outer_distance(cout.position, cin.position, metric='cosine') / 2.0)
        img_re, dep_re = reproject_views(image_in[i], depth_in[i], depth_out[i], cin, cout)
        imgs.append(img_re)
        deps.append(dep_re)
    return torch.stack(imgs, dim=0), torch.stack(deps, dim=0), torch.stack(dist_r, dim=0), torch.stack(dist_t, dim=0)


def render_ibr(camera_in, camera_out, image_in, depth_fake_in, depth_fake_out, p=0.5, weight_type='cam_dist', eps=1e-2):
    """Blend of the reprojected input views, weights softmax(1 / clamp(d^p, eps)) over the inputs
    (reference ibr.py:181-222; weight types cam_dist / cam_angle / cam_hybrid / depth)."""
    outs, reprojs = [], []
    nb = image_in.shape[0]
    n_in, n_out = len(camera_in) // nb, len(camera_out) // nb
    for i in range(nb):
        cin, cout = camera_in[i * n_in:(i + 1) * n_in], camera_out[i * n_out:(i + 1) * n_out]
        img_re, depth_re = reproject_views(image_in[i], depth_fake_in[i], depth_fake_out[i], cin, cout)
        reprojs.append(img_re)
        if weight_type == 'depth':
            diff = (depth_re - depth_fake_out[i].unsqueeze(1).expand_as(depth_re)).abs()
            wgt = torch.softmax(1.0 / ((diff / diff.max()) ** p + eps), dim=1).squeeze(2)
        else:
            if weight_type == 'cam_dist':
                d = outer_distance(cout.position, cin.position, metric='cosine', eps=eps) / 2.0
            elif weight_type == 'cam_angle':
                d = three.quaternion.angular_distance(cout.quaternion, cin.quaternion) / math.pi
            elif weight_type == 'cam_hybrid':
                d_t = outer_distance(cout.position, cin.position, metric='cosine') / 2.0
                d_r = (three.quaternion.angular_distance(cout.quaternion, cin.quaternion) / (math.pi / 8)).clamp(0.0, 1.0)
                d = 1.0 - (1.0 - d_t) * (1.0 - d_r)
            else:
                raise ValueError(f'Unknown weight_type {weight_type}')
            wgt = torch.softmax(1.0 / (d[..., None, None] ** p).clamp(min=eps), dim=1)
        outs.append((wgt.unsqueeze(2) * img_re).sum(dim=1))
    return torch.stack(outs, dim=0), torch.stack(reprojs, dim=0)


def render_latent_ibr(photographer, z_obj, camera_in, camera_out, image_in, p=0.5, weight_type='cam_dist', eps=0.0001):
    """Depths of the input and output views from the latent renderer, colour by IBR; differentiable, returns
    (colour, depth_out, mask_out, reprojections) (reference ibr.py:141-154)."""
    fake_in, _, _ = photographer.decode(z_obj, camera_in)
    fake_out, _, _ = photographer.decode(z_obj, camera_out)
    color, reproj = render_ibr(camera_in, camera_out, image_in, fake_in['depth'], fake_out['depth'], p, weight_type, eps)
    return color, fake_out['depth'], fake_out['mask'], reproj


def blend_logits(logits, image_reproj):
    """Per-pixel softmax blend of the reprojected views: logits (B,V_i,H,W), image_reproj (B,V_i,C,H,W)
    (reference ibr.py:225-228)."""
    weights = torch.softmax(logits, dim=1).unsqueeze(2)
    return (weights * image_reproj).sum(dim=1), weights


def warp_blend_logits(logits, image_reproj, flow_size):
    """Softmax blend + a learned residual flow of at most `flow_size` pixels per view: logits (B,3*V_i,H,W) =
    [blend | flow x | flow y], image_reproj (B,V_i,C,H,W).  Returns (image, weights, flow_dx, flow_dy)
    (reference ibr.py:231-249: identity lattice linspace(-1,1), tanh-bounded offsets, grid clamped to [-1,1],
    bilinear sampling with zeros padding -- lf_grid_sample2d_fwd on the device)."""
    dev = image_reproj.device
    v_i = image_reproj.shape[1]
    h, w = image_reproj.shape[-2:]
    blend, fx_logits, fy_logits = torch.split(logits, v_i, dim=1)
    weights = torch.softmax(blend, dim=1).unsqueeze(2)
    flow_dx = flow_size / w * torch.tanh(fx_logits)
    flow_dy = flow_size / h * torch.tanh(fy_logits)
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, h, device=dev), torch.linspace(-1, 1, w, device=dev), indexing='ij')
    grid = torch.stack((gx[None, None].expand_as(flow_dx) + flow_dx, gy[None, None].expand_as(flow_dy) + flow_dy),
                       dim=-1).clamp(-1, 1)
    warped = b2bv(image_ops._sample(bv2b(image_reproj).contiguous(), bv2b(grid).contiguous(), 'bilinear', 'zeros'), v_i)
    return (weights * warped).sum(dim=1), weights, flow_dx, flow_dy


def render_latent_ibr2(photographer, z_obj, camera_in, camera_out, image_in, p=0.5, weight_type='cam_dist',
                       return_latent=True, eps=0.0001, apply_mask=False):
    """Depth from the latent renderer for input and output views, colour by IBR (reference ibr.py:157-178)."""
    with torch.no_grad():
        y_in, _, _ = photographer.decode(z_obj, camera_in, apply_mask=apply_mask)
        y_out, z_out, _ = photographer.decode(z_obj, camera_out, return_latent=return_latent, apply_mask=apply_mask)
    color, _ = render_ibr(camera_in, camera_out, image_in, y_in['depth'], y_out['depth'], p, weight_type, eps)
    y_out['color'] = color * (y_out['mask'] > 0.5) if apply_mask else color
    return y_out, z_out
