"""ORACLE (test infrastructure, not product): pinhole camera algebra on CPU tensors.

Restates latentfusion/modules/geometry.py:46-590 (class Camera) as a plain record + free
functions.  State: K (B,3,4), viewport (B,4)=(xmin,ymin,xmax,ymax), log_q (B,3), t (B,3),
scalars z_span / width / height.  Everything is differentiable w.r.t. log_q, t, viewport.
"""
import torch
import torch.nn.functional as F

from . import quat


class Cam:
    __slots__ = ('K', 'viewport', 'log_q', 't', 'z_span', 'width', 'height')

    def __init__(self, K, log_q, t, viewport=None, z_span=0.5, width=640, height=480):
        if K.dim() == 2:
            K = K.unsqueeze(0)
        if K.shape[-1] == 3:                                   # three/rigid.py:10-20
            K = torch.cat((K, torch.zeros(K.shape[0], 3, 1, dtype=K.dtype)), dim=-1)
        if log_q.dim() == 1:
            log_q = log_q.unsqueeze(0)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        if viewport is None:                                   # geometry.py:60-63
            viewport = torch.tensor((0, 0, width, height), dtype=torch.float32).view(1, 4) \
                .expand(K.shape[0], -1)
        if viewport.dim() == 1:
            viewport = viewport.unsqueeze(0)
        self.K, self.log_q, self.t, self.viewport = K, log_q, t, viewport
        self.z_span, self.width, self.height = z_span, width, height

    @classmethod
    def from_extrinsic(cls, K, E, **kw):
        """geometry.py:80-87: log_q = qlog(mat_to_quat(R))[1:], t = E[:3,3]."""
        if E.dim() == 2:
            E = E.unsqueeze(0)
        q = quat.from_matrix(E[:, :3, :3].contiguous())
        return cls(K, quat.qlog(q)[:, 1:], E[:, :3, 3].contiguous(), **kw)

    def __len__(self):
        return self.K.shape[0]

    def like(self, **kw):
        d = dict(K=self.K, log_q=self.log_q, t=self.t, viewport=self.viewport,
                 z_span=self.z_span, width=self.width, height=self.height)
        d.update(kw)
        return Cam(**d)

    def __getitem__(self, idx):
        if isinstance(idx, int):
            idx = slice(idx, idx + 1)
        return self.like(K=self.K[idx], log_q=self.log_q[idx], t=self.t[idx], viewport=self.viewport[idx])

    def clone(self):
        return self.like(K=self.K.clone(), log_q=self.log_q.clone(), t=self.t.clone(),
                         viewport=self.viewport.clone())

    def detach(self):
        return self.like(K=self.K.detach(), log_q=self.log_q.detach(), t=self.t.detach(),
                         viewport=self.viewport.detach())

    def repeat(self, n):
        return self.like(K=self.K.repeat(n, 1, 1), log_q=self.log_q.repeat(n, 1), t=self.t.repeat(n, 1),
                         viewport=self.viewport.repeat(n, 1))

    # --- derived quantities --------------------------------------------------------------
    @property
    def quaternion(self):                                      # geometry.py:106-108
        return quat.qexp(self.log_q)

    @property
    def R(self):                                               # geometry.py:147-153 (3x3 part)
        return quat.to_matrix(quat.unit(self.quaternion))

    @property
    def fu(self):
        return self.K[:, 0, 0]

    @property
    def fv(self):
        return self.K[:, 1, 1]

    @property
    def u0(self):
        return self.K[:, 0, 2]

    @property
    def v0(self):
        return self.K[:, 1, 2]

    @property
    def vw(self):
        return self.viewport[:, 2] - self.viewport[:, 0]

    @property
    def vh(self):
        return self.viewport[:, 3] - self.viewport[:, 1]

    @property
    def znear(self):                                           # geometry.py:249-251
        return self.t[:, 2] - self.z_span

    @property
    def zfar(self):                                            # geometry.py:253-255
        return self.t[:, 2] + self.z_span

    def _R4(self):                                             # geometry.py:147-153
        R = F.pad(self.R, (0, 1, 0, 1))
        R[:, -1, -1] = 1.0
        return R

    def _T4(self, sign=1.0):                                   # geometry.py:155-163
        return F.pad((sign * self.t).unsqueeze(2), (3, 0, 0, 1)) + torch.eye(4)

    @property
    def obj_to_cam(self):                                      # geometry.py:207-209  T @ R
        return self._T4() @ self._R4()

    @property
    def cam_to_obj(self):                                      # geometry.py:211-213  R^T @ T^-1
        return self._R4().transpose(2, 1) @ self._T4(-1.0)

    @property
    def position(self):                                        # geometry.py:219-224  -R^T t
        return -(self.R.transpose(1, 2) @ self.t.unsqueeze(2)).squeeze(2)

    # --- depth range mapping -------------------------------------------------------------
    def normalize_depth(self, depth, eps=0.01):                # geometry.py:560-565
        zn = (self.znear - eps).view(-1, 1, 1, 1)
        zf = (self.zfar + eps).view(-1, 1, 1, 1)
        return ((depth - zn) / (zf - zn)).clamp(0, 1) * 2.0 - 1.0

    def denormalize_depth(self, depth, eps=0.01):              # geometry.py:555-558
        zn = (self.znear - eps).view(*depth.shape[:-3], 1, 1, 1)
        zf = (self.zfar + eps).view(*depth.shape[:-3], 1, 1, 1)
        return (depth / 2.0 + 0.5) * (zf - zn) + zn

    # --- image <-> viewport resampling ---------------------------------------------------
    def zoom_boxes(self, target_size, target_dist, image_scale=1.0):
        """New viewports of the zoomed camera (geometry.py:294-339).  Note Q5: symmetric in
        (target_size, target_dist)."""
        zs = self.t[:, 2]
        bu = target_dist * (1.0 / zs) / self.fu * self.fu * target_size / self.width * image_scale
        bv = target_dist * (1.0 / zs) / self.fv * self.fv * target_size / self.height * image_scale
        origin = torch.tensor((0.0, 0.0, 0.0, 1.0)).view(1, 4, 1).expand(len(self), -1, -1)
        uvw = self.K @ self.obj_to_cam @ origin
        uv = (uvw[:, :2] / uvw[:, 2, None]).squeeze(2)
        cu, cv = uv[:, 0] / self.width, uv[:, 1] / self.height
        return torch.stack(((cu - bu / 2) * float(self.width), (cv - bv / 2) * float(self.height),
                            (cu + bu / 2) * float(self.width), (cv + bv / 2) * float(self.height)), dim=1)

    def zoom(self, image, target_size, target_dist, scale_mode='bilinear'):
        boxes = self.zoom_boxes(target_size, target_dist)
        cam = self.like(viewport=boxes)
        if image is None:
            return cam
        return crop_boxes(image, boxes, self.height, self.width, target_size, scale_mode), cam

    def uncrop(self, image=None, scale_mode='nearest'):
        """Paste a viewport crop back into the full frame (geometry.py:261-285, A3)."""
        cam = self.like(viewport=None)
        if image is None:
            return cam
        yy, xx = torch.meshgrid(torch.arange(0, self.height, dtype=torch.float32),
                                torch.arange(0, self.width, dtype=torch.float32), indexing='ij')
        yy = (yy.unsqueeze(0) - self.viewport[:, 1, None, None]) / self.vh[:, None, None] * 2 - 1
        xx = (xx.unsqueeze(0) - self.viewport[:, 0, None, None]) / self.vw[:, None, None] * 2 - 1
        grid = torch.stack((xx, yy), dim=-1)
        out = F.grid_sample(image.float(), grid.float(), mode=scale_mode, padding_mode='border',
                            align_corners=False)
        return out, cam


def crop_boxes(image, boxes, in_h, in_w, out_size, scale_mode):
    """grid_sample crop with zeros padding (geometry.py:20-44,350-352).

    Quirk Q15: the reference's TorchScript `bbox_to_grid` compiles `xmin / w` as
    Int(xmin) / Int(w), i.e. the box corners are TRUNCATED TOWARD ZERO to integers before the
    crop grid is built (the camera keeps the un-truncated float viewport)."""
    grids = []
    for b in boxes:
        x0, y0, x1, y1 = (int(v) for v in b.tolist())
        gy = torch.linspace(y0 / int(in_h), y1 / int(in_h), out_size) * 2 - 1
        gx = torch.linspace(x0 / int(in_w), x1 / int(in_w), out_size) * 2 - 1
        yy, xx = torch.meshgrid(gy, gx, indexing='ij')
        grids.append(torch.stack((xx, yy), dim=-1))
    return F.grid_sample(image.float(), torch.stack(grids, 0), mode=scale_mode, align_corners=False)


def cat(cams):
    c0 = cams[0]
    return c0.like(K=torch.cat([c.K for c in cams]), log_q=torch.cat([c.log_q for c in cams]),
                   t=torch.cat([c.t for c in cams]), viewport=torch.cat([c.viewport for c in cams]))


def rotate(cam, q):
    """cam.quaternion <- cam.quaternion * q  (geometry.py:235-237; setter :110-112)."""
    new_q = quat.qmul(cam.quaternion, q)
    return cam.like(log_q=quat.qlog(new_q)[:, 1:])


def flip(cam, axis):
    """pose/utils.py:74-78."""
    ax = torch.tensor([axis], dtype=torch.float32).expand(len(cam), -1)
    return rotate(cam, quat.axis_angle(ax, 3.141592653589793))
