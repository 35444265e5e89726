"""Quaternion algebra in the (w, x, y, z) convention -- host-side math of the camera model.

API mirror of the reference's latentfusion/three/quaternion.py (same function names and
argument meaning) so callers can switch imports; the bodies are independent closed forms.
All functions are differentiable torch code and run on whatever device their inputs live on
(these are (B,4)-sized tensors: plumbing, not the hot path).
"""
import math

import torch


def identity(n, device='cpu'):
    q = torch.zeros(n, 4, device=device)
    q[:, 0] = 1.0
    return q


def normalize(quaternion, eps=1e-12):
    """Unit quaternion; divides by max(|q|, eps)  (reference :15-36)."""
    if quaternion.shape[-1] != 4:
        raise ValueError(f'Input must be a tensor of shape (*, 4). Got {tuple(quaternion.shape)}')
    return quaternion / quaternion.norm(dim=-1, keepdim=True).clamp(min=eps)


def quat_to_mat(quaternion):
    """(*,4) -> (*,3,3) rotation matrix of the normalised quaternion  (reference :39-93)."""
    squeeze = quaternion.dim() == 1
    q = normalize(quaternion.reshape(-1, 4))
    w, x, y, z = q.unbind(-1)
    x2, y2, z2 = 2.0 * x, 2.0 * y, 2.0 * z
    m = torch.stack((
        1.0 - (y2 * y + z2 * z), y2 * x - z2 * w, z2 * x + y2 * w,
        y2 * x + z2 * w, 1.0 - (x2 * x + z2 * z), z2 * y - x2 * w,
        z2 * x - y2 * w, z2 * y + x2 * w, 1.0 - (x2 * x + y2 * y)), dim=-1).view(-1, 3, 3)
    return m[0] if squeeze else m


def mat_to_quat(rotation_matrix, eps=1e-8):
    """(*,3,3) -> (*,4); numerically safe branch selection on the trace  (reference :96-176)."""
    squeeze = rotation_matrix.dim() == 2
    m = rotation_matrix.reshape(-1, 3, 3)
    tiny = torch.finfo(m.dtype).tiny
    d0, d1, d2 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    a, b, c = m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1]
    p, q, r = m[:, 0, 1] + m[:, 1, 0], m[:, 0, 2] + m[:, 2, 0], m[:, 1, 2] + m[:, 2, 1]

    def div(num, den):
        return num / den.clamp(min=tiny)
    s = torch.sqrt(d0 + d1 + d2 + 1.0) * 2.0
    cand_t = torch.stack((0.25 * s, div(a, s), div(b, s), div(c, s)), -1)
    s = torch.sqrt(1.0 + d0 - d1 - d2 + eps) * 2.0
    cand_x = torch.stack((div(a, s), 0.25 * s, div(p, s), div(q, s)), -1)
    s = torch.sqrt(1.0 + d1 - d0 - d2 + eps) * 2.0
    cand_y = torch.stack((div(b, s), div(p, s), 0.25 * s, div(r, s)), -1)
    s = torch.sqrt(1.0 + d2 - d0 - d1 + eps) * 2.0
    cand_z = torch.stack((div(c, s), div(q, s), div(r, s), 0.25 * s), -1)
    out = torch.where((d1 > d2).unsqueeze(-1), cand_y, cand_z)
    out = torch.where(((d0 > d1) & (d0 > d2)).unsqueeze(-1), cand_x, out)
    out = torch.where((d0 + d1 + d2 > 0.0).unsqueeze(-1), cand_t, out)
    return out[0] if squeeze else out


def random(k=1, device='cpu'):
    """Uniform random unit quaternions (reference :179-195; same RNG consumption)."""
    u = torch.rand(k, 3, device=device)
    a, b = torch.sqrt(1.0 - u[:, 0]), torch.sqrt(u[:, 0])
    t1, t2 = 2.0 * math.pi * u[:, 1], 2.0 * math.pi * u[:, 2]
    return torch.stack((torch.cos(t2) * b, torch.sin(t1) * a, torch.cos(t1) * a, torch.sin(t2) * b), dim=1)


def qmul(q1, q2):
    """Hamilton product q1 * q2 (reference :198-218)."""
    assert q1.shape[-1] == 4 and q2.shape[-1] == 4
    shape = q1.shape
    a, b = q1.reshape(-1, 4), q2.reshape(-1, 4)
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    out = torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                       aw * bx + ax * bw + ay * bz - az * by,
                       aw * by - ax * bz + ay * bw + az * bx,
                       aw * bz + ax * by - ay * bx + az * bw), dim=1)
    return out.view(shape)


def rotate_vector(quat, vector):
    """v + 2 (w (u x v) + u x (u x v)), u = vector part (reference :221-238)."""
    shape = vector.shape
    q, v = quat.reshape(-1, 4), vector.reshape(-1, 3)
    u = q[:, 1:]
    uv = torch.cross(u, v, dim=1)
    return (v + 2 * (q[:, :1] * uv + torch.cross(u, uv, dim=1))).view(shape)


def from_axis_angle(axis, angle):
    """(B,3) axis, angle (B,) or float -> (B,4)  (reference :254-284)."""
    if torch.is_tensor(axis) and isinstance(angle, float):
        angle = torch.full((axis.shape[0],), angle, dtype=axis.dtype, device=axis.device)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    half = angle / 2.0
    return torch.cat((torch.cos(half).unsqueeze(-1), torch.sin(half).unsqueeze(-1) * axis), dim=-1)


def qexp(q, eps=1e-8):
    """Quaternion exponential; (B,3) input is a pure quaternion (reference :287-311).
    The rotation angle is clamped at eps only inside the division."""
    if q.shape[1] == 4:
        s, v = q[:, :1], q[:, 1:]
    else:
        s, v = torch.zeros_like(q[:, :1]), q
    theta = v.norm(dim=-1, keepdim=True)
    return torch.exp(s) * torch.cat((torch.cos(theta), 1.0 / theta.clamp(min=eps) * torch.sin(theta) * v), dim=-1)


def qlog(q, eps=1e-8):
    """Quaternion logarithm -> (B,4)  (reference :314-334)."""
    mag = q.norm(dim=-1, keepdim=True)
    s, v = q[..., :1], q[..., 1:]
    ang = torch.acos((s / mag.clamp(min=eps)).clamp(-1.0 + 1e-7, 1.0 - 1e-7))
    return torch.cat((torch.log(mag), v / v.norm(dim=-1, keepdim=True).clamp(min=eps) * ang), dim=-1)


def qdelta(n, std, device=None):
    omega = torch.cat((torch.zeros(n, 1, device=device), torch.randn(n, 3, device=device)), dim=-1)
    return qexp(std / 2.0 * omega)


def perturb(q, std):
    squeeze = q.dim() == 1
    qq = q.unsqueeze(0) if squeeze else q
    out = qmul(qdelta(qq.shape[0], std, device=qq.device), qq)
    return out.squeeze(0) if squeeze else out


def angular_distance(q1, q2, eps=1e-7):
    """Pairwise geodesic distance 2 acos|<q1,q2>| -> (N1,N2)  (reference :372-377)."""
    d = (normalize(q1) @ normalize(q2).t()).abs().clamp(-1.0 + eps, 1.0 - eps)
    return 2 * torch.acos(d)


def from_spherical(theta, phi, r=1.0):
    """Pure quaternion of a spherical direction (reference quaternion.py:248-254; only z is scaled by r)."""
    x = torch.cos(theta) * torch.sin(phi)
    y = torch.sin(theta) * torch.sin(phi)
    z = r * torch.cos(phi)
    return torch.stack((torch.zeros_like(x), x, y, z), dim=-1)
