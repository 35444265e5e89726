"""ORACLE (test infrastructure, not product): quaternion / rigid algebra on CPU tensors.

Restates latentfusion/three/quaternion.py and three/rigid.py of the reference in closed form
(w, x, y, z convention).  Each function cites the reference lines it follows.
"""
import math

import torch


def unit(q, eps=1e-12):
    """q / max(|q|, eps)   (three/quaternion.py:15-36, F.normalize semantics)."""
    return q / q.norm(dim=-1, keepdim=True).clamp(min=eps)


def qexp(v, eps=1e-8):
    """exp of a pure quaternion (0; v), v:(B,3) -> (B,4)   (three/quaternion.py:287-311).

    theta is clamped at eps only in the division (quirk Q7)."""
    theta = v.norm(dim=-1, keepdim=True)
    return torch.cat((torch.cos(theta), 1.0 / theta.clamp(min=eps) * torch.sin(theta) * v), dim=-1)


def qlog(q, eps=1e-8):
    """log of a quaternion -> (B,4) = (log|q|, v/|v| * acos(s/|q|))   (three/quaternion.py:314-334)."""
    mag = q.norm(dim=-1, keepdim=True)
    s, v = q[..., :1], q[..., 1:]
    c = (s / mag.clamp(min=eps)).clamp(-1.0 + 1e-7, 1.0 - 1e-7)     # acos_safe, three/core.py:4-6
    return torch.cat((torch.log(mag), v / v.norm(dim=-1, keepdim=True).clamp(min=eps) * torch.acos(c)),
                     dim=-1)


def qmul(a, b):
    """Hamilton product a*b   (three/quaternion.py:198-218)."""
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw), dim=-1)


def to_matrix(q):
    """(B,4) -> (B,3,3); normalises its input first   (three/quaternion.py:39-93)."""
    w, x, y, z = unit(q).unbind(-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    rows = (1.0 - (ty * y + tz * z), ty * x - tz * w, tz * x + ty * w,
            ty * x + tz * w, 1.0 - (tx * x + tz * z), tz * y - tx * w,
            tz * x - ty * w, tz * y + tx * w, 1.0 - (tx * x + ty * y))
    return torch.stack(rows, dim=-1).reshape(-1, 3, 3)


def from_matrix(m, eps=1e-8):
    """(B,3,3) -> (B,4), branch on the trace / largest diagonal   (three/quaternion.py:96-176)."""
    tiny = torch.finfo(m.dtype).tiny

    def sdiv(n, d):
        return n / d.clamp(min=tiny)
    m00, m01, m02 = m[:, 0, 0], m[:, 0, 1], m[:, 0, 2]
    m10, m11, m12 = m[:, 1, 0], m[:, 1, 1], m[:, 1, 2]
    m20, m21, m22 = m[:, 2, 0], m[:, 2, 1], m[:, 2, 2]
    tr = m00 + m11 + m22
    s0 = torch.sqrt(tr + 1.0) * 2.0
    q0 = torch.stack((0.25 * s0, sdiv(m21 - m12, s0), sdiv(m02 - m20, s0), sdiv(m10 - m01, s0)), -1)
    s1 = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0
    q1 = torch.stack((sdiv(m21 - m12, s1), 0.25 * s1, sdiv(m01 + m10, s1), sdiv(m02 + m20, s1)), -1)
    s2 = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0
    q2 = torch.stack((sdiv(m02 - m20, s2), sdiv(m01 + m10, s2), 0.25 * s2, sdiv(m12 + m21, s2)), -1)
    s3 = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0
    q3 = torch.stack((sdiv(m10 - m01, s3), sdiv(m02 + m20, s3), sdiv(m12 + m21, s3), 0.25 * s3), -1)
    pick23 = torch.where((m11 > m22)[:, None], q2, q3)
    pick1 = torch.where(((m00 > m11) & (m00 > m22))[:, None], q1, pick23)
    return torch.where((tr > 0.0)[:, None], q0, pick1)


def axis_angle(axis, angle):
    """(B,3),(B,) or float -> (B,4)   (three/quaternion.py:254-284)."""
    if not torch.is_tensor(angle):
        angle = torch.full((axis.shape[0],), float(angle), dtype=axis.dtype)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    h = angle / 2.0
    return torch.cat((torch.cos(h)[:, None], torch.sin(h)[:, None] * axis), dim=-1)


def angular_distance(a, b, eps=1e-7):
    """pairwise 2*acos|<a,b>| -> (Na,Nb)   (three/quaternion.py:372-377)."""
    d = (unit(a) @ unit(b).t()).abs().clamp(-1.0 + eps, 1.0 - eps)
    return 2.0 * torch.acos(d)


def extrinsic(translation, q):
    """T(t) @ R(q) as (B,4,4)   (three/rigid.py:167-173)."""
    n = q.shape[0]
    m = torch.zeros(n, 4, 4, dtype=q.dtype)
    m[:, :3, :3] = to_matrix(q)
    m[:, :3, 3] = translation
    m[:, 3, 3] = 1.0
    return m


def sunflower_points(n, hemisphere=False):
    """Evenly spread sphere points (three/orientation.py:126-158, default pole only)."""
    k = torch.arange(0, n, dtype=torch.float32) + 0.5
    phi = torch.acos(1 - 2 * k / n / 2) if hemisphere else torch.acos(1 - 2 * k / n)
    th = math.pi * (1 + 5 ** 0.5) * k
    return torch.stack((torch.cos(th) * torch.sin(phi), torch.sin(th) * torch.sin(phi), torch.cos(phi)), 1)


def look_quats(forward, up=None):
    """Orientation whose +z is `forward`, random roll unless `up` (three/orientation.py:69-92).

    Consumes torch's global RNG exactly like the reference (one randn(n,3) + normalise)."""
    n = forward.shape[0]
    if up is None:
        d = torch.randn(n, 3)
        down = d / d.norm(p=2.0, dim=1, keepdim=True)
    else:
        down = -(torch.tensor(up).unsqueeze(0).expand(n, 3) + forward)
    right = torch.cross(down, forward, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    down = torch.cross(forward, right, dim=-1)
    down = down / down.norm(dim=-1, keepdim=True)
    return from_matrix(torch.stack((right, down, forward), dim=1))


def evenly_distributed_quats(n, hemisphere=False, upright=False):
    """three/orientation.py:161-164."""
    return look_quats(-sunflower_points(n, hemisphere), (0.0, 0.0, 1.0) if upright else None)
