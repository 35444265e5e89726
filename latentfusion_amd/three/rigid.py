"""Rigid-transform matrices (API mirror of the used part of latentfusion/three/rigid.py)."""
import torch

from . import quaternion
from .core import ensure_batch_dim


def intrinsic_to_3x4(matrix):
    matrix, squeezed = ensure_batch_dim(matrix, 2)
    out = torch.cat((matrix, torch.zeros(matrix.shape[0], 3, 1, dtype=matrix.dtype, device=matrix.device)), dim=-1)
    return out.squeeze(0) if squeezed else out


def rotation_to_4x4(matrix):
    matrix, squeezed = ensure_batch_dim(matrix, 2)
    out = torch.zeros(matrix.shape[0], 4, 4, dtype=matrix.dtype, device=matrix.device)
    out[:, :3, :3] = matrix
    out[:, 3, 3] = 1.0
    return out.squeeze(0) if squeezed else out


def translation_to_4x4(translation):
    translation, squeezed = ensure_batch_dim(translation, 1)
    out = torch.eye(4, dtype=translation.dtype, device=translation.device).repeat(translation.shape[0], 1, 1)
    out[:, :3, 3] = translation
    return out.squeeze(0) if squeezed else out


def decompose(matrix):
    """Splits (B,4,4) rigid transforms into pure rotation and pure translation matrices."""
    matrix, squeezed = ensure_batch_dim(matrix, 2)
    R = rotation_to_4x4(matrix[:, :3, :3])
    T = translation_to_4x4(matrix[:, :3, 3])
    if squeezed:
        return R.squeeze(0), T.squeeze(0)
    return R, T


def inverse_transform(matrix):
    matrix, squeezed = ensure_batch_dim(matrix, 2)
    Rt = matrix[:, :3, :3].transpose(1, 2)
    out = torch.zeros_like(matrix)
    out[:, :3, :3] = Rt
    out[:, :3, 3] = -(Rt @ matrix[:, :3, 3:]).squeeze(2)
    out[:, 3, 3] = 1
    return out.squeeze(0) if squeezed else out


def to_extrinsic_matrix(translation, quat):
    """T(t) @ R(q)  (reference rigid.py:167-173)."""
    return translation_to_4x4(translation) @ rotation_to_4x4(quaternion.quat_to_mat(quat))


def extrinsic_to_position(extrinsic):
    extrinsic, squeezed = ensure_batch_dim(extrinsic, 2)
    pos = (extrinsic[:, :3, :3].transpose(1, 2) @ extrinsic[:, :3, 3:]).squeeze(-1)
    return pos.squeeze(0) if squeezed else pos


def translate_matrix(matrix, offset):
    """Shifts the frame the transform maps FROM by `offset` (reference rigid.py:53-63): inv(inv(M) + t)."""
    single = matrix.dim() == 2
    m = matrix.unsqueeze(0) if single else matrix
    out = inverse_transform(m)
    out[:, :3, 3] += offset
    out = inverse_transform(out)
    return out.squeeze(0) if single else out


def scale_matrix(matrix, scale):
    """Scales the frame the transform maps FROM (reference rigid.py:66-76): inv(scale_t(inv(M)))."""
    single = matrix.dim() == 2
    m = matrix.unsqueeze(0) if single else matrix
    out = inverse_transform(m)
    out[:, :3, 3] *= scale
    out = inverse_transform(out)
    return out.squeeze(0) if single else out


def extrinsic_to_quat(extrinsic):
    rot, _ = decompose(extrinsic)
    return quaternion.mat_to_quat(rot[..., :3, :3])
