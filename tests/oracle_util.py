"""Shared by the CPU and GPU suites: one render + pose-loss + backward of the oracle in a chosen precision."""
import torch

import lf_oracle as O
from lf_oracle import pose

WEIGHTS = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.1, 'mask': 0.2}


def noise_case(S, C, N, device='cpu'):
    """Weights, latent volume, target data and initial cameras (all fp32, seeded)."""
    from latentfusion_amd import synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.pose import utils as pu
    model, cks = synth.build_model(S, C, 'pool:mean', seed=0, device=device, bias_std=0.05)
    z = torch.randn(1, 1, C, S, S, S, generator=torch.Generator().manual_seed(1))
    td = synth.make_observation_data(1, seed=2)
    torch.manual_seed(3)
    init = pu.sample_cameras_with_estimate(N, Camera(td['intrinsic'], td['extrinsic']))
    return {'S': S, 'model': model, 'cks': cks, 'z': z, 'td': td, 'init': init}


def _cast(o, dt):
    if torch.is_tensor(o):
        return o.to(dt) if o.is_floating_point() else o
    if isinstance(o, dict):
        return {k: _cast(v, dt) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_cast(v, dt) for v in o)
    return o


def oracle_loss_grad(dt, case):
    """(weighted total loss (N,), d mean(total) / d [log_q, t, viewport] (N,10)) of the oracle evaluated in `dt`."""
    S, init = case['S'], case['init']
    # the oracle keeps the reference's explicit .float() up-casts; in the fp64 pass they must not truncate
    keep_float, keep_default = torch.Tensor.float, torch.get_default_dtype()
    try:
        torch.set_default_dtype(dt)
        if dt == torch.float64:
            torch.Tensor.float = lambda self, *a, **k: self.double()
        ck, t = _cast(case['cks'], dt), _cast(case['td'], dt)
        model = pose.Model(*ck)
        cam = O.Cam(*(v.detach().clone().to(dt) for v in (init.intrinsic, init.log_quaternion, init.translation)))
        cam = cam.zoom(None, S, ck[3])
        for p in (cam.log_q, cam.t, cam.viewport):
            p.requires_grad_(True)
        target = pose.Obs(None, t['depth'], t['mask'], O.Cam.from_extrinsic(t['intrinsic'], t['extrinsic']))
        y, _ = model.render_latent_object(case['z'].to(dt), cam, apply_mask=True)
        ld = pose.pose_loss(target, cam.denormalize_depth(y['depth'].squeeze(0)), y['mask_logits'].squeeze(0), cam)
        total = pose.weigh(ld, WEIGHTS)
        total.mean().backward()
        return total.detach().double(), torch.cat((cam.log_q.grad, cam.t.grad, cam.viewport.grad), dim=1).double()
    finally:
        torch.Tensor.float = keep_float
        torch.set_default_dtype(keep_default)


def in_fp64(fn):
    """Runs fn() with the oracle in double precision: default dtype fp64 and the reference's explicit .float() up-casts
    (which the oracle keeps) turned into .double()."""
    keep_float, keep_default = torch.Tensor.float, torch.get_default_dtype()
    try:
        torch.set_default_dtype(torch.float64)
        torch.Tensor.float = lambda self, *a, **k: self.double()
        return fn()
    finally:
        torch.Tensor.float = keep_float
        torch.set_default_dtype(keep_default)


def cast(o, dt):
    return _cast(o, dt)
