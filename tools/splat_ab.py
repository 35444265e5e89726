#!/usr/bin/env python
"""A/B of the deterministic camera -> object splat with a volume per sample (lf_resample3d_bwd_vol_det_io, bf16 storage both ways):
tile form with box culling (lf_set_tuning(4, 3)) against the binned form (default, round 5), N x 128^3 x 16, HIP-event times and
bit-identity of the results.   python tools/splat_ab.py [N=8 [wide]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import _lib, ops, synth  # noqa: E402
from latentfusion_amd.modules.geometry import Camera, c2o_coefficients  # noqa: E402
from latentfusion_amd.pose import utils as pu  # noqa: E402

N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 128
if os.environ.get('LF_HIP_LIB'):                                    # A/B of two BUILDS of the library (tool only)
    _lib.LIB_PATH = os.environ['LF_HIP_LIB']
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
cl3 = torch.channels_last_3d
gp = ops.cl((torch.randn(N, 16, S, S, S, generator=g) * 1e-3).cuda()).to(torch.bfloat16).contiguous(memory_format=cl3)
if len(sys.argv) > 2 and sys.argv[2] == 'wide':
    # samples spread around a target pose and a wide zoom: two thirds of the object's voxels fall outside the camera volume (clamped)
    td = synth.make_observation_data(1, seed=2)
    torch.manual_seed(3)
    cams = pu.sample_cameras_with_estimate(N, Camera(td['intrinsic'], td['extrinsic'])).zoom(None, S, 2.85).to('cuda')
else:
    # the training step's geometry (bench.py cfg5: synthetic reference views zoomed at the model's camera distance; 18 % clamped)
    d = synth.make_observation_data(N, seed=100)
    cams = Camera(d['intrinsic'], d['extrinsic'], width=d['width'], height=d['height']).zoom(
        None, S, synth.make_syn_checkpoints(S, 16, 'gru', 0)[3]).to('cuda')
coef = c2o_coefficients(cams, 1.0).cuda()
cf = torch.zeros(N, _lib.LF_MAP_COEFS, device='cuda')
cf[:, :coef.shape[1]] = coef
nb = max(L.lf_resample3d_bwd_vol_det_io_scratch_bytes(N, N, S, S, S), L.lf_resample3d_bwd_vol_det_io_scratch_bytes(1, N, S, S, S))
scr = torch.empty(nb // 8 + 1, device='cuda', dtype=torch.int64)
outs = {}
for name, variant in (('tile form (box culling)', 3), ('binned, 16 lanes per entry', 4), ('binned, quad per entry', 2)):
    prev = L.lf_set_tuning(4, variant)
    gv = ops.empty_cl16((N, 16, S, S, S), 'cuda', True)
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.lf_resample3d_bwd_vol_det_io(gp.data_ptr(), cf.data_ptr(), _lib.LF_MAP_C2O, gv.data_ptr(), N, scr.data_ptr(),
                                                  scr.numel() * 8, N, S, S, S, 3, s), 'splat')
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    L.lf_set_tuning(4, prev)
    outs[name] = gv
    print(f'{name:28s} N = {N}: {min(ts[1:]):8.3f} ms  (runs {", ".join("%.3f" % t for t in ts)}), scratch {nb / 2**20:.0f} MiB')
a, q, b = outs.values()
print('bit-identical:', bool(torch.equal(a, b)), bool(torch.equal(q, b)))
print('checksum C2O:', int(b.view(torch.int16).to(torch.int64).sum().item()), float(b.float().abs().sum().item()))
# the renderer's direction: object -> camera, ONE volume shared by the N samples
from latentfusion_amd.modules.geometry import o2c_coefficients  # noqa: E402
coef = o2c_coefficients(cams, 1.0).cuda()
cf = torch.zeros(N, _lib.LF_MAP_COEFS, device='cuda')
cf[:, :coef.shape[1]] = coef
outs = {}
for name, variant in (('O2C shared: tile form', 3), ('O2C shared: binned 16 lanes', 4), ('O2C shared: binned quad', 2)):
    prev = L.lf_set_tuning(4, variant)
    gv = ops.empty_cl16((1, 16, S, S, S), 'cuda', True)
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.lf_resample3d_bwd_vol_det_io(gp.data_ptr(), cf.data_ptr(), _lib.LF_MAP_O2C, gv.data_ptr(), 1, scr.data_ptr(),
                                                  scr.numel() * 8, N, S, S, S, 3, s), 'splat')
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    L.lf_set_tuning(4, prev)
    outs[name] = gv
    print(f'{name:28s} N = {N}: {min(ts[1:]):8.3f} ms  (runs {", ".join("%.3f" % t for t in ts)})')
a, q, b = outs.values()
print('bit-identical:', bool(torch.equal(a, b)), bool(torch.equal(q, b)))
print('checksum O2C:', int(b.view(torch.int16).to(torch.int64).sum().item()), float(b.float().abs().sum().item()))
