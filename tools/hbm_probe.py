#!/usr/bin/env python
"""Workload for the PMC passes (tools/pmc_collect.sh): a calibration copy of known size (1 GiB read + 1 GiB written)
followed by the heavy kernels of one pose iteration at the bench shape SYN(128,16), N=8: the Winograd conv3d
(forward form and data-gradient form with the fused previous-layer backward), the direct fp32 and f16x3 conv3d, the O2C resampler
(forward and coefficient gradient), the factor projection (forward / backward) and the column sum."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import _lib, ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, LF_MAP_O2C  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, C, S = 8, 16, 128
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(N, C, S, S, S, generator=g).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = torch.zeros(16).cuda()
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
he = ops.he_constant(w)
for _ in range(REP):
    y = x.clone()            # calibration: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
for _ in range(REP):                   # direct implicit-GEMM kernel (conv3d_c16_persistent_kernel)
    y, nrm = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, flags, True)
torch.cuda.synchronize()
up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)
for _ in range(REP):                   # Winograd, forward form
    y, nrm = ops.conv3d_c16_wino(x, up, b, he, flags)
torch.cuda.synchronize()
for _ in range(REP):                   # Winograd, data-gradient form with the producer's epilogue backward fused
    gx, _ = ops.conv3d_c16_wino(x, upt, None, he, 0, prev=(y, nrm, flags))
torch.cuda.synchronize()

# round 4: the factor projection fused into the last camera block's kernels (forward: default; backward: opt-in)
pwf = torch.randn(16, C * S, 1, 1, generator=g).cuda()
wdm = pwf.reshape(16, C, S).permute(0, 2, 1).reshape(16, S * C).contiguous()
wA, wtA = ops.pack_wino_proj(wdm), ops.pack_wino_proj(wdm, transpose=True)
phe = ops.he_constant(pwf)
for _ in range(REP):
    yf, nrmf, zpf, pnf = ops.conv3d_c16_wino_projfwd(x, up, b, he, flags, wA, None, phe, flags)
torch.cuda.synchronize()
gpf = ops.cl(torch.randn(N, 16, S, S, generator=g).cuda())
for _ in range(REP):
    gxf = ops.conv3d_c16_wino_projbwd(gpf, wtA, phe, yf, nrmf, flags, upt, he, prev=(y, nrm, flags))
torch.cuda.synchronize()
del yf, nrmf, gxf

sp, spt = ops.pack_conv3d_c16_split(w), ops.pack_conv3d_c16_split(w, transpose=True)
am = ops.amax_buffer(x.abs().max(), 'cuda')
for _ in range(REP):                   # direct f16x3 kernel, forward form
    ys, nrms = ops.conv3d_c16_split(x, sp, b, he, flags)
torch.cuda.synchronize()
for _ in range(REP):                   # ... and as a data gradient with the producer's epilogue backward fused
    gs, _ = ops.conv3d_c16_split(x, spt, None, he, 0, prev=(y, nrm, flags), amax_in=am)
torch.cuda.synchronize()

# O2C resampler: one object volume broadcast to N pose hypotheses (the engine's call)
from latentfusion_amd import synth  # noqa: E402
from latentfusion_amd.modules.geometry import Camera, o2c_coefficients  # noqa: E402
from latentfusion_amd.pose import utils as pu  # noqa: E402
td = synth.make_observation_data(1, seed=2)
torch.manual_seed(3)
cams = pu.sample_cameras_with_estimate(N, Camera(td['intrinsic'], td['extrinsic'])).zoom(None, S, 2.85).to('cuda')
cf = torch.zeros(N, 20, device='cuda')
cf[:, :18] = o2c_coefficients(cams, 1.0)
vol = ops.cl(torch.randn(1, C, S, S, S, generator=g).cuda())
out = ops.empty_cl((N, C, S, S, S), 'cuda')
for _ in range(REP):
    _lib.check(L.lf_resample3d_fwd(vol.data_ptr(), 1, cf.data_ptr(), LF_MAP_O2C, out.data_ptr(), N, S, S, S, C, s), 'fwd')
torch.cuda.synchronize()
gco = torch.empty(N, 18, device='cuda')
scr = torch.empty(L.lf_resample3d_bwd_coef_scratch_bytes(N, S, S, S) // 4 + 1, device='cuda')
for _ in range(REP):
    _lib.check(L.lf_resample3d_bwd_coef(x.data_ptr(), vol.data_ptr(), 1, cf.data_ptr(), gco.data_ptr(), scr.data_ptr(),
                                        scr.numel() * 4, N, S, S, S, C, s), 'bwd_coef')
torch.cuda.synchronize()

# factor projection (K = C*D folded by addressing) forward and its unfolding backward
pw = torch.randn(16, C * S, 1, 1, generator=g).cuda()
xr = x.detach().requires_grad_(True)
for _ in range(REP):
    zp = ops.factor_project(xr, pw, None)
    zp.sum().backward()
torch.cuda.synchronize()
for _ in range(REP):
    cs = ops.column_sum(x)
torch.cuda.synchronize()
print('hbm_probe done')
