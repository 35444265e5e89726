#!/usr/bin/env python
"""Workload for the PMC passes over the training-step kernels (tools/pmc_collect_train.sh): a calibration copy of known
size (1 GiB read + 1 GiB written), then at 8 x 128^3 x 16: the bf16 ring convolution (forward and addend form), the bf16
and the fp32 weight gradient, and the deterministic volume splat (object->camera map, per-sample volumes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops, synth  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402
from latentfusion_amd.modules.geometry import Camera, o2c_coefficients  # noqa: E402
from latentfusion_amd.pose import utils as pu  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, C, S = 8, 16, 128
g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(N, C, S, S, S, generator=g).cuda())
gp = ops.cl((torch.randn(N, C, S, S, S, generator=g) * 1e-3).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = torch.zeros(16).cuda()
he = ops.he_constant(w)
for _ in range(REP):
    y = x.clone()            # calibration: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
wp = ops.pack_conv3d_c16_ring_bf16(w)
for _ in range(REP):
    y, nrm = ops.conv3d_c16_ring_bf16(x, wp, b, he, LF_EPI_LRELU | LF_EPI_PIXELNORM, 1)
torch.cuda.synchronize()
for _ in range(REP):
    ya, _ = ops.conv3d_c16_ring_bf16(x, wp, None, he, 0, 0, addend=gp)
torch.cuda.synchronize()
for _ in range(REP):
    with ops.autocast():
        gw, _ = ops.conv_bwd_weight(x, gp, 3, 16, he, want_bias=False)
torch.cuda.synchronize()
for _ in range(REP):
    gw32, _ = ops.conv_bwd_weight(x, gp, 3, 16, he, want_bias=False)
torch.cuda.synchronize()
td = synth.make_observation_data(1, seed=2)
torch.manual_seed(3)
cams = pu.sample_cameras_with_estimate(N, Camera(td['intrinsic'], td['extrinsic'])).zoom(None, S, 2.85).to('cuda')
coef = o2c_coefficients(cams, 1.0).cuda()
for _ in range(REP):
    v = x.clone().requires_grad_(True)
    ops.resample_o2c(v, coef).backward(gp)
torch.cuda.synchronize()
print('ok')
