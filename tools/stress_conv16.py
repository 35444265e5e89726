"""Random-shape stress of the 16->16 conv kernels (Winograd, split-Winograd, data gradient, weight gradient)
against the direct fp32 kernels / ATen:  python tools/stress_conv16.py"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
random.seed(0); torch.manual_seed(0)
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
worst = 0
for it in range(60):
    N = random.choice([1, 1, 2, 3, 5]); D = random.randint(1, 21); H = random.randint(1, 37); W = random.randint(1, 45)
    x = ops.cl(torch.randn(N, 16, D, H, W).cuda())
    w = torch.randn(16, 16, 3, 3, 3).cuda(); b = torch.randn(16).cuda() * 0.1
    he = ops.he_constant(w)
    ref, nref = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, flags, True)
    got, ngot = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w), b, he, flags)
    gs, ngs = ops.conv3d_c16_wino_split(x, ops.pack_conv3d_c16_wino_split(w), b, he, flags)
    gref = ops.conv3x3_bwd_data(x, ops.pack_conv3x3(w, transpose=True), 16, he, (ref, nref, flags))
    gw, _ = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w, transpose=True), None, he, 0, prev=(ref, nref, flags))
    torch.cuda.synchronize()
    e = max((got - ref).abs().max().item(), (gs - ref).abs().max().item(), (ngot - nref).abs().max().item(),
            ((gw - gref).abs().max() / gref.abs().max().clamp(min=1e-6)).item())
    worst = max(worst, e)
    if e > 5e-5 or not torch.isfinite(got).all() or not torch.isfinite(gs).all():
        print('BAD', (N, D, H, W), e)
# weight-gradient kernel (fast path threshold 8192 voxels) on odd shapes
for it in range(20):
    N = random.choice([1, 2, 3]); D = random.randint(3, 25); H = random.randint(8, 40); W = random.randint(8, 50)
    x = ops.cl(torch.randn(N, 16, D, H, W).cuda()); gp = ops.cl(torch.randn(N, 16, D, H, W).cuda())
    gwt, gb = ops.conv_bwd_weight(x, gp, 3, 16, 0.5)
    xr = x.double().cpu(); gr = gp.double().cpu()
    want = torch.nn.grad.conv3d_weight(xr, (16, 16, 3, 3, 3), gr, padding=1) * 0.5
    got = gwt.reshape(3, 3, 3, 16, 16).permute(3, 4, 0, 1, 2).double().cpu()
    e = ((got - want).abs().max() / want.abs().max()).item()
    worst = max(worst, e)
    if e > 1e-4:
        print('BAD wgrad', (N, D, H, W), e, 'fast' if N * D * H * W >= 8192 else 'generic')
print('worst', worst)
