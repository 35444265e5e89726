#!/usr/bin/env python
"""Workload for the HBM-traffic PMC passes: a calibration copy of known size (1 GiB read + 1 GiB
written by ATen's copy kernel) followed by the conv3d kernels (direct and Winograd) on the bench shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(8, 16, 128, 128, 128, generator=g).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = torch.zeros(16).cuda()
for _ in range(3):
    y = x.clone()            # calibration: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402
for _ in range(3):                   # direct implicit-GEMM kernel (conv3d_c16_persistent_kernel)
    y, _n = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, ops.he_constant(w), LF_EPI_LRELU | LF_EPI_PIXELNORM, True)
torch.cuda.synchronize()
up = ops.pack_conv3d_c16_wino(w)
for _ in range(3):
    y, _n = ops.conv3d_c16_wino(x, up, b, ops.he_constant(w), LF_EPI_LRELU | LF_EPI_PIXELNORM)
torch.cuda.synchronize()
