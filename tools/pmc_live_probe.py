#!/usr/bin/env python
"""Workload of bench.py's LIVE counter passes (tools/pmc_live.py): a calibration copy of known size (1 GiB read + 1 GiB written),
then the headline's dominant kernel at the bench shape SYN(128,16), N = 8 -- the Winograd conv3d in its forward form and in its
data-gradient form with the fused previous-layer backward -- REP launches each.   python tools/pmc_live_probe.py [REP=2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, C, S = 8, 16, 128
g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(N, C, S, S, S, generator=g).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = torch.zeros(16).cuda()
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
he = ops.he_constant(w)
for _ in range(REP):
    y = x.clone()            # calibration: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)
for _ in range(REP):                   # forward form
    y, nrm = ops.conv3d_c16_wino(x, up, b, he, flags)
torch.cuda.synchronize()
for _ in range(REP):                   # data-gradient form with the producer's epilogue backward fused
    gx, _ = ops.conv3d_c16_wino(x, upt, None, he, 0, prev=(y, nrm, flags))
torch.cuda.synchronize()
print('ok')
