#!/usr/bin/env python
"""Workload for PMC passes over lf_wino_fused_gemm: the released 3-D layer (256 -> 256 on 16^3, N = 8) and one 2-D
decoder layer (196 -> 128 on 128^2, N = 8), five launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU  # noqa: E402

for dims, cin, cout, S in ((3, 256, 256, 16), (2, 196, 128, 128)):
    g = torch.Generator().manual_seed(1)
    x = ops.cl(torch.randn((8, cin) + (S,) * dims, generator=g).cuda())
    w = torch.randn((cout, cin) + (3,) * dims, generator=g).cuda()
    for _ in range(5):
        ops.wide_conv(x, w, None, ops.he_constant(w), LF_EPI_LRELU)
    torch.cuda.synchronize()
