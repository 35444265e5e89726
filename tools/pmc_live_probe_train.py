#!/usr/bin/env python
"""Workload of bench.py's LIVE counter passes for the cfg 5 block (tools/pmc_live.py): a 1 GiB calibration copy, then the dominant
kernel of the training step -- the one-group bf16 ring convolution with the Block epilogue (bias, LeakyReLU, PixelNorm; bf16 records
in and out, norms stored) -- on 8 volumes of 128^3 x 16, REP launches.   python tools/pmc_live_probe_train.py [REP=2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import _lib, ops, ops_train  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, C, S = 8, 16, 128
g = torch.Generator().manual_seed(0)
x32 = ops.cl(torch.randn(N, C, S, S, S, generator=g).cuda())
x = x32.to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = torch.zeros(16).cuda()
he = ops.he_constant(w)
for _ in range(REP):
    y = x32.clone()            # calibration: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
p1 = ops.pack_conv3d_c16_ring_bf16(w).reshape(1, 14, 16, 32)
o16 = torch.empty_like(x)
nrm = torch.empty(N * S ** 3, device='cuda')
for _ in range(REP):
    ops_train.ring_multi(x, p1, he, [(o16, None, True)], extra=_lib.LF_RING_EX_BLOCK, e0=b, o2=nrm)            # ring_multi_kernel<1,true,4,6>
torch.cuda.synchronize()
print('ok')
