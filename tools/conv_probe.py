#!/usr/bin/env python
"""Times the dominant kernel (fused conv3d C->C block on an (N,C,S,S,S) activation) in isolation
with HIP events on the launch stream.   python tools/conv_probe.py [S] [C] [N] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(N, C, S, S, S, generator=g).cuda())
w = torch.randn(C, C, 3, 3, 3, generator=g).cuda()
b = torch.zeros(C).cuda()
for _ in range(2):
    y = ops.conv3x3(x, w, b)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for e0, e1 in ev:
    e0.record()
    y = ops.conv3x3(x, w, b)
    e1.record()
torch.cuda.synchronize()
ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
flops = 2.0 * 27 * C * C * S ** 3 * N
print(f'conv3d {C}->{C} on {N}x{S}^3: median {ms[len(ms) // 2]:.3f} ms  min {ms[0]:.3f} ms  '
      f'{flops / ms[len(ms) // 2] / 1e9:.1f} TFLOP/s (median)  {flops / ms[0] / 1e9:.1f} TFLOP/s (best)')
