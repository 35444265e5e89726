#!/usr/bin/env python
"""Accuracy and speed of the split-precision (f16x3) conv3d against the exact-fp32 MFMA kernel.
   python tools/conv_split_probe.py [S] [N] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda())
x = x / torch.sqrt((x ** 2).mean(dim=1, keepdim=True))                   # PixelNorm-like range
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = (torch.randn(16, generator=g) * 0.1).cuda()
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
he = ops.he_constant(w)
ref, nref = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, flags, True)
ws = ops.pack_conv3d_c16_split(w)
got, ngot = ops.conv3d_c16_split(x, ws, b, he, flags)
torch.cuda.synchronize()
err = (got - ref).abs()
print(f'fwd: max abs err {err.max().item():.3e}  rms err {err.pow(2).mean().sqrt().item():.3e}  (|y| rms {ref.pow(2).mean().sqrt().item():.3f}); '
      f'norm max rel err {((ngot - nref).abs() / nref).max().item():.3e}')
# fp64 reference on a sub-block to see which of the two is closer to the truth
xs = x[:1, :, :12, :12, :20].double().cpu()
w64 = w.double().cpu()
y64 = torch.nn.functional.conv3d(xs.contiguous(), w64, None, 1, 1) * he + b.double().cpu().view(1, -1, 1, 1, 1)
y64 = torch.nn.functional.leaky_relu(y64, 0.2)
y64 = y64 / torch.sqrt((y64 ** 2).mean(dim=1, keepdim=True) + 1e-8)
sl = (slice(0, 1), slice(None), slice(1, 11), slice(1, 11), slice(1, 19))
print(f'vs fp64 (interior block): exact-fp32 kernel {(ref[sl].cpu().double() - y64[sl]).abs().max().item():.3e}   '
      f'f16x3 kernel {(got[sl].cpu().double() - y64[sl]).abs().max().item():.3e}')
# gradient-scale input (tiny values) through the data-gradient form with amax scaling
gsmall = x * 3e-7
amax = ops.amax_buffer(gsmall.abs().max(), 'cuda')
wt = ops.pack_conv3x3(w, transpose=True)
gref = ops.conv3x3_bwd_data(gsmall, wt, 16, he, None)
gsp, _ = ops.conv3d_c16_split(gsmall, ops.pack_conv3d_c16_split(w, transpose=True), None, he, 0, amax_in=amax)
e2 = (gsp - gref).abs().max().item() / gref.abs().max().item()
print(f'tiny-gradient data-grad (amax-scaled): max err / max|g| = {e2:.3e}')
for name, fn in (('exact fp32 MFMA', lambda: ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, flags, True)),
                 ('f16x3 split    ', lambda: ops.conv3d_c16_split(x, ws, b, he, flags))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    fl = 2.0 * 27 * 256 * S ** 3 * N
    print(f'{name}: median {ms[len(ms) // 2]:.3f} ms  ({fl / ms[len(ms) // 2] / 1e9:.1f} algorithmic TFLOP/s, '
          f'{(2 * 16 * S ** 3 * N * 4) / ms[len(ms) // 2] / 1e6:.0f} GB/s algorithmic)')
