#!/usr/bin/env python
"""Accuracy and speed of the Winograd F(2^3,3^3) fp32 conv3d against the direct fp32 MFMA kernel.
   python tools/conv_wino_probe.py [S] [N] [reps]"""
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
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = (torch.randn(16, generator=g) * 0.1).cuda()
he = ops.he_constant(w)
up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)

# ---- odd-shaped small volume (partial tiles on every axis) against fp64 ----
for shape in ((2, 16, 10, 20, 40), (1, 16, 5, 7, 9), (3, 16, 4, 8, 16)):
    xs = torch.randn(*shape, generator=g)
    xs = xs / torch.sqrt((xs ** 2).mean(dim=1, keepdim=True))
    xd = ops.cl(xs.cuda())
    y64 = torch.nn.functional.conv3d(xs.double(), w.double().cpu(), None, 1, 1) * he + b.double().cpu().view(1, -1, 1, 1, 1)
    raw64 = y64.clone()
    y64 = torch.nn.functional.leaky_relu(y64, 0.2)
    n64 = torch.sqrt((y64 ** 2).mean(dim=1, keepdim=True) + 1e-8)
    y64 = y64 / n64
    ref, nref = ops._conv3x3_raw(xd, ops.pack_conv3x3(w), b, 16, he, flags, True)
    got, ngot = ops.conv3d_c16_wino(xd, up, b, he, flags)
    torch.cuda.synchronize()
    print(f'{shape}: vs fp64  direct {(ref.cpu().double() - y64).abs().max().item():.3e}  wino {(got.cpu().double() - y64).abs().max().item():.3e}'
          f'   norm err {(ngot.cpu().double().view(n64.shape) - n64).abs().max().item():.3e}')
    # raw (no epilogue, no bias) data-gradient form
    gref = ops.conv3x3_bwd_data(xd, ops.pack_conv3x3(w, transpose=True), 16, he, None)
    gw, _ = ops.conv3d_c16_wino(xd, upt, None, he, 0)
    g64 = torch.nn.functional.conv_transpose3d(xs.double(), w.double().cpu(), None, 1, 1) * he
    print(f'     data-grad vs fp64: direct {(gref.cpu().double() - g64).abs().max().item():.3e}  wino {(gw.cpu().double() - g64).abs().max().item():.3e}')
    # fused previous-layer backward
    prev = (ref, nref, flags)
    gref2 = ops.conv3x3_bwd_data(xd, ops.pack_conv3x3(w, transpose=True), 16, he, prev)
    amax = ops.amax_buffer(None, 'cuda')
    gw2, _ = ops.conv3d_c16_wino(xd, upt, None, he, 0, prev=prev, amax_out=amax)
    torch.cuda.synchronize()
    print(f'     fused prev-bwd: wino vs direct {(gw2 - gref2).abs().max().item():.3e} (max |g| {gref2.abs().max().item():.3e}); amax {amax.max().item():.6e} vs {gw2.abs().max().item():.6e}')

x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda())
x = x / torch.sqrt((x ** 2).mean(dim=1, keepdim=True))
ref, nref = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, flags, True)
got, ngot = ops.conv3d_c16_wino(x, up, b, he, flags)
torch.cuda.synchronize()
err = (got - ref).abs()
print(f'S={S} N={N} fwd: wino vs direct max {err.max().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e}; norm max rel {((ngot - nref).abs() / nref).max().item():.3e}')
wp = ops.pack_conv3x3(w)
us, ust = ops.pack_conv3d_c16_wino_split(w), ops.pack_conv3d_c16_wino_split(w, transpose=True)
gs, _ = ops.conv3d_c16_wino_split(x, us, b, he, flags)
print(f'split-winograd vs direct: max {(gs - ref).abs().max().item():.3e}')
for name, fn in (('direct fp32 MFMA', lambda: ops._conv3x3_raw(x, wp, b, 16, he, flags, True)),
                 ('winograd f16x3  ', lambda: ops.conv3d_c16_wino_split(x, us, b, he, flags)),
                 ('winograd f16x3 bwd', lambda: ops.conv3d_c16_wino_split(x, ust, None, he, 0, prev=(ref, nref, flags))),
                 ('winograd fp32   ', lambda: ops.conv3d_c16_wino(x, up, b, he, flags)),
                 ('winograd bwd+prev', lambda: ops.conv3d_c16_wino(x, upt, None, he, 0, prev=(ref, nref, flags)))):
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
    print(f'{name}: median {ms[len(ms) // 2]:.3f} ms  ({fl / ms[len(ms) // 2] / 1e9:.1f} algorithmic TFLOP/s)')
