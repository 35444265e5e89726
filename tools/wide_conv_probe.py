#!/usr/bin/env python
"""A/B timing of the wide (>= 64-channel) 3x3(x3) convolution paths at the released model's layer shapes
(tools/train/train.sh:28-66): 'fused' = Winograd input transform + lf_wino_fused_gemm (own fp32-MFMA GEMM with the
output transform / epilogue folded in), 'bmm' = three-stage form with the per-frequency products on the library GEMM.
Prints per shape: total ms of the conv (+ PixelNorm), ms of the GEMM stage alone (HIP events on the launch stream), the
executed fp32-MFMA rate of that stage and the max abs difference between the two paths.

    python tools/wide_conv_probe.py [N] [--cfgs]     (--cfgs: also time every workgroup shape of the fused GEMM, lf_set_tuning key 3)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

CFGS = '--cfgs' in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(args[0]) if args else 8
from latentfusion_amd import _lib  # noqa: E402
L = _lib.lib()
L.lf_set_tuning.restype = __import__('ctypes').c_int
L.lf_set_tuning.argtypes = [__import__('ctypes').c_int] * 2
SHAPES = [(3, 256, 256, 16), (2, 64, 64, 256), (2, 128, 64, 256), (2, 196, 128, 128), (2, 256, 196, 64), (2, 512, 256, 32),
          (2, 1024, 512, 16), (2, 1024, 512, 8), (2, 512, 512, 4)]
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
if '--more' in sys.argv:                                         # low-channel / high-resolution layers (and transposed forms)
    SHAPES = [(2, 64, 64, 256), (2, 64, 128, 256), (2, 128, 64, 256), (2, 64, 128, 128), (2, 128, 128, 128), (2, 128, 196, 64), (2, 196, 128, 64),
              (2, 64, 64, 128), (3, 64, 64, 16), (3, 64, 64, 32)]
out = []
for dims, cin, cout, S in SHAPES:
    g = torch.Generator().manual_seed(cin + cout + S)
    x = ops.cl(torch.randn((N, cin) + (S,) * dims, generator=g).cuda())
    w = torch.randn((cout, cin) + (3,) * dims, generator=g).cuda()
    b = torch.zeros(cout).cuda()
    he = ops.he_constant(w)
    F = 64 if dims == 3 else 16
    tiles = N * (S // 2) ** dims
    flops = 2.0 * F * tiles * cin * cout
    rec = {'dims': dims, 'cin': cin, 'cout': cout, 'S': S, 'N': N, 'executed_GFLOP': flops / 1e9}
    ys = {}
    for mode in ('fused', 'bmm'):
        ops.WIDE_CONV_MODE = mode
        for _ in range(3):
            y, _n = ops.wide_conv(x, w, b, he, flags)
        torch.cuda.synchronize()
        ops.KERNEL_TIMER = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y, _n = ops.wide_conv(x, w, b, he, flags)
        e1.record()
        torch.cuda.synchronize()
        tm, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
        gemm = [a.elapsed_time(c) for n_, a, c in tm if n_.endswith('_fused') or n_.endswith('_gemm')]
        rec[mode] = {'conv_ms': e0.elapsed_time(e1) / 10, 'gemm_stage_ms': sum(gemm) / len(gemm),
                     'gemm_stage_TFLOPs': flops / (sum(gemm) / len(gemm) * 1e-3) / 1e12}
        ys[mode] = y
    ops.WIDE_CONV_MODE = 'fused'
    rec['max_abs_diff'] = (ys['fused'] - ys['bmm']).abs().max().item()
    if '--direct' in sys.argv:                                       # the direct implicit-GEMM kernel (no Winograd) on the same layer
        wp = ops.pack_conv3x3(w)
        for _ in range(3):
            yd, _n = ops._conv3x3_raw(x, wp, b, cout, he, flags, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            yd, _n = ops._conv3x3_raw(x, wp, b, cout, he, flags, True)
        e1.record()
        torch.cuda.synchronize()
        rec['direct'] = {'conv_ms': e0.elapsed_time(e1) / 10, 'max_abs_diff_vs_bmm': (yd - ys['bmm']).abs().max().item()}
    if CFGS:
        for cfg in range(5):
            if dims == 3 and cfg >= 3:
                continue
            L.lf_set_tuning(3, cfg)
            for _ in range(3):
                y, _n = ops.wide_conv(x, w, b, he, flags)
            torch.cuda.synchronize()
            ops.KERNEL_TIMER = []
            for _ in range(10):
                y, _n = ops.wide_conv(x, w, b, he, flags)
            torch.cuda.synchronize()
            tm, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
            gemm = [a.elapsed_time(c) for n_, a, c in tm if n_.endswith('_fused')]
            rec[f'cfg{cfg}'] = {'gemm_stage_ms': sum(gemm) / len(gemm), 'gemm_stage_TFLOPs': flops / (sum(gemm) / len(gemm) * 1e-3) / 1e12,
                                'max_abs_diff_vs_bmm': (y - ys['bmm']).abs().max().item()}
        L.lf_set_tuning(3, -1)
        for _ in range(3):
            y, _n = ops.wide_conv(x, w, b, he, flags)
        torch.cuda.synchronize()
        ops.KERNEL_TIMER = []
        for _ in range(10):
            y, _n = ops.wide_conv(x, w, b, he, flags)
        torch.cuda.synchronize()
        tm, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
        gemm = [a.elapsed_time(c) for n_, a, c in tm if n_.endswith('_fused')]
        rec['auto_again'] = {'gemm_stage_TFLOPs': flops / (sum(gemm) / len(gemm) * 1e-3) / 1e12}
    out.append(rec)
    print(json.dumps(rec))
