#!/usr/bin/env python
"""A/B of builds of wgrad.hip in one process: lf_conv_bwd_weight_bf16_io (x, gpre in bf16 storage) on 1 and 32 volumes of
128^3 x 16.
    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared -Ilatentfusion_amd/csrc <wgrad source> -o scratch/wg_X.so
    python tools/wgrad_ab.py scratch/wg_a.so scratch/wg_b.so"""
import ctypes
import os
import sys

import torch

P = ctypes.c_void_p
S = 128
res = {}
for N in (1, 32):
    g = torch.Generator().manual_seed(N)
    x = (torch.randn(N, S, S, S, 16, generator=g)).to(torch.bfloat16).cuda()
    gp = (torch.randn(N, S, S, S, 16, generator=g) * 1e-2).to(torch.bfloat16).cuda()
    outs = []
    for path in sys.argv[1:]:
        lib = ctypes.CDLL(os.path.abspath(path))
        f = lib.lf_conv_bwd_weight_bf16_io
        f.restype = ctypes.c_int
        f.argtypes = [P, P, P, P, ctypes.c_size_t] + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_int, P]
        sb = lib.lf_conv_bwd_weight_scratch_bytes
        sb.restype = ctypes.c_size_t
        sb.argtypes = [ctypes.c_int] * 7
        nb = sb(3, N, S, S, S, 16, 16)
        scr = torch.empty(nb // 4 + 1, device='cuda')
        gw = torch.empty(27, 16, 16, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(x.data_ptr(), gp.data_ptr(), gw.data_ptr(), scr.data_ptr(), scr.numel() * 4, 3, N, S, S, S, 16, 16, 0.1, 3, st)  # noqa: E731
        assert call() == 0
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        outs.append(gw.clone())
        print(f'N={N} {path}: median {sorted(ts)[3] * 1e3:.1f} us per launch (incl. the reduce kernel)  '
              f'= {2 * N * S ** 3 * 32 / sorted(ts)[3] / 1e6:.0f} GB/s of bf16 operands')
    print('  identical results:', all(torch.equal(o, outs[0]) for o in outs), ' max |diff| / max |ref|:',
          [float((o - outs[0]).abs().max() / outs[0].abs().max()) for o in outs[1:]])
