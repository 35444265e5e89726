#!/usr/bin/env python
"""Ablation timing of the bf16 ring convolution in its training form (bf16 storage in and out, forward with bias + LeakyReLU +
PixelNorm): builds of conv_split.hip with -DSPLIT_ABL=<mask> (1 no MFMAs, 2 no operand reads, 4 no halo fetch, 8 no stores,
16 no commit of the staged planes to LDS), one process, round-robin, N x 128^3 x 16.

    for v in 0 1 2 4 8 16; do hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -shared \
        -Ilatentfusion_amd/csrc -Iinclude -DSPLIT_ABL=$v latentfusion_amd/csrc/conv_split.hip -o scratch/ring_abl$v.so; done
    python tools/ring_bf16_abl.py [N=8] scratch/ring_abl0.so scratch/ring_abl1.so ..."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
N = int(sys.argv[1])
S = 128
g = torch.Generator().manual_seed(0)
cl3 = torch.channels_last_3d
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda()).to(torch.bfloat16).contiguous(memory_format=cl3)
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = (torch.randn(16, generator=g) * 0.1).cuda()
he = ops.he_constant(w)
wp = ops.pack_conv3d_c16_ring_bf16(w)
y = torch.empty_like(x)
nrm = torch.empty(N * S ** 3, device='cuda')
st = torch.cuda.current_stream().cuda_stream
libs = []
for path in sys.argv[2:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.lf_conv3d_c16_ring_bf16_io.restype = I
    L.lf_conv3d_c16_ring_bf16_io.argtypes = [P, P, P, P, P, I, I, I, I, F, U, F, F, P, I, I, P]
    libs.append((os.path.basename(path), L))


def run(L):
    rc = L.lf_conv3d_c16_ring_bf16_io(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), nrm.data_ptr(), N, S, S, S, he,
                                      LF_EPI_LRELU | LF_EPI_PIXELNORM, 0.2, 1e-8, None, 1, 3, st)
    assert rc == 0, rc


times = {n: [] for n, _ in libs}
for rnd in range(6):
    for name, L in libs:
        run(L)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run(L)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
for name, _ in libs:
    t = sorted(times[name])
    print(f'{name:20s} N = {N}: median {t[len(t) // 2]:7.3f} ms   min {t[0]:7.3f}')
