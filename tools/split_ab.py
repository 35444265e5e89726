#!/usr/bin/env python
"""A/B timing of builds of the split-precision (f16x3) direct conv kernel in ONE process:

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -shared -Ilatentfusion_amd/csrc -Iinclude \
          latentfusion_amd/csrc/conv_split.hip -o scratch/split_a.so        (one per variant)
    python tools/split_ab.py scratch/split_a.so scratch/split_b.so ...

Every variant packs the weights with its OWN tap-pair table (lf_conv3d_c16_split_pairs), is checked against the
library's all-fp32 Winograd kernel (forward with bias + LeakyReLU + PixelNorm, and the data gradient fused with the
previous layer's backward, the latter on a tiny-valued gradient through the max-abs side channel) and timed
round-robin at the bench shape (N = 8, 128^3 x 16)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
S, N, ROUNDS = int(os.environ.get('AB_S', 128)), int(os.environ.get('AB_N', 8)), 7
g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = (torch.randn(16, generator=g) * 0.1).cuda()
gin = ops.cl((torch.randn(N, 16, S, S, S, generator=g) * 1e-7).cuda())
he = ops.he_constant(w)
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
st = torch.cuda.current_stream().cuda_stream

# reference: the all-fp32 Winograd kernel of the in-tree library
up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)
yref, nref = ops.conv3d_c16_wino(x, up, b, he, flags)
gref, _ = ops.conv3d_c16_wino(gin, upt, None, he, 0, prev=(yref, nref, flags))
amax = ops.amax_buffer(gin.abs().max(), 'cuda')


def bind(path):
    L = ctypes.CDLL(os.path.abspath(path))
    L.lf_conv3d_c16_split.restype = I
    L.lf_conv3d_c16_split.argtypes = [P, P, P, P, P, I, I, I, I, F, U, F, F, P, P, U, P, P, P]
    table = (I * 28)()
    L.lf_conv3d_c16_split_pairs(table)

    def pack(wt):
        taps = wt.reshape(16, 16, 27)
        k = torch.zeros(14, 16, 32, device='cuda')
        for p in range(14):
            for sel in range(2):
                if table[2 * p + sel] >= 0:
                    k[p, :, sel * 16:(sel + 1) * 16] = taps[:, :, table[2 * p + sel]]
        hi = k.half()
        return torch.stack((hi, (k - hi.float()).half()), dim=1).contiguous()
    return L, pack(w), pack(w.transpose(0, 1).flip(dims=(2, 3, 4)))


libs = [bind(p) for p in sys.argv[1:]]
state = []
for L, wp, wpt in libs:
    y, nrm, go = torch.empty_like(x), torch.empty(N * S ** 3, device='cuda'), torch.empty_like(x)
    fwd = lambda L=L, wp=wp, y=y, nrm=nrm: L.lf_conv3d_c16_split(  # noqa: E731
        x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), nrm.data_ptr(), N, S, S, S, he, flags, 0.2, 1e-8, None, None, 0, None, None, st)
    bwd = lambda L=L, wpt=wpt, go=go: L.lf_conv3d_c16_split(  # noqa: E731
        gin.data_ptr(), wpt.data_ptr(), None, go.data_ptr(), None, N, S, S, S, he, 0, 0.2, 1e-8, yref.data_ptr(), nref.data_ptr(), flags,
        amax.data_ptr(), None, st)
    assert fwd() == 0 and bwd() == 0
    torch.cuda.synchronize()
    state.append((y, nrm, go, fwd, bwd))
for i, p in enumerate(sys.argv[1:]):
    y, nrm, go = state[i][:3]
    print(f'{p}: fwd max|diff| vs fp32 Winograd {(y - yref).abs().max().item():.2e} (norm {(nrm - nref).abs().max().item():.2e}), '
          f'bwd max rel diff {((go - gref).abs().max() / gref.abs().max()).item():.2e}')
tf, tb = [[] for _ in libs], [[] for _ in libs]
for r in range(ROUNDS):
    for i in range(len(libs)):
        for fn, acc in ((state[i][3], tf), (state[i][4], tb)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            acc[i].append(e0.elapsed_time(e1) / 5)
for i, p in enumerate(sys.argv[1:]):
    a, c = sorted(tf[i]), sorted(tb[i])
    print(f'{p}: fwd median {a[len(a) // 2]:.4f} ms (min {a[0]:.4f}), bwd+prev median {c[len(c) // 2]:.4f} ms (min {c[0]:.4f})')
