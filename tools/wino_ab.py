#!/usr/bin/env python
"""A/B timing of several builds of the Winograd conv kernel in ONE process (box-to-box variance is ~3 %, more than
most of the deltas being chased):

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared -Ilatentfusion_amd/csrc \
          latentfusion_amd/csrc/conv_wino.hip -o scratch/a.so        (one per variant)
    python tools/wino_ab.py scratch/a.so scratch/b.so ...

Each variant is checked against the first one (max |diff| of forward and fused-backward outputs) and timed
round-robin: forward (bias + LeakyReLU + PixelNorm) and data-gradient fused with the previous layer's backward."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

P = ctypes.c_void_p
S, N, ROUNDS = 128, 8, 7
g = torch.Generator().manual_seed(0)
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = (torch.randn(16, generator=g) * 0.1).cuda()
he = ops.he_constant(w)
up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
st = torch.cuda.current_stream().cuda_stream


def bind(path):
    L = ctypes.CDLL(os.path.abspath(path))
    f = L.lf_conv3d_c16_wino
    f.restype = ctypes.c_int
    f.argtypes = [P, P, P, P, P] + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_uint, ctypes.c_float, ctypes.c_float, P, P, ctypes.c_uint, P, P]
    return f


def run(f, y, nrm, gout):
    assert f(x.data_ptr(), up.data_ptr(), b.data_ptr(), y.data_ptr(), nrm.data_ptr(), N, S, S, S, he, flags, 0.2, 1e-8,
             None, None, 0, None, st) == 0
    return lambda: f(x.data_ptr(), upt.data_ptr(), None, gout.data_ptr(), None, N, S, S, S, he, 0, 0.2, 1e-8,
                     y.data_ptr(), nrm.data_ptr(), flags, None, st)


# `path:fwd` / `path:bwd` restricts a variant to one of the two calls (builds with a hard-wired epilogue)
specs = [(a.split(':') + ['both'])[:2] for a in sys.argv[1:]]
sys.argv[1:] = [a for a, _ in specs]
only = [o for _, o in specs]
fs = [bind(p) for p in sys.argv[1:]]
outs = []
for i, f in enumerate(fs):
    y, nrm, go = torch.empty_like(x), torch.empty(N * S ** 3, device='cuda'), torch.empty_like(x)
    bw = run(fs[0] if only[i] == 'bwd' else f, y, nrm, go)
    if only[i] != 'fwd':
        assert bw() == 0
    else:
        go.copy_(outs[0][2])
    torch.cuda.synchronize()
    outs.append((y, nrm, go, bw))
for i, p in enumerate(sys.argv[1:]):
    print(f'{p}: fwd diff vs first {(outs[i][0] - outs[0][0]).abs().max().item():.2e}, bwd diff {(outs[i][2] - outs[0][2]).abs().max().item():.2e}')
tf = [[] for _ in fs]
tb = [[] for _ in fs]
for r in range(ROUNDS):
    for i, f in enumerate(fs):
        y, nrm, go, bw = outs[i]
        for which, acc in ((0, tf), (1, tb)):
            if only[i] == ('bwd', 'fwd')[which]:
                acc[i].append(float('nan'))
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                if which == 0:
                    f(x.data_ptr(), up.data_ptr(), b.data_ptr(), y.data_ptr(), nrm.data_ptr(), N, S, S, S, he, flags, 0.2, 1e-8, None, None, 0, None, st)
                else:
                    bw()
            e1.record()
            torch.cuda.synchronize()
            acc[i].append(e0.elapsed_time(e1) / 5)
for i, p in enumerate(sys.argv[1:]):
    a, c = sorted(tf[i]), sorted(tb[i])
    print(f'{p}: fwd median {a[len(a) // 2]:.4f} ms (min {a[0]:.4f}), bwd+prev median {c[len(c) // 2]:.4f} ms (min {c[0]:.4f})')
