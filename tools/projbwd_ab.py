#!/usr/bin/env python
"""A/B timing of several builds of conv_wino.hip's fused projection-backward kernel (lf_conv3d_c16_wino_projbwd) in ONE
process at the headline shape, next to the plain data-gradient launch of the same build:

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared -Ilatentfusion_amd/csrc \
          [-DWINO_PJ_ABL=k] latentfusion_amd/csrc/conv_wino.hip -o scratch/v.so        (one per variant)
    python tools/projbwd_ab.py scratch/a.so scratch/b.so ...

Every variant is compared with the first (max |diff| -- ablation builds are wrong on purpose) and timed round-robin."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
S, N, ROUNDS = 128, 8, 7
g = torch.Generator().manual_seed(0)
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda())
w1 = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
wp = torch.randn(16, 16 * S, 1, 1, generator=g).cuda()
he, phe = ops.he_constant(w), ops.he_constant(wp)
wdm = wp.reshape(16, 16, S).permute(0, 2, 1).reshape(16, S * 16).contiguous()
upt = ops.pack_conv3d_c16_wino(w, transpose=True)
wtA = ops.pack_wino_proj(wdm, transpose=True)
act1, nrm1 = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w1), None, ops.he_constant(w1), flags)
act2, nrm2 = ops.conv3d_c16_wino(act1, ops.pack_conv3d_c16_wino(w), None, he, flags)
gp = ops.cl(torch.randn(N, 16, S, S, generator=g).cuda())
st = torch.cuda.current_stream().cuda_stream


def bind(path):
    L = ctypes.CDLL(os.path.abspath(path))
    f = L.lf_conv3d_c16_wino_projbwd
    f.restype = I
    f.argtypes = [P, P, F, P, P, U, P, P, I, I, I, I, F, F, P, P, U, P]
    c = L.lf_conv3d_c16_wino
    c.restype = I
    c.argtypes = [P, P, P, P, P, I, I, I, I, F, U, F, F, P, P, U, P, P]
    return f, c


def call(f, out):
    return f(gp.data_ptr(), wtA.data_ptr(), phe, act2.data_ptr(), nrm2.data_ptr(), flags, upt.data_ptr(), out.data_ptr(),
             N, S, S, S, he, 0.2, act1.data_ptr(), nrm1.data_ptr(), flags, st)


def plain(c, out):
    return c(x.data_ptr(), upt.data_ptr(), None, out.data_ptr(), None, N, S, S, S, he, 0, 0.2, 1e-8, act1.data_ptr(),
             nrm1.data_ptr(), flags, None, st)


paths = sys.argv[1:]
fs = [bind(p) for p in paths]
outs = [torch.empty_like(x) for _ in fs]
for (f, c), o in zip(fs, outs):
    assert call(f, o) == 0
torch.cuda.synchronize()
for p, o in zip(paths, outs):
    print(f'{p}: max |diff| vs first {(o - outs[0]).abs().max().item():.3e}')
tf = [[] for _ in fs]
tp = [[] for _ in fs]
for r in range(ROUNDS):
    for i, (f, c) in enumerate(fs):
        for fn, acc in ((lambda: call(f, outs[i]), tf), (lambda: plain(c, outs[i]), tp)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            acc[i].append(e0.elapsed_time(e1) / 5)
for i, p in enumerate(paths):
    print(f'{p}: projbwd {sorted(tf[i])[ROUNDS // 2]:.4f} ms (min {min(tf[i]):.4f}); plain data-gradient conv of the same build '
          f'{sorted(tp[i])[ROUNDS // 2]:.4f} ms')
