#!/usr/bin/env python
"""Per-phase timeline of one workgroup of the split-precision direct conv kernel (cycle-counter stamps written by lane 0
of every wave when conv_split.hip is compiled with -DSPLIT_ABL=32):

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -shared -DSPLIT_ABL=32 -Ilatentfusion_amd/csrc \
          -Iinclude latentfusion_amd/csrc/conv_split.hip -o scratch/split_ts.so
    python tools/split_timeline.py scratch/split_ts.so
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402

P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
g = torch.Generator().manual_seed(0)
S, N = 128, 8
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = (torch.randn(16, generator=g) * 0.1).cuda()
he = ops.he_constant(w)
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
L.lf_conv3d_c16_split.restype = I
L.lf_conv3d_c16_split.argtypes = [P, P, P, P, P, I, I, I, I, F, U, F, F, P, P, U, P, P, P]
table = (I * 28)()
L.lf_conv3d_c16_split_pairs(table)
taps = w.reshape(16, 16, 27)
k = torch.zeros(14, 16, 32, device='cuda')
for p in range(14):
    for sel in range(2):
        if table[2 * p + sel] >= 0:
            k[p, :, sel * 16:(sel + 1) * 16] = taps[:, :, table[2 * p + sel]]
hi = k.half()
wp = torch.stack((hi, (k - hi.float()).half()), dim=1).contiguous()
y = torch.empty_like(x)
nrm = torch.zeros(N * S ** 3, device='cuda')
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
for _ in range(3):
    assert L.lf_conv3d_c16_split(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), nrm.data_ptr(), N, S, S, S, he, flags, 0.2,
                                 1e-8, None, None, 0, None, None, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
T = int(os.environ.get("TL_T", 128))
ts = nrm[:T * 4 * 8].cpu().numpy().view(np.uint32).reshape(T, 4, 8).astype(np.int64)
# stamps: 0 top of tile, 1 halo loads issued, 2 MFMA phase (+ previous tile's epilogue) done, 3 staged loads landed,
# 4 converted and written to LDS, 6 past the barrier
seg = [('fetch issue', 0, 1), ('MFMA phase + previous epilogue', 1, 2), ('vmcnt(0) wait', 2, 3), ('convert + LDS write', 3, 4),
       ('barrier', 4, 6)]
sel = [i for i in range(8, T - 8) if i % 64 not in (62, 63, 0)]           # skip the column changes
for wv in range(4):
    d = [np.mean((ts[sel, wv, b_] - ts[sel, wv, a_]) & 0xffffffff) for _, a_, b_ in seg]
    nxt = np.mean((ts[[i + 1 for i in sel], wv, 0] - ts[sel, wv, 6]) & 0xffffffff)
    tot = np.mean((ts[[i + 1 for i in sel], wv, 0] - ts[sel, wv, 0]) & 0xffffffff)
    print(f'wave {wv}: total {tot:.0f} ticks/tile: ' + ', '.join(f'{n} {v:.0f}' for (n, _, _), v in zip(seg, d)) + f', to next top {nxt:.0f}')
