#!/usr/bin/env python
"""Per-phase timeline of one workgroup of the Winograd conv kernel (s_memtime stamps written by lane 0 of
every wave when the kernel is compiled with -DWINO_ABL=16):

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared -DWINO_ABL=16 \
          latentfusion_amd/csrc/conv_wino.hip -o /tmp/wino_ts.so
    python tools/wino_timeline.py /tmp/wino_ts.so
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
P = ctypes.c_void_p
g = torch.Generator().manual_seed(0)
S, N = 128, 8
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).cuda())
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = (torch.randn(16, generator=g) * 0.1).cuda()
he = ops.he_constant(w)
up = ops.pack_conv3d_c16_wino(w)
y = torch.empty_like(x); nrm = torch.zeros(N * S ** 3, device='cuda')
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
L = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else '/tmp/wino_ts.so')
f = L.lf_conv3d_c16_wino
f.restype = ctypes.c_int
f.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint, ctypes.c_float, ctypes.c_float, P, P, ctypes.c_uint, P, P]
for _ in range(3):
    assert f(x.data_ptr(), up.data_ptr(), b.data_ptr(), y.data_ptr(), nrm.data_ptr(), N, S, S, S, he, flags, 0.2, 1e-8, None, None, 0, None, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
ts = nrm[:128 * 4 * 16].cpu().numpy().view(np.uint32).reshape(128, 4, 16).astype(np.int64)
names = ['compute+Pwrite', 'B1 wait', 'DMA issue', 'idx+prev loads', '-', 'Pread+math', 'vmcnt wait', 'stores', 'B2 wait(to next top)']
for wv in (0, 1, 3):
    d = []
    for k in range(8):
        d.append(np.mean((ts[8:120, wv, k + 1] - ts[8:120, wv, k]) & 0xffffffff))
    d.append(np.mean((ts[9:121, wv, 0] - ts[8:120, wv, 8]) & 0xffffffff))
    tot = np.mean((ts[9:121, wv, 0] - ts[8:120, wv, 0]) & 0xffffffff)
    print(f'wave {wv}: total {tot:.0f} ticks/tile: ' + ', '.join(f'{n} {v:.0f}' for n, v in zip(names, d)))
    c = [np.mean((ts[8:120, wv, b] - ts[8:120, wv, a]) & 0xffffffff) for a, b in ((0, 9), (9, 10), (10, 11), (11, 12))]
    print('         compute split: ' + ', '.join(f'{n} {v:.0f}' for n, v in zip(('g0 load+xform', 'g0 MFMA+out', 'g1 load+xform', 'g1 MFMA+out'), c)))
