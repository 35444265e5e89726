#!/usr/bin/env python
"""A/B of builds of wino_fused.hip in one process at the cfg 3 shape (3-D 256 -> 256 on 16^3, N = 128 hypotheses):

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared -Ilatentfusion_amd/csrc [-DWF_ABL_XFORM=1] \
          latentfusion_amd/csrc/wino_fused.hip -o scratch/wf_a.so
    python tools/wino_fused_ab.py scratch/wf_base.so scratch/wf_xform.so

Times lf_wino_fused_gemm (median of 7 rounds x 3 launches) per variant and, from the product library, the stand-alone
input transform lf_wino3d_input_transform it would replace."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import _lib  # noqa: E402

P = ctypes.c_void_p
N, S, C = 128, 16, 256
T = N * (S // 2) ** 3
g = torch.Generator().manual_seed(0)
x = torch.randn(N, S, S, S, C, generator=g).cuda()
V = torch.empty(64, T, C, device='cuda')
U2 = torch.randn(64, C, C, generator=g).cuda() * 0.05
bias = torch.zeros(C, device='cuda')
y = torch.empty(N, S, S, S, C, device='cuda')
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.lf_wino3d_input_transform(x.data_ptr(), V.data_ptr(), N, S, S, S, C, st), 'input transform')
nscr = L.lf_wino_fused_scratch_bytes(3, N, S, S, S, C)
scr = torch.empty(max(nscr, 4) // 4, device='cuda')


def timeit(fn, k=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


ti = sorted(timeit(lambda: L.lf_wino3d_input_transform(x.data_ptr(), V.data_ptr(), N, S, S, S, C, st)) for _ in range(7))
print(f'lf_wino3d_input_transform (product library): median {ti[3]:.3f} ms')
fs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    f = lib.lf_wino_fused_gemm
    f.restype = ctypes.c_int
    f.argtypes = [P, P, P, P, P, ctypes.c_size_t] + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_uint, ctypes.c_float, P]
    fs.append((path, lambda f=f: f(V.data_ptr(), U2.data_ptr(), bias.data_ptr(), y.data_ptr(), scr.data_ptr(), nscr, 3, N, S, S, S, C, C,
                                   0.02, 1, 0.2, st)))
times = {p: [] for p, _ in fs}
for _ in range(7):
    for p, fn in fs:
        assert fn() == 0
        times[p].append(timeit(fn))
fl = 2.0 * 64 * T * C * C
for p, _ in fs:
    t = sorted(times[p])
    print(f'{p}: lf_wino_fused_gemm median {t[3]:.3f} ms (min {t[0]:.3f}) = {fl / t[3] / 1e9:.1f} TFLOP/s fp32 MFMA')
