#!/usr/bin/env python
"""A/B timing of builds of resample.hip in ONE process: O2C gather (lf_resample3d_fwd) and its coefficient
gradient (lf_resample3d_bwd_coef) at the bench shape (N = 8, 128^3 x 16), checked against the first variant.

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared -Ilatentfusion_amd/csrc -Iinclude \
          latentfusion_amd/csrc/resample.hip -o scratch/rs_a.so
    python tools/resample_ab.py scratch/rs_a.so scratch/rs_b.so
    python tools/resample_ab.py latentfusion_amd/csrc/liblf_hip.so@1 latentfusion_amd/csrc/liblf_hip.so@2
(`path@v` selects kernel variant v of that library through lf_set_tuning(1, v) before every call)
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import ops, synth  # noqa: E402
from latentfusion_amd import engine  # noqa: E402
from latentfusion_amd.pose import utils as pu  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402

P, I = ctypes.c_void_p, ctypes.c_int
S, C, N, ROUNDS = int(os.environ.get('AB_S', 128)), int(os.environ.get('AB_C', 16)), 8, 7
dev = 'cuda'
g = torch.Generator().manual_seed(0)
z = ops.cl(torch.randn(1, C, S, S, S, generator=g).to(dev))
gout = ops.cl(torch.randn(N, C, S, S, S, generator=g).to(dev))
# realistic coefficient blocks: the bench's sampled cameras around a target pose
tdata = synth.make_observation_data(1, seed=200)
torch.manual_seed(300)
from latentfusion_amd.recon.utils import optimal_camera_dist  # noqa: E402
from latentfusion_amd import consts  # noqa: E402
dist = optimal_camera_dist(consts.INTRINSIC[1][1], S, 0.5, slack=128 / S)
cam = pu.sample_cameras_with_estimate(N, Camera(tdata['intrinsic'], tdata['extrinsic'])).zoom(None, S, dist).to(dev)
coefs = engine.camera_coefs(cam, 1.0, S, S).detach()
cf20 = torch.zeros(N, 20, device=dev)
cf20[:, :18] = coefs[:, :18]
st = torch.cuda.current_stream().cuda_stream


def bind(spec):
    path, _, var = spec.partition('@')
    L = ctypes.CDLL(os.path.abspath(path))
    var, _, var2 = var.partition('.')                      # path@<resampler variant>[.<coefficient-gradient variant>]
    L._variant = int(var) if var else None
    L._variant2 = int(var2) if var2 else None
    if L._variant is not None:
        L.lf_set_tuning.restype = I
        L.lf_set_tuning.argtypes = [I, I]
    L.lf_resample3d_fwd.restype = I
    L.lf_resample3d_fwd.argtypes = [P, I, P, I, P, I, I, I, I, I, P]
    L.lf_resample3d_bwd_coef_scratch_bytes.restype = ctypes.c_size_t
    L.lf_resample3d_bwd_coef_scratch_bytes.argtypes = [I, I, I, I]
    L.lf_resample3d_bwd_coef.restype = I
    L.lf_resample3d_bwd_coef.argtypes = [P, P, I, P, P, P, ctypes.c_size_t, I, I, I, I, I, P]
    return L


def select(L):
    if L._variant is not None:
        L.lf_set_tuning(1, L._variant)
    if L._variant2 is not None:
        L.lf_set_tuning(2, L._variant2)


libs = [bind(p) for p in sys.argv[1:]]
state = []
for L in libs:
    select(L)
    out = torch.empty_like(gout)
    nb = L.lf_resample3d_bwd_coef_scratch_bytes(N, S, S, S)
    scratch = torch.empty(nb // 4 + 1, device=dev)
    gc = torch.empty(N, 18, device=dev)
    fwd = lambda L=L, out=out: L.lf_resample3d_fwd(z.data_ptr(), 1, cf20.data_ptr(), 0, out.data_ptr(), N, S, S, S, C, st)  # noqa: E731
    bwd = lambda L=L, scratch=scratch, gc=gc: L.lf_resample3d_bwd_coef(gout.data_ptr(), z.data_ptr(), 1, cf20.data_ptr(), gc.data_ptr(),  # noqa: E731
                                                                       scratch.data_ptr(), scratch.numel() * 4, N, S, S, S, C, st)
    assert fwd() == 0 and bwd() == 0
    torch.cuda.synchronize()
    state.append((out, gc, fwd, bwd))
for i, p in enumerate(sys.argv[1:]):
    rel = ((state[i][1] - state[0][1]).abs() / state[0][1].abs().clamp_min(1e-3)).max().item()
    print(f'{p}: fwd diff vs first {(state[i][0] - state[0][0]).abs().max().item():.2e}, coef-grad max rel diff {rel:.2e}')
tf, tb = [[] for _ in libs], [[] for _ in libs]
for r in range(ROUNDS):
    for i in range(len(libs)):
        for fn, acc in ((state[i][2], tf), (state[i][3], tb)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            select(libs[i])
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            acc[i].append(e0.elapsed_time(e1) / 5)
for i, p in enumerate(sys.argv[1:]):
    a, c = sorted(tf[i]), sorted(tb[i])
    print(f'{p}: O2C fwd median {a[len(a) // 2]:.4f} ms (min {a[0]:.4f}), coef-grad median {c[len(c) // 2]:.4f} ms (min {c[0]:.4f})')
