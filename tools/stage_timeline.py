#!/usr/bin/env python
"""Phase timeline of the LDS-staged resamplers (csrc/resample_staged.inc) from s_memtime stamps.

    hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -shared -DSTAGE_TS -DSTAGED_TPW_OVERRIDE=1 \
          -Ilatentfusion_amd/csrc -Iinclude latentfusion_amd/csrc/resample.hip -o scratch/rs_ts.so
    python tools/stage_timeline.py scratch/rs_ts.so

Thread 0 of every 64th workgroup stamps the phase boundaries of its tile; printed: median cycles between consecutive stamps
over those workgroups, at the bench shape (N = 8, 128^3 x 16, the bench's sampled cameras), for the gather and the gradient."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import consts, engine, ops, synth  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402
from latentfusion_amd.pose import utils as pu  # noqa: E402
from latentfusion_amd.recon.utils import optimal_camera_dist  # noqa: E402

P, I = ctypes.c_void_p, ctypes.c_int
S, C, N = 128, 16, 8
dev = 'cuda'
g = torch.Generator().manual_seed(0)
z = ops.cl(torch.randn(1, C, S, S, S, generator=g).to(dev))
gout = ops.cl(torch.randn(N, C, S, S, S, generator=g).to(dev))
tdata = synth.make_observation_data(1, seed=200)
torch.manual_seed(300)
dist = optimal_camera_dist(consts.INTRINSIC[1][1], S, 0.5, slack=128 / S)
cam = pu.sample_cameras_with_estimate(N, Camera(tdata['intrinsic'], tdata['extrinsic'])).zoom(None, S, dist).to(dev)
cf20 = torch.zeros(N, 20, device=dev)
cf20[:, :18] = engine.camera_coefs(cam, 1.0, S, S).detach()[:, :18]
st = torch.cuda.current_stream().cuda_stream
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
L.lf_set_tuning.restype = I
L.lf_set_tuning.argtypes = [I, I]
L.lf_resample3d_fwd.restype = I
L.lf_resample3d_fwd.argtypes = [P, I, P, I, P, I, I, I, I, I, P]
L.lf_resample3d_bwd_coef_scratch_bytes.restype = ctypes.c_size_t
L.lf_resample3d_bwd_coef_scratch_bytes.argtypes = [I, I, I, I]
L.lf_resample3d_bwd_coef.restype = I
L.lf_resample3d_bwd_coef.argtypes = [P, P, I, P, P, P, ctypes.c_size_t, I, I, I, I, I, P]
L.lf_debug_stage_ts.restype = I
L.lf_debug_stage_ts.argtypes = [P]
L.lf_set_tuning(1, 4)
out = torch.empty_like(gout)
nb = L.lf_resample3d_bwd_coef_scratch_bytes(N, S, S, S)
scratch = torch.empty(nb // 4 + 1, device=dev)
gc = torch.empty(N, 18, device=dev)
NAMES = ['tap->', 'origin', 'atomics', 'barrier1', 'scan', 'barrier2a+offsets+recmap', 'barrier2', 'load', 'barrier3', 'compute']


def stamps():
    buf = np.zeros((256, 16), dtype=np.uint64)
    assert L.lf_debug_stage_ts(buf.ctypes.data) == 0
    return buf.astype(np.int64)


for name, fn in (('gather', lambda: L.lf_resample3d_fwd(z.data_ptr(), 1, cf20.data_ptr(), 0, out.data_ptr(), N, S, S, S, C, st)),
                 ('gradient', lambda: L.lf_resample3d_bwd_coef(gout.data_ptr(), z.data_ptr(), 1, cf20.data_ptr(), gc.data_ptr(),
                                                               scratch.data_ptr(), scratch.numel() * 4, N, S, S, S, C, st))):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    ts = stamps()
    ok = ts[:, 9] > ts[:, 0]
    d = np.diff(ts[ok][:, :10], axis=1)
    print(f'{name}: {ok.sum()} stamped workgroups; median cycles per phase (tile start -> end of compute: '
          f'{int(np.median(ts[ok][:, 9] - ts[ok][:, 0]))})')
    for i in range(9):
        print(f'   {NAMES[i + 1]:>24s}: median {int(np.median(d[:, i])):6d}   p90 {int(np.percentile(d[:, i], 90)):6d}')
