#!/usr/bin/env python
"""Workload for the PMC passes over the training-step kernels (tools/pmc_collect_train.sh): a calibration copy of known
size (1 GiB read + 1 GiB written), then at 8 x 128^3 x 16 in the bf16 STORAGE the autocast step uses (round 5): the bf16
ring convolution (forward with epilogue, and the addend form), the bf16 weight gradient, the 16-channel epilogue backward,
the camera->object gather and its deterministic splat, the fused lift passes and the ConvGRU backward stage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import _lib, ops, synth  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM  # noqa: E402
from latentfusion_amd.modules.geometry import Camera, c2o_coefficients  # noqa: E402
from latentfusion_amd.pose import utils as pu  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, C, S = 8, 16, 128
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
cl3 = torch.channels_last_3d
x32 = ops.cl(torch.randn(N, C, S, S, S, generator=g).cuda())
x = x32.to(torch.bfloat16).contiguous(memory_format=cl3)
gp = ops.cl((torch.randn(N, C, S, S, S, generator=g) * 1e-3).cuda()).to(torch.bfloat16).contiguous(memory_format=cl3)
w = torch.randn(16, 16, 3, 3, 3, generator=g).cuda()
b = torch.zeros(16).cuda()
he = ops.he_constant(w)
for _ in range(REP):
    y = x32.clone()            # calibration: 1 GiB in, 1 GiB out
torch.cuda.synchronize()
wp = ops.pack_conv3d_c16_ring_bf16(w)
for _ in range(REP):
    y, nrm = ops.conv3d_c16_ring_bf16_io(x, wp, b, he, LF_EPI_LRELU | LF_EPI_PIXELNORM, 1, out_bf16=True)      # <false,1,3>
torch.cuda.synchronize()
for _ in range(REP):
    ya, _ = ops.conv3d_c16_ring_bf16_io(x, wp, None, he, 0, 0, addend=gp, out_bf16=True)                       # <true,1,7>
torch.cuda.synchronize()
nb = L.lf_conv_bwd_weight_scratch_bytes(3, N, S, S, S, 16, 16)
scr = torch.empty(nb // 4 + 1, device='cuda')
gw = torch.empty(27, 16, 16, device='cuda')
for _ in range(REP):
    _lib.check(L.lf_conv_bwd_weight_bf16_io(x.data_ptr(), gp.data_ptr(), gw.data_ptr(), scr.data_ptr(), scr.numel() * 4, 3, N, S, S, S, 16, 16,
                                            he, 3, s), 'wgrad')                                                 # <3>
torch.cuda.synchronize()
for _ in range(REP):
    ops.epilogue_bwd_c16(gp, y, nrm, LF_EPI_LRELU | LF_EPI_PIXELNORM, True)                                    # <7>
torch.cuda.synchronize()
td = synth.make_observation_data(1, seed=2)
torch.manual_seed(3)
cams = pu.sample_cameras_with_estimate(N, Camera(td['intrinsic'], td['extrinsic'])).zoom(None, S, 2.85).to('cuda')
coef = c2o_coefficients(cams, 1.0).cuda()
with ops.autocast():
    for _ in range(REP):
        v = x.clone().requires_grad_(True)
        ops.resample_c2o(v, coef).backward(gp)                                                                 # gather <1,3>, splat <1,3>
torch.cuda.synchronize()
# fused lift passes at 8 views (rows [8 * 128^2][2048])
rows = torch.randn(N * S * S, C * S, generator=g).cuda()
vol = ops.empty_cl16((N, C, S, S, S), 'cuda', True)
norm = torch.empty(N * S * S, device='cuda')
gpr = torch.empty(N * S * S, C * S, device='cuda')
for _ in range(REP):
    _lib.check(L.lf_lift_norm_unfold(rows.data_ptr(), vol.data_ptr(), norm.data_ptr(), N, S * S, C, S, 1e-8, 1, s), 'lift fwd')
    _lib.check(L.lf_lift_bwd(gp.data_ptr(), vol.data_ptr(), norm.data_ptr(), gpr.data_ptr(), N, S * S, C, S, 0.2, 1, 3, s), 'lift bwd')
torch.cuda.synchronize()
# round 6: the one-group ring kernels with compile-time epilogues (data gradient = rounded form; in-place bf16 addend form), the
# fused lift and the renderer's fused 3-D -> 2-D projection (8 views of 128^2 pixels each)
from latentfusion_amd import ops_train  # noqa: E402
p1 = wp.reshape(1, 14, 16, 32)
o16 = torch.empty_like(x)
for _ in range(REP):
    ops_train.ring_multi(x, p1, he, [(o16, None, True)])                                                         # <1,true,0,6>
torch.cuda.synchronize()
for _ in range(REP):
    ops_train.ring_multi(x, p1, he, [(o16, o16, False)])                                                         # <1,true,0,11>
torch.cuda.synchronize()
nrm2 = torch.empty(N * S ** 3, device='cuda')
for _ in range(REP):
    ops_train.ring_multi(x, p1, he, [(o16, None, True)], extra=_lib.LF_RING_EX_BLOCK, e0=b, o2=nrm2)            # <1,true,4,6>
torch.cuda.synchronize()
xin = torch.randn(N * S * S, 16, generator=g).cuda()
wl = torch.randn(16 * S, 16, generator=g).cuda()
wtab = wl.reshape(16, S, 16).permute(1, 0, 2).contiguous().to(torch.bfloat16)
wtab_t = wl.reshape(16, S, 16).permute(1, 2, 0).contiguous().to(torch.bfloat16)
btab = torch.zeros(S, 16, device='cuda')
gx = torch.empty(N * S * S, 16, device='cuda')
gwl, gbl = torch.empty(16 * S, 16, device='cuda'), torch.empty(16 * S, device='cuda')
scr2 = torch.empty(L.lf_lift16_bwd_scratch_bytes(S) // 4 + 4, device='cuda')
for _ in range(REP):
    _lib.check(L.lf_lift16_fwd(xin.data_ptr(), wtab.data_ptr(), btab.data_ptr(), vol.data_ptr(), norm.data_ptr(), N * S * S, S * S, S, 0.35, 0.2,
                               1e-8, s), 'lift16 fwd')
    _lib.check(L.lf_lift16_bwd(gp.data_ptr(), vol.data_ptr(), norm.data_ptr(), xin.data_ptr(), wtab_t.data_ptr(), gx.data_ptr(), gwl.data_ptr(),
                               gbl.data_ptr(), scr2.data_ptr(), scr2.numel() * 4, N * S * S, S * S, S, 0.35, 0.2, 1, s), 'lift16 bwd')
torch.cuda.synchronize()
y2 = torch.empty(N * S * S, 16, device='cuda')
wp2 = torch.randn(16, 16 * S, generator=g).cuda()
ptab = wp2.reshape(16, 16, S).permute(2, 0, 1).contiguous().to(torch.bfloat16)
ptab_t = wp2.reshape(16, 16, S).permute(2, 1, 0).contiguous().to(torch.bfloat16)
gwp = torch.empty(16, 16 * S, device='cuda')
scr3 = torch.empty(L.lf_proj16_bwd_scratch_bytes(S) // 4 + 4, device='cuda')
for _ in range(REP):
    _lib.check(L.lf_proj16_fwd(x.data_ptr(), ptab.data_ptr(), None, y2.data_ptr(), norm.data_ptr(), N * S * S, S * S, S, 0.03, 0.2, 1e-8, s), 'proj16 fwd')
    _lib.check(L.lf_proj16_bwd(gx.data_ptr(), x.data_ptr(), ptab_t.data_ptr(), o16.data_ptr(), gwp.data_ptr(), scr3.data_ptr(), scr3.numel() * 4,
                               N * S * S, S * S, S, 0.03, s), 'proj16 bwd')
torch.cuda.synchronize()
# round 6: pointwise 16 -> 16 layer (bf16 records in and out; the data gradient with the bias sums)
wq = torch.randn(16, 16, generator=g).cuda().to(torch.bfloat16)
bq = torch.randn(16, generator=g).cuda()
gbq = torch.empty(16, device='cuda')
scr4 = torch.empty(L.lf_pw16_bwd_scratch_bytes(N * S ** 3) // 4 + 4, device='cuda')
for _ in range(REP):
    _lib.check(L.lf_pw16_fwd(x.data_ptr(), 1, wq.data_ptr(), bq.data_ptr(), 0.35, 0, 0.2, o16.data_ptr(), N * S ** 3, s), 'pw16 fwd')
    _lib.check(L.lf_pw16_bwd(x.data_ptr(), wq.data_ptr(), 0.35, o16.data_ptr(), 1, gbq.data_ptr(), scr4.data_ptr(), scr4.numel() * 4, N * S ** 3, s),
               'pw16 bwd')
torch.cuda.synchronize()
print('ok')
