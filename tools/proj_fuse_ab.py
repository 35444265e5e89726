#!/usr/bin/env python
"""A/B of the factor projection fused into the Winograd kernels of the last camera block (round 4) against the separate
launches, in ONE process at the headline shape (N = 8 hypotheses, 128^3 x 16):

  forward : lf_conv3d_c16_wino + lf_conv1x1_fwd            vs  lf_conv3d_c16_wino_projfwd
  backward: lf_conv1x1_bwd_data + lf_conv3d_c16_wino(bwd)  vs  lf_conv3d_c16_wino_projbwd
  loop    : pose iterations/s of the render-loop engine with fuse_projection = none / fwd / bwd / both

    python tools/proj_fuse_ab.py [--size 128] [--samples 8] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentfusion_amd import _lib, ops, synth  # noqa: E402
from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM, check  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=128)
ap.add_argument('--samples', type=int, default=8)
ap.add_argument('--rounds', type=int, default=7)
ap.add_argument('--no-loop', action='store_true')
ap.add_argument('--json', default=None)
a = ap.parse_args()
S, N = a.size, a.samples
DEV = 'cuda'
L = _lib.lib()
g = torch.Generator().manual_seed(0)
flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
x = ops.cl(torch.randn(N, 16, S, S, S, generator=g).to(DEV))
w1 = torch.randn(16, 16, 3, 3, 3, generator=g).to(DEV)
w = torch.randn(16, 16, 3, 3, 3, generator=g).to(DEV)
b = (torch.randn(16, generator=g) * 0.1).to(DEV)
wp = torch.randn(16, 16 * S, 1, 1, generator=g).to(DEV)
pb = (torch.randn(16, generator=g) * 0.1).to(DEV)
he, phe = ops.he_constant(w), ops.he_constant(wp)
wdm = wp.reshape(16, 16, S).permute(0, 2, 1).reshape(16, S * 16).contiguous()
up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)
ppack, ppack_t = ops.pack_conv1x1(wdm), ops.pack_conv1x1(wdm.t().contiguous())
wA, wtA = ops.pack_wino_proj(wdm), ops.pack_wino_proj(wdm, transpose=True)
act1, nrm1 = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w1), None, ops.he_constant(w1), flags)
gp = ops.cl(torch.randn(N, 16, S, S, generator=g).to(DEV))
st = torch.cuda.current_stream().cuda_stream


def fwd_two():
    y, n = ops.conv3d_c16_wino(act1, up, b, he, flags)
    zp = ops.empty_cl((N, 16, S, S), DEV)
    pn = ops._conv1x1_raw(y, ppack, pb, N, S * S, 16, S, S * S * S * 16, S * S * 16, 16, zp, phe, flags)
    return y, n, zp, pn


def fwd_one():
    return ops.conv3d_c16_wino_projfwd(act1, up, b, he, flags, wA, pb, phe, flags)


act2, nrm2, zp0, pn0 = fwd_two()
_, _, zp1, pn1 = fwd_one()
gvol = ops.empty_cl((N, 16, S, S, S), DEV)


def bwd_two():
    check(L.lf_conv1x1_bwd_data(gp.data_ptr(), ppack_t.data_ptr(), gvol.data_ptr(), N, S * S, 16, S * 16, S * S * S * 16, 16, 16,
                                S * S * 16, phe, act2.data_ptr(), nrm2.data_ptr(), flags, ops.SLOPE, None, st), 'bwd')
    return ops.conv3d_c16_wino(gvol, upt, None, he, 0, prev=(act1, nrm1, flags))[0]


def bwd_one():
    return ops.conv3d_c16_wino_projbwd(gp, wtA, phe, act2, nrm2, flags, upt, he, prev=(act1, nrm1, flags))


g2, g1 = bwd_two(), bwd_one()
torch.cuda.synchronize()
out = {'shape': [N, 16, S, S, S],
       'fwd_zp_bit_identical': bool(torch.equal(zp0, zp1) and torch.equal(pn0, pn1)),
       'fwd_zp_max_abs_diff': (zp0 - zp1).abs().max().item(),
       'bwd_max_abs_diff': (g2 - g1).abs().max().item(), 'bwd_max_abs': g2.abs().max().item()}
del g2, g1


def timeit(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def conv_fwd_alone():
    return ops.conv3d_c16_wino(act1, up, b, he, flags)


def conv_bwd_alone():
    return ops.conv3d_c16_wino(gvol, upt, None, he, 0, prev=(act1, nrm1, flags))


cases = {'fwd_conv_alone': conv_fwd_alone, 'fwd_two_launches': fwd_two, 'fwd_fused': fwd_one,
         'bwd_conv_alone': conv_bwd_alone, 'bwd_two_launches': bwd_two, 'bwd_fused': bwd_one}
times = {k: [] for k in cases}
for k, fn in cases.items():
    fn()
torch.cuda.synchronize()
for r in range(a.rounds):
    for k, fn in cases.items():
        times[k].append(timeit(fn))
out['ms'] = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
out['ms_min'] = {k: min(v) for k, v in times.items()}
del x, act1, act2, gvol
torch.cuda.empty_cache()

if not a.no_loop:
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation, utils as pu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model, _ = synth.build_model(S, 16, 'gru', seed=0, device=DEV)
    model.freeze()
    td = synth.make_observation_data(1, seed=200)
    target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(DEV)
    z_obj = torch.randn(1, 1, 16, S, S, S, generator=g).to(DEV)
    cfg = estimation._load_toml(os.path.join(root, 'configs', 'adam_quick.toml'))
    cfg['args']['num_samples'] = N
    cfg['args']['ranking_size'] = N
    torch.manual_seed(300)
    init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
    loop, first = {}, {}
    sels = {'none': False, 'fwd': ('fwd',), 'bwd': ('bwd',), 'both': True}
    states = {}
    for name, sel in sels.items():
        est = estimation.load_from_config(cfg, model, converge_patience=10 ** 6, fuse_projection=sel)
        stt = est.start(z_obj, target, init.zoom(None, model.input_size, model.camera_dist).to(DEV))
        with torch.no_grad():
            l0, g0 = stt['engine'].forward_backward(stt['cam'], need_grad=True)
        first[name] = (l0.cpu(), g0.cpu())
        for _ in range(3):
            est.iterate(stt)
        states[name] = (est, stt)
        loop[name] = []
    for r in range(5):
        for name, (est, stt) in states.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                est.iterate(stt)
            torch.cuda.synchronize()
            loop[name].append(20 / (time.perf_counter() - t0))
    out['loop_iters_per_s'] = {k: sorted(v)[len(v) // 2] for k, v in loop.items()}
    out['loop_iteration0_vs_unfused'] = {
        k: {'losses_identical': bool(torch.equal(first[k][0], first['none'][0])),
            'camera_grad_max_rel_l2': ((first[k][1] - first['none'][1]).norm(dim=1) / first['none'][1].norm(dim=1)).max().item()}
        for k in sels if k != 'none'}
print(json.dumps(out, indent=1))
if a.json:
    os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
    json.dump(out, open(a.json, 'w'), indent=1)
