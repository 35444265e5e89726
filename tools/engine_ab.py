#!/usr/bin/env python
"""A/B of engine-level options in ONE process at the headline workload (box-to-box variance is ~1.5 %): the adam_quick loop on
SYN(128,16), N = 8, timed round-robin in blocks of 20 iterations.

    python tools/engine_ab.py [--json out.json]
Variants: explicit 2-D decoder on / off (RenderLoopEngine.EXPLICIT_DECODER), fused projection none / fwd."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latentfusion_amd import synth  # noqa: E402
from latentfusion_amd.engine import RenderLoopEngine  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402
from latentfusion_amd.observation import Observation  # noqa: E402
from latentfusion_amd.pose import estimation, utils as pu  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--json', default=None)
ap.add_argument('--rounds', type=int, default=7)
a = ap.parse_args()
S, N, DEV = 128, 8, 'cuda'
model, _ = synth.build_model(S, 16, 'gru', seed=0, device=DEV)
model.freeze()
td = synth.make_observation_data(1, seed=200)
target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(DEV)
z_obj = torch.randn(1, 1, 16, S, S, S, generator=torch.Generator().manual_seed(0)).to(DEV)
cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_quick.toml'))
cfg['args']['num_samples'] = cfg['args']['ranking_size'] = N
torch.manual_seed(300)
init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
variants = {'explicit decoder, fused fwd projection (default)': (True, None),
            'autograd decoder, fused fwd projection': (False, None),
            'explicit decoder, separate projection': (True, False),
            'autograd decoder, separate projection (round 3)': (False, False)}
states, first = {}, {}
for name, (expl, fuse) in variants.items():
    RenderLoopEngine.EXPLICIT_DECODER = expl
    est = estimation.load_from_config(cfg, model, converge_patience=10 ** 6, fuse_projection=fuse)
    st = est.start(z_obj, target, init.zoom(None, model.input_size, model.camera_dist).to(DEV))
    assert (st['engine'].dec is not None) == expl
    with torch.no_grad():
        l0, g0 = st['engine'].forward_backward(st['cam'], need_grad=True)
    first[name] = (l0.cpu(), g0.cpu())
    for _ in range(3):
        est.iterate(st)
    states[name] = (est, st)
RenderLoopEngine.EXPLICIT_DECODER = True
rates = {k: [] for k in variants}
for r in range(a.rounds):
    for name, (est, st) in states.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            est.iterate(st)
        torch.cuda.synchronize()
        rates[name].append(20 / (time.perf_counter() - t0))
base = 'autograd decoder, separate projection (round 3)'
out = {'iters_per_s_median': {k: sorted(v)[len(v) // 2] for k, v in rates.items()},
       'iteration0_vs_round3_form': {k: {'losses_max_abs_diff': (first[k][0] - first[base][0]).abs().max().item(),
                                         'camera_grad_max_rel_l2': ((first[k][1] - first[base][1]).norm(dim=1)
                                                                    / first[base][1].norm(dim=1)).max().item()} for k in variants}}
print(json.dumps(out, indent=1))
if a.json:
    os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
    json.dump(out, open(a.json, 'w'), indent=1)
