#!/usr/bin/env python
"""Pose iterations/s of the renderer VARIANTS of the reference on the fused engine vs the generic module path (VERDICT r03
item 5): 'factor' (the headline), 'sum' projection, occlusion module (reference recon/models.py:378-395,427-437), at
SYN(S,16), N = 8 hypotheses, adam_quick.  Random He-equalised weights (constructor initialisation under a seed).

    python tools/variant_probe.py [--size 128] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latentfusion_amd import consts, synth  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402
from latentfusion_amd.observation import Observation  # noqa: E402
from latentfusion_amd.pose import estimation, utils as pu  # noqa: E402
from latentfusion_amd.recon.models import Photographer  # noqa: E402
from latentfusion_amd.recon.utils import optimal_camera_dist  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=128)
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--json', default=None)
ap.add_argument('--only', default=None, help='one variant (factor | sum | occlusion)')
ap.add_argument('--engine-only', action='store_true')
a = ap.parse_args()
S, C, N, DEV = a.size, 16, 8, 'cuda'
VARIANTS = {
    'factor': dict(projection_type='factor', object_config=[], occlusion_config=False, image_config=[[16, 32], [32, 16]]),
    'sum': dict(projection_type='sum', object_config=[], occlusion_config=False, image_config=[[16, 32], [32, 16]]),
    'occlusion': dict(projection_type='factor', object_config=[16, 16], occlusion_config=[[17, 16], [16, 16]],
                      image_config=[[16, 32], [32, 16]]),
}
td = synth.make_observation_data(1, seed=200)
target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(DEV)
z_obj = torch.randn(1, 1, C, S, S, S, generator=torch.Generator().manual_seed(0)).to(DEV)
dist = optimal_camera_dist(consts.INTRINSIC[1][1], S, 0.5, slack=128 / S)
cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_quick.toml'))
cfg['args']['num_samples'] = cfg['args']['ranking_size'] = N
torch.manual_seed(300)
init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
out = {'shape': f'SYN({S},{C}), N = {N}, adam_quick', 'variants': {}}
for name, kw in VARIANTS.items():
    if a.only and name != a.only:
        continue
    torch.manual_seed(1)
    ph = Photographer(in_size=S, camera_config=[C, C], predict_color=False, predict_depth=True, predict_mask=True,
                      scale_mode='nearest', cube_size=1.0, **kw).to(DEV)
    for p in ph.parameters():
        p.requires_grad_(False)

    class M:                                                      # the facade surface the estimator touches
        photographer, device, input_size, camera_dist = ph, torch.device(DEV), S, dist

        @staticmethod
        def render_latent_object(z, cam, return_latent=True, apply_mask=True):
            y, zl, _ = ph.decode(z, cam, return_latent=return_latent, apply_mask=apply_mask)
            return y, (zl.squeeze(0) if return_latent else zl)
    res = {}
    for mode, use_engine in ((('engine', True),) if a.engine_only else (('engine', True), ('modules', False))):
        est = estimation.load_from_config(cfg, M, converge_patience=10 ** 6, use_engine=use_engine)
        st = est.start(z_obj, target, init.zoom(None, S, dist).to(DEV))
        assert ('engine' in st) == use_engine, (name, mode)
        for _ in range(2):
            est.iterate(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            est.iterate(st)
        torch.cuda.synchronize()
        res[mode + '_iters_per_s'] = a.iters / (time.perf_counter() - t0)
        res[mode + '_best_loss'] = st['ranking'][0][1]
        del est, st
        torch.cuda.empty_cache()
    if not a.engine_only:
        res['speedup'] = res['engine_iters_per_s'] / res['modules_iters_per_s']
    out['variants'][name] = res
    del ph
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
if a.json:
    os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
    json.dump(out, open(a.json, 'w'), indent=1)
