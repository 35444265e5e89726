#!/usr/bin/env python
"""One-GPU measurements behind the expected multi-GPU numbers of DESIGN section 6 (no multi-GPU box was available to this
build): the adam_quick iteration at N = 1, 2, 4, 8 hypotheses of ONE object (what a rank of the hypothesis-sharded loop
runs), and the 16-view GRU build split into its per-view encodes (shardable over the ranks) and its 15 recurrent steps
(serial).  Prints JSON.

    python tools/scale_paper_probe.py [out.json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latentfusion_amd import synth  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402
from latentfusion_amd.observation import Observation  # noqa: E402
from latentfusion_amd.pose import estimation, utils as pu  # noqa: E402
from latentfusion_amd.three.batchview import b2bv  # noqa: E402

dev = 'cuda:0'
S, C, V = 128, 16, 16
model, _ = synth.build_model(S, C, 'gru', seed=0, device=dev)
model.freeze()
rd, td = synth.make_observation_data(V, seed=100), synth.make_observation_data(1, seed=200)
ref = Observation(rd['color'], rd['depth'], rd['mask'], Camera(rd['intrinsic'], rd['extrinsic'], width=rd['width'], height=rd['height'])).to(dev)
target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(dev)


def timed(fn, k=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, r


out = {'shape': f'SYN({S},{C}), {V} views, GRU fuser'}
with torch.no_grad():
    obs = model.preprocess_observation(ref)
    sc = model.sculptor
    x = torch.cat((obs.color, obs.mask * 2.0 - 1.0), dim=1)
    t_enc, zc = timed(lambda: sc(x, obs.camera)[0])
    z_views = b2bv(zc, V)
    t_fuse, _ = timed(lambda: model.fuser(z_views, None, None, obs.camera)[0])
    t_enc2, _ = timed(lambda: sc(x[:2], obs.camera[:2])[0])
    t_build, z_obj = timed(lambda: model.build_latent_object(ref))
out['build'] = {'encode_16_views_ms': t_enc, 'encode_2_views_ms': t_enc2, 'gru_15_steps_ms': t_fuse, 'build_latent_object_ms': t_build,
                'gru_step_ms': t_fuse / (V - 1)}
cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_quick.toml'))
loop = {}
for N in (1, 2, 4, 8):
    c = dict(cfg, args=dict(cfg['args'], num_samples=N, ranking_size=N))
    est = estimation.load_from_config(c, model, converge_patience=10 ** 6)
    torch.manual_seed(300)
    init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
    st = est.start(z_obj, target, init.zoom(None, model.input_size, model.camera_dist).to(dev))
    for _ in range(3):
        est.iterate(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        est.iterate(st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    loop[f'N={N}'] = {'ms_per_iteration': ms, 'iterations_per_s': 1e3 / ms, 'ms_per_hypothesis': ms / N}
out['pose_loop_one_object'] = loop
txt = json.dumps(out, indent=1)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write(txt)
