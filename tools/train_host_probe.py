#!/usr/bin/env python
"""Is the cfg 5 training step host-bound?  Time for run_iteration() to RETURN (everything enqueued) vs the time until the
device is idle, and the GPU-busy share from HIP events around the step."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latentfusion_amd import synth  # noqa: E402
from latentfusion_amd.recon import training  # noqa: E402

dev = 'cuda:0'
S = 128
model, _ = synth.build_model(S, 16, 'gru', seed=0, device=dev)
obs_in = model.preprocess_observation(synth.make_observation(32, seed=1, device=dev))
obs_out = model.preprocess_observation(synth.make_observation(8, seed=2, device=dev))
step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, g_depth_recon_loss_k=S * S // 4, use_amp=True)
batch = {'in': {'camera': obs_in.camera, 'image': obs_in.color.unsqueeze(0), 'mask': obs_in.mask.unsqueeze(0)},
         'out_gt': {'camera': obs_out.camera, 'depth': obs_out.depth.unsqueeze(0), 'mask': obs_out.mask.unsqueeze(0)}}
for _ in range(2):
    step.run_iteration(batch)
torch.cuda.synchronize()
rows = []
for _ in range(4):
    t0 = time.perf_counter()
    step.run_iteration(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append({'enqueue_ms': (t1 - t0) * 1e3, 'total_ms': (t2 - t0) * 1e3})
print(json.dumps(rows))
