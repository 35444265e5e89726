#!/usr/bin/env python
"""A/B of lf_set_tuning keys on the cfg 5 training step (32 + 8 views, SYN(128,16), bf16 autocast): step time per setting.
    python tools/train_tune_ab.py KEY VALUE [KEY VALUE ...]   (each pair measured after the default)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latentfusion_amd import _lib, synth  # noqa: E402
from latentfusion_amd.recon import training  # noqa: E402

dev = 'cuda:0'
S = 128
model, _ = synth.build_model(S, 16, 'gru', seed=0, device=dev)
obs_in = model.preprocess_observation(synth.make_observation(32, seed=1, device=dev))
obs_out = model.preprocess_observation(synth.make_observation(8, seed=2, device=dev))
step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, g_depth_recon_loss_k=S * S // 4, use_amp=True)
batch = {'in': {'camera': obs_in.camera, 'image': obs_in.color.unsqueeze(0), 'mask': obs_in.mask.unsqueeze(0)},
         'out_gt': {'camera': obs_out.camera, 'depth': obs_out.depth.unsqueeze(0), 'mask': obs_out.mask.unsqueeze(0)}}


def measure(k=4):
    step.run_iteration(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step.run_iteration(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


L = _lib.lib()
out = {'default_ms': measure()}
args = [int(a) for a in sys.argv[1:]]
for key, val in zip(args[0::2], args[1::2]):
    prev = L.lf_set_tuning(key, val)
    out[f'key{key}={val}_ms'] = measure()
    L.lf_set_tuning(key, prev)
out['default_again_ms'] = measure()
print(json.dumps(out))
