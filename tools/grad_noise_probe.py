#!/usr/bin/env python
"""How far the camera gradients of one render + loss + backward are from an fp64 evaluation of the same arithmetic,
per conv3d variant of the engine, next to the reference's own fp32 arithmetic (the CPU oracle evaluated in fp32).
Uses the oracle -> measurement tool only (the product path never imports it).

    python tools/grad_noise_probe.py [S] [N]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle_util import WEIGHTS as W, noise_case, oracle_loss_grad  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
from latentfusion_amd.engine import RenderLoopEngine  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402
from latentfusion_amd.observation import Observation  # noqa: E402
case = noise_case(S, 16, N, device='cuda')
l32, g32 = oracle_loss_grad(torch.float32, case)
l64, g64 = oracle_loss_grad(torch.float64, case)
td, model = case['td'], case['model']
target = Observation(None, td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to('cuda')
cams = case['init'].zoom(None, model.input_size, model.camera_dist).to('cuda')


def err(g):
    e = (g.double() - g64).norm(dim=1) / g64.norm(dim=1)
    return {'max': e.max().item(), 'mean': e.mean().item(), 'per_sample': [round(v, 5) for v in e.tolist()]}


out = {'S': S, 'N': N, 'oracle_fp32': err(g32)}
for mode in ('winograd', 'fp32', 'f16x3', 'winograd_f16x3'):
    eng = RenderLoopEngine(model.photographer, case['z'].to('cuda'), target, W, conv_mode=mode)
    _, g = eng.forward_backward(cams)
    out[mode] = err(g.cpu())
print(json.dumps(out, indent=1))
