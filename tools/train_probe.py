#!/usr/bin/env python
"""cfg5 of SURVEY 8(d) in fp32 on one GPU: one generator training step (encode V_in views, decode V_out
views, losses, backward incl. weight gradients, flat Adam) on SYN(S,C) with synthetic observations.

    python tools/train_probe.py [--size 128] [--channels 16] [--views-in 8] [--views-out 8] [--steps 3] [--amp]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--channels', type=int, default=16)
    ap.add_argument('--views-in', type=int, default=8)
    ap.add_argument('--views-out', type=int, default=8)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--fuser', default='gru')
    ap.add_argument('--amp', action='store_true', help='bf16 autocast policy (GeneratorStep(use_amp=True))')
    a = ap.parse_args()
    dev = 'cuda:0'
    from latentfusion_amd import synth
    from latentfusion_amd.recon import training
    S = a.size
    model, _ = synth.build_model(S, a.channels, a.fuser, seed=0, device=dev)
    obs_in = model.preprocess_observation(synth.make_observation(a.views_in, seed=1, device=dev))
    obs_out = model.preprocess_observation(synth.make_observation(a.views_out, seed=2, device=dev))
    step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, g_depth_recon_loss_k=S * S // 4, use_amp=a.amp)
    batch = {'in': {'camera': obs_in.camera, 'image': obs_in.color.unsqueeze(0), 'mask': obs_in.mask.unsqueeze(0)},
             'out_gt': {'camera': obs_out.camera, 'depth': obs_out.depth.unsqueeze(0), 'mask': obs_out.mask.unsqueeze(0)}}
    out = {'params_M': sum(q.numel() for q in step.flat.params) / 1e6, 'views_in': a.views_in, 'views_out': a.views_out, 'size': S, 'amp': a.amp}
    times, losses = [], []
    for i in range(a.steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        l = step.run_iteration(batch)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        losses.append(float(l['total']))
    out['step_s_first'] = times[0]
    out['step_s'] = sum(times[1:]) / len(times[1:])
    out['loss'] = losses
    out['peak_mem_GB'] = torch.cuda.max_memory_allocated() / 2 ** 30
    print(json.dumps(out))


if __name__ == '__main__':
    main()
