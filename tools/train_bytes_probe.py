#!/usr/bin/env python
"""Algorithmic bytes of the cfg-5 training step per library entry point (latentfusion_amd._lib.BYTE_LOG: per launch, the sizes
of its input / output tensors, each once, scratch excluded) next to the measured step time -> the step's HBM floor.

    python tools/train_bytes_probe.py [--views-in 32] [--views-out 8] [--steps 3] [out.json]
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
    ap.add_argument('--views-in', type=int, default=32)
    ap.add_argument('--views-out', type=int, default=8)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('out', nargs='?')
    a = ap.parse_args()
    dev = 'cuda:0'
    from latentfusion_amd import _lib, synth
    from latentfusion_amd.recon import training
    S = a.size
    model, _ = synth.build_model(S, 16, 'gru', seed=0, device=dev)
    obs_in = model.preprocess_observation(synth.make_observation(a.views_in, seed=1, device=dev))
    obs_out = model.preprocess_observation(synth.make_observation(a.views_out, seed=2, device=dev))
    step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, g_depth_recon_loss_k=S * S // 4, use_amp=True)
    batch = {'in': {'camera': obs_in.camera, 'image': obs_in.color.unsqueeze(0), 'mask': obs_in.mask.unsqueeze(0)},
             'out_gt': {'camera': obs_out.camera, 'depth': obs_out.depth.unsqueeze(0), 'mask': obs_out.mask.unsqueeze(0)}}
    times = []
    for i in range(a.steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step.run_iteration(batch)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    ms = sorted(times[1:])[(len(times) - 2) // 2] * 1e3
    _lib.BYTE_LOG = {}
    step.run_iteration(batch)
    torch.cuda.synchronize()
    log, _lib.BYTE_LOG = _lib.BYTE_LOG, None
    total = float(sum(v[1] for v in log.values()))
    out = {'ms_per_step': ms, 'algorithmic_GB_per_step': total / 1e9, 'hbm_floor_ms': total / 8e12 * 1e3, 'frac': total / 8e12 * 1e3 / ms,
           'launches': int(sum(v[0] for v in log.values())), 'peak_mem_GB': torch.cuda.max_memory_allocated() / 2 ** 30,
           'entry_points': {k: {'launches': v[0], 'GB': v[1] / 1e9} for k, v in sorted(log.items(), key=lambda kv: -kv[1][1])}}
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main()
