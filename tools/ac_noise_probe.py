#!/usr/bin/env python
"""Gradient noise of the bf16 autocast step at SYN(16,16) (the workload of tests/test_autocast_gpu.py): cosine similarity of
every parameter gradient with the ORACLE's fp32 gradient, for the oracle under torch.autocast(cpu, bf16), the HIP step with
fp32 storage (ops.BF16_STORAGE = False) and with bf16 storage (default)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import lf_oracle as O  # noqa: E402
from lf_oracle import nets  # noqa: E402
from latentfusion_amd import losses as L, ops, synth  # noqa: E402
from latentfusion_amd.modules.geometry import Camera  # noqa: E402
from latentfusion_amd.observation import Observation  # noqa: E402
from latentfusion_amd.recon import training  # noqa: E402

DEV = 'cuda'
S, C = 16, 16
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kw = dict(g_depth_recon_loss_type='l1', g_depth_recon_loss_weight=25.0, g_mask_recon_loss_weight=25.0, generator_lr=1e-3)


def hip_grads(storage):
    ops.BF16_STORAGE = storage
    model, (sck, fck, pck, dist) = synth.build_model(S, C, 'gru', seed=seed, device=DEV, bias_std=0.1)
    d = synth.make_observation_data(3, seed=7)
    obs = model.preprocess_observation(Observation(d['color'], d['depth'], d['mask'], Camera(d['intrinsic'], d['extrinsic'])).to(DEV))
    gen = torch.Generator().manual_seed(5)
    tgt_depth = torch.rand(1, 3, 1, S, S, generator=gen) * 2 - 1
    tgt_mask = (torch.rand(1, 3, 1, S, S, generator=gen) > 0.5).float()
    step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, use_amp=True, **kw)
    batch = {'in': {'camera': obs.camera, 'image': obs.color.unsqueeze(0), 'mask': obs.mask.unsqueeze(0)},
             'out_gt': {'camera': obs.camera, 'depth': tgt_depth.to(DEV), 'mask': tgt_mask.to(DEV)}}
    got = step.run_iteration(batch, is_step=False)
    mods = {'s': model.sculptor, 'f': model.fuser, 'p': model.photographer}
    g = {(k, n): p.grad.detach().cpu().clone() for k, m in mods.items() for n, p in m.named_parameters()}
    ops.BF16_STORAGE = True
    return float(got['total']), g, (sck, fck, pck), obs, tgt_depth, tgt_mask


tot_a, g_a, cks_, obs, tgt_depth, tgt_mask = hip_grads(False)
tot_b, g_b, *_ = hip_grads(True)
sck, fck, pck = cks_
cks = {k: {**ck, 'state_dict': {n: v.clone().requires_grad_(True) for n, v in ck.get('state_dict', {}).items()}}
       for k, ck in (('s', sck), ('f', fck), ('p', pck))}
cam = obs.camera.to('cpu')
ocam = O.Cam(cam.intrinsic, cam.log_quaternion, cam.translation, viewport=cam.viewport, z_span=cam.z_span, width=cam.width, height=cam.height)


def oracle_loss(autocast):
    for ck in cks.values():
        for v in ck['state_dict'].values():
            v.grad = None
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
        z = nets.encode(cks['s'], cks['f'], ocam, obs.color.cpu(), obs.depth.cpu(), obs.mask.cpu())
        y, _, _ = nets.decode(cks['p'], z, ocam, apply_mask=False)
    tot = 25.0 * L.reduce_loss(L.get_recon_criterion('l1')(y['depth'].float(), tgt_depth)) + \
        25.0 * L.reduce_loss(L.get_recon_criterion('binary_cross_entropy')(y['mask_logits'].float(), tgt_mask))
    tot.backward()
    return float(tot.detach()), {(k, n): v.grad.clone() for k, ck in cks.items() for n, v in ck['state_dict'].items() if v.grad is not None}


w32, g32 = oracle_loss(False)
w16, g16 = oracle_loss(True)
print(f'loss: oracle fp32 {w32:.5f}  oracle bf16 {w16:.5f}  HIP fp32-storage {tot_a:.5f}  HIP bf16-storage {tot_b:.5f}')
cos = lambda u, v: F.cosine_similarity(u.reshape(1, -1).double(), v.reshape(1, -1).double()).item()   # noqa: E731
alls = {'orc': [], 'a': [], 'b': [], 'ref': []}
print(f'{"parameter":58s} {"oracle16":>9s} {"hip f32st":>9s} {"hip b16st":>9s}')
for key, ref in g32.items():
    alls['orc'].append(g16[key].reshape(-1)); alls['a'].append(g_a[key].reshape(-1)); alls['b'].append(g_b[key].reshape(-1)); alls['ref'].append(ref.reshape(-1))
    print(f'{key[0] + "." + key[1]:58s} {cos(g16[key], ref):9.4f} {cos(g_a[key], ref):9.4f} {cos(g_b[key], ref):9.4f}')
cat = {k: torch.cat(v) for k, v in alls.items()}
print(f'{"ALL":58s} {cos(cat["orc"], cat["ref"]):9.4f} {cos(cat["a"], cat["ref"]):9.4f} {cos(cat["b"], cat["ref"]):9.4f}')
