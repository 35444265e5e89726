#!/usr/bin/env python
"""cfg3 of SURVEY 8(d): the released architecture (tools/train/train.sh:28-66 -- 16^3 x 256-channel
latent volume, 256^2 images, GRU fuser) with random He-equalised weights, V = 8 synthetic reference
views, the cross_entropy_linemod preset (N = 128 renders/iteration, no gradient) and the adam_quick
preset.  Prints build time and iterations/sec of both loops.

    python tools/rel_probe.py [--views 8] [--ce-iters 10] [--adam-iters 20] [--wide fused|bmm|both]

--wide selects how the >= 64-channel 3x3(x3) convolutions run: 'fused' = Winograd input transform + lf_wino_fused_gemm
(this library's fp32-MFMA GEMM with the output transform and epilogue folded in), 'bmm' = the three-stage form with the
per-frequency products on the library GEMM; 'both' measures both in this process and checks that the cross-entropy
losses of the two agree (parity of what is timed: the architecture's GPU tests are tests/test_released_width_gpu.py).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel_model(device, seed=0):
    from latentfusion_amd import synth
    return synth.build_released_model(device, seed)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--ce-iters', type=int, default=10)
    ap.add_argument('--adam-iters', type=int, default=20)
    ap.add_argument('--wide', default='fused', choices=['fused', 'bmm', 'both'])
    a = ap.parse_args()
    dev = 'cuda:0'
    from latentfusion_amd import synth
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation, utils as pu
    model = rel_model(dev)
    n_par = sum(p.numel() for m in (model.sculptor, model.fuser, model.photographer) for p in m.parameters())
    ref = synth.make_observation(a.views, seed=100, device=dev)
    td = synth.make_observation_data(1, seed=200)
    target = Observation(td['color'], td['depth'], td['mask'], Camera(td['intrinsic'], td['extrinsic'])).to(dev)
    from latentfusion_amd import ops
    res = {'params_M': n_par / 1e6, 'views': a.views}
    check = {}
    for mode in (['fused', 'bmm'] if a.wide == 'both' else [a.wide]):
        ops.WIDE_CONV_MODE = mode
        out = {}
        for rep in range(2):                                   # second pass = warm (weight packs cached)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            z_obj = model.build_latent_object(ref)
            torch.cuda.synchronize(); out['t_build_s' if rep else 't_build_cold_s'] = time.perf_counter() - t0
        out['z_obj'] = list(z_obj.shape)

        torch.manual_seed(300)
        cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'cross_entropy_linemod.toml'))
        cfg['args']['num_iters'] = a.ce_iters
        est = estimation.load_from_config(cfg, model)
        # one evaluation on fixed cameras: the quantity both modes must agree on
        torch.manual_seed(301)
        cams = pu.sample_cameras_with_estimate(8, target.camera.to('cpu'), hemisphere=True, upright=True).to(dev)
        check[mode] = (z_obj.clone(), est.evaluate_samples(z_obj, target, cams)[1].clone())
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            est.estimate(z_obj, target, camera=target.camera)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            out['ce_it_per_s' if rep else 'ce_cold_it_per_s'] = a.ce_iters / el
        out['ce_renders_per_s'] = out['ce_it_per_s'] * cfg['args']['num_samples']

        cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_quick.toml'))
        cfg['args']['num_iters'] = a.adam_iters
        est = estimation.load_from_config(cfg, model, converge_patience=10 ** 6)
        torch.manual_seed(302)
        init8 = pu.sample_cameras_with_estimate(cfg['args']['num_samples'], target.camera.to('cpu'))
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            est.estimate(z_obj, target, camera=init8.to(dev))
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            out['adam_it_per_s' if rep else 'adam_cold_it_per_s'] = a.adam_iters / el
        # adam_latent (the notebook's fine stage: depth + overlap depth + latent term, 16 hypotheses): on the fused engine
        # (latent term included since round 3) and on the autograd-module path
        cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'adam_latent.toml'))
        cfg['args']['num_iters'] = max(4, a.adam_iters // 2)
        torch.manual_seed(303)
        init16 = pu.sample_cameras_with_estimate(cfg['args']['num_samples'], target.camera.to('cpu'))
        for tag, use_engine in (('adam_latent_engine_it_per_s', True), ('adam_latent_modules_it_per_s', False)):
            est = estimation.load_from_config(cfg, model, converge_patience=10 ** 6, use_engine=use_engine)
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                est.estimate(z_obj, target, camera=init16.to(dev))
                torch.cuda.synchronize(); el = time.perf_counter() - t0
            out[tag] = cfg['args']['num_iters'] / el
        # cross_entropy_latent (the notebook's coarse stage: latent term only, 96 = 24 x 4 flips renders per iteration, plus the
        # target's latent code under every sample): module path (the fused loss implements the gradient estimator's form)
        cfg = estimation._load_toml(os.path.join(ROOT, 'configs', 'cross_entropy_latent.toml'))
        cfg['args']['num_iters'] = max(3, a.ce_iters // 3)
        est = estimation.load_from_config(cfg, model)
        for rep in range(2):
            torch.manual_seed(304)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            est.estimate(z_obj, target, camera=target.camera)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        out['ce_latent_it_per_s'] = cfg['args']['num_iters'] / el
        res[mode] = out
    ops.WIDE_CONV_MODE = 'fused'
    if len(check) == 2:
        zf, lf_ = check['fused']
        zb, lb = check['bmm']
        res['fused_vs_bmm'] = {'z_obj_max_abs_diff': (zf - zb).abs().max().item(), 'z_obj_max_abs': zb.abs().max().item(),
                               'ce_loss_max_rel_diff': ((lf_ - lb).abs() / lb.abs().clamp_min(1e-12)).max().item(),
                               'ce_order_equal': bool(torch.equal(torch.argsort(lf_), torch.argsort(lb)))}
    out = res
    print(json.dumps(out))


if __name__ == '__main__':
    main()
