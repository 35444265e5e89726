"""Generates tests/golden/g26_headline_trace.pt from the REAL reference at the HEADLINE shape (build container only).

    python oracle/make_golden_headline.py [T]          # T = pose iterations (default 100 = the adam_quick preset's length)
    python oracle/make_golden_headline.py T TARGET_SEED INIT_SEED NAME     # further seeds of the target frame / the initial
                                                                           # hypotheses -> tests/golden/NAME.pt (same object; the
                                                                           # latent volume is not stored again)
    python oracle/make_golden_headline.py 30 200 300 g26n_headline_trace_threads3 --threads 3
                                                                           # the NOISE-FLOOR CONTROL: the same reference run on
                                                                           # the same seeds as g26 with another thread count
                                                                           # (ATen's reductions split differently): the
                                                                           # reference-vs-reference deviation the loop
                                                                           # tolerances of the GPU tests are anchored to

BASELINE cfg 2 exactly as bench.py runs it: SYN(128,16) with the GRU fuser (weights seed 0), 16 reference views (seed 100),
one target frame (seed 200), 8 initial hypotheses (seed 300), configs/adam_quick.toml.  The REFERENCE (imported from
/root/reference through oracle/refharness.py) reconstructs the object and runs GradientPoseEstimator._optimize_camera
(reference pose/estimation.py:583-677) for T iterations with its convergence stop disabled, as bench.py does.

Inputs are pinned by seeds (latentfusion_amd.synth: plain torch RNG calls, no product compute), so the fixture holds reference
OUTPUTS only: per-iteration loss terms and camera parameters, argmin, the final ranking, a sub-sampled latent volume and the
iteration-0 renders.  Slow (about 40 s reconstruction + 12-15 s per iteration on 8 cores, 14 GB), hence its own script.
"""
import copy
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import refharness  # noqa: E402
from make_golden import cam_dict, save  # noqa: E402

S, C, V, N = 128, 16, 16, 8
SEED_MODEL, SEED_REF, SEED_TARGET, SEED_INIT = 0, 100, 200, 300


def main():
    global SEED_TARGET, SEED_INIT
    threads = os.cpu_count() or 8
    if '--threads' in sys.argv:
        i = sys.argv.index('--threads')
        threads = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    name = 'g26_headline_trace'
    if len(sys.argv) > 4:
        SEED_TARGET, SEED_INIT, name = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    lean = name != 'g26_headline_trace'
    refharness.load_reference()
    import tomli
    from latentfusion.modules.geometry import Camera
    from latentfusion.observation import Observation
    from latentfusion.pose import estimation, utils as pu
    from latentfusion.recon import fusion
    from latentfusion.recon.inference import LatentFusionModel
    from latentfusion.recon.models import Photographer, Sculptor
    from latentfusion_amd import synth                       # seeded inputs only: no product compute is used
    torch.set_num_threads(threads)

    sck, fck, pck, dist = synth.make_syn_checkpoints(S, C, 'gru', SEED_MODEL)
    model = LatentFusionModel(Sculptor.from_checkpoint(copy.deepcopy(sck)), fusion.from_checkpoint(copy.deepcopy(fck)),
                              Photographer.from_checkpoint(copy.deepcopy(pck)), dist, 'cpu')
    for m in (model.sculptor, model.fuser, model.photographer):
        for p in m.parameters():                             # weight gradients are never observed by the loop
            p.requires_grad_(False)

    def obs(n, seed):
        d = synth.make_observation_data(n, seed)
        return Observation(d['color'], d['depth'], d['mask'], Camera(d['intrinsic'], d['extrinsic'],
                                                                     width=d['width'], height=d['height']))
    ref_obs, target = obs(V, SEED_REF), obs(1, SEED_TARGET)
    t0 = time.time()
    with torch.no_grad():
        z_obj = model.build_latent_object(ref_obs)
    print(f'reference build: {time.time() - t0:.1f} s', flush=True)
    del ref_obs

    with open('/root/reference/configs/adam_quick.toml', 'rb') as f:
        cfg = tomli.load(f)
    cfg['args']['num_iters'] = T
    cfg['args']['num_samples'] = N
    cfg['args']['ranking_size'] = N
    cfg['args']['converge_patience'] = 10 ** 6               # bench.py times the loop with the convergence stop disabled
    torch.manual_seed(SEED_INIT)
    init = pu.sample_cameras_with_estimate(N, target.camera)
    zoomed = init.zoom(None, model.input_size, model.camera_dist)
    with torch.no_grad():
        y0, _ = model.render_latent_object(z_obj, zoomed, return_latent=True)
    est = estimation.load_from_config(copy.deepcopy(cfg), model, track_stats=True, return_camera_history=True)
    record = est._record_stat_dict

    def record_and_checkpoint(history, d):               # a partial trace survives an interrupted run (2 h on 8 cores at T = 100)
        record(history, d)
        n = history['rank_loss'].shape[0] if history['rank_loss'].dim() > 1 else 1
        if n % 10 == 0:
            os.makedirs('/tmp/g26', exist_ok=True)
            torch.save({k: v.clone() for k, v in history.items() if torch.is_tensor(v)}, '/tmp/g26/partial.pt')
            print(f'iteration {n}: {time.time() - t0:.0f} s', flush=True)
    est._record_stat_dict = record_and_checkpoint
    t0 = time.time()
    best, stats, hist = est.estimate(z_obj, target, camera=init)
    el = time.time() - t0
    print(f'reference loop: {T} iterations in {el:.1f} s ({el / T:.2f} s/iteration)', flush=True)
    rank = stats['rank_loss']
    out = {
        'S': S, 'C': C, 'V': V, 'N': N, 'T': T, 'cfg': cfg, 'camera_dist': dist,
        'seeds': {'model': SEED_MODEL, 'ref': SEED_REF, 'target': SEED_TARGET, 'init': SEED_INIT},
        'z_obj_sub': z_obj[..., ::4, ::4, ::4].clone(), 'z_obj_absmax': z_obj.abs().max(),
        'z_obj_channel_mean': z_obj.mean(dim=(0, 1, 3, 4, 5)).clone(),
        'z_obj_channel_sq': (z_obj ** 2).mean(dim=(0, 1, 3, 4, 5)).clone(),
        'init': cam_dict(init), 'init_zoomed_viewport': zoomed.viewport.clone(),
        'iter0': {k: y0[k].squeeze(0).clone() for k in ('depth_logits', 'mask_logits', 'depth', 'mask')},
        'rank_loss': rank.clone(), 'depth_loss': stats['depth_loss'].clone(), 'ov_depth_loss': stats['ov_depth_loss'].clone(),
        'iou_loss': stats['iou_loss'].clone(), 'mask_loss': stats['mask_loss'].clone(),
        'argmin': torch.argmin(rank, dim=1), 'order': torch.argsort(rank, dim=1),
        'hist_log_q': torch.stack([c.log_quaternion for _, c in hist]), 'hist_t': torch.stack([c.translation for _, c in hist]),
        'hist_viewport': torch.stack([c.viewport for _, c in hist]),
        'best': cam_dict(best), 'reference_seconds_per_iteration': el / T, 'reference_threads': torch.get_num_threads(),
    }
    if lean:
        for k in ('z_obj_sub', 'z_obj_channel_mean', 'z_obj_channel_sq'):
            del out[k]
    save(name, out)
    top2 = torch.sort(rank, dim=1).values
    print('argmin per iteration:', out['argmin'].tolist())
    print('min top-2 gap: %.3e' % float((top2[:, 1] - top2[:, 0]).min()))


if __name__ == '__main__':
    main()
