"""BASELINE.json configurations at their STATED shapes on the HIP path (the other GPU tests use reduced sizes):

  cfg 3  the TRUE released architecture (reference tools/train/train.sh:28-66: 256^2 inputs, 16^3 x 256-channel volume,
         512-channel U-Net levels, GRU fuser, 68 M parameters), 8 reference views, cross_entropy_linemod -- against golden g25,
         which the REAL reference produced for the same seeded network (oracle/make_golden.py g25_released_arch);
  cfg 2  the 16-view reconstruction at 128^3 (GRU recurrence of 15 steps) -- against the CPU oracle on the same inputs, and
         the render + loss + camera gradients on the ORACLE-built volume;
  cfg 5  one generator training step at 32 + 8 views, 128^3, bf16 autocast -- properties (the oracle cannot run this size in
         test time): finite, run-to-run identical gradients, decreasing loss, bf16 gradient direction = fp32's.

North-star bar: rendered depth / mask within 1e-3 relative of the reference, identical loss order."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def close(a, b, atol=1e-4, rtol=1e-3):
    torch.testing.assert_close(a.detach().cpu().contiguous(), b.detach().cpu().contiguous(), atol=atol, rtol=rtol)


def prod_camera(d, device=DEV):
    from latentfusion_amd.modules.geometry import Camera
    return Camera(d['K'].to(device), None, d['z_span'], d['viewport'].to(device), width=d['width'],
                  height=d['height'], log_quaternion=d['log_q'].to(device), translation=d['t'].to(device))


def same_order_up_to_ties(loss, ref_loss, tol):
    """The HIP losses sort like the reference's wherever the reference separates two samples by more than tol."""
    loss, ref_loss = loss.detach().cpu(), ref_loss.detach().cpu()
    order = torch.argsort(ref_loss)
    return all(not (ref_loss[b] - ref_loss[a] > tol) or bool(loss[a] < loss[b]) for a, b in zip(order[:-1].tolist(), order[1:].tolist()))


def _observation(d, device=DEV):
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    return Observation(d['color'], d['depth'], d['mask'], Camera(d['intrinsic'], d['extrinsic'], width=d['width'],
                                                                  height=d['height'])).to(device)


# ----------------------------------------------------------------------------------------------------------------------
# cfg 3: the released architecture
# ----------------------------------------------------------------------------------------------------------------------
def test_cfg3_released_architecture_vs_reference(golden):
    """8-view build_latent_object + CrossEntropyPoseEstimator.evaluate_samples (4 cameras x 4 flips, linemod weights) of
    the 68 M-parameter released architecture on the HIP path == the real reference's outputs (golden g25): volume, rendered
    depth / mask logits (1e-3 relative), the four loss terms, the weighted loss and its ORDER; then the full preset
    (N = 128 = 32 x 4 flips, 2 iterations) twice: finite and run-to-run identical ranking."""
    from latentfusion_amd import synth
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.pose.loss import default_pose_loss
    g = golden('g25_released_arch')
    seed = g['seed']
    model, cks = synth.build_released_model(DEV, seed, 0.1)
    assert abs(cks[3] - g['camera_dist']) < 1e-9
    assert sum(p.numel() for p in model.parameters()) == g['params'] and g['params'] > 60e6
    ref = _observation(synth.make_observation_data(g['views'], seed + 10))
    target = _observation(synth.make_observation_data(1, seed + 20))
    z_obj = model.build_latent_object(ref)
    assert tuple(z_obj.shape) == (1, 1, 256, 16, 16, 16)
    scale = float(g['z_obj_absmax'])
    close(z_obj[..., ::2, ::2, ::2], g['z_obj_sub'], atol=1e-3 * scale, rtol=1e-3)
    close(z_obj.mean(dim=(0, 1, 3, 4, 5)), g['z_obj_channel_mean'], atol=2e-4)
    close((z_obj ** 2).mean(dim=(0, 1, 3, 4, 5)), g['z_obj_channel_sq'], atol=1e-3, rtol=1e-3)

    est = estimation.CrossEntropyPoseEstimator(model=model, num_samples=16, num_elites=6, num_iters=1, num_gmm_components=2,
                                               learning_rate=0.9, sample_flipped=True, ranking_size=4, loss_weights=g['weights'])
    cams, loss = est.evaluate_samples(z_obj, target, prod_camera(g['cams']))
    close(cams.log_quaternion, g['all_cams']['log_q'], atol=1e-5)
    close(loss, g['loss'], atol=2e-5, rtol=1e-3)
    assert same_order_up_to_ties(loss, g['loss'], 2e-5) and int(torch.argmin(loss)) == int(g['order'][0])
    with torch.no_grad():
        zd, zl, _, zc = est._render_observation(z_obj, cams)
        ld = default_pose_loss(target, zd, zl, zc)
    close(zc.viewport, g['zoom_viewport'], atol=1e-2)
    close(zd[..., ::4, ::4], g['depth_crop_sub'], atol=1e-4, rtol=1e-3)
    close(zl[..., ::4, ::4], g['mask_logits_crop_sub'], atol=1e-3 * float(g['mask_logits_crop_sub'].abs().max()), rtol=1e-3)
    for k in ('depth', 'ov_depth', 'iou', 'mask'):
        close(ld[k], g['loss_terms'][k], atol=5e-5, rtol=1e-3)

    # the whole preset at its stated size: 32 samples x 4 flips = 128 renders per iteration
    cfg = estimation._load_toml(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs',
                                             'cross_entropy_linemod.toml'))
    assert cfg['args']['num_samples'] == 128
    cfg['args']['num_iters'] = 2
    runs = []
    for _ in range(2):
        torch.manual_seed(7)
        import numpy as np
        np.random.seed(7)
        est = estimation.load_from_config(cfg, model)
        best = est.estimate(z_obj, target, camera=target.camera)
        runs.append(torch.cat((best.log_quaternion, best.translation), dim=1).cpu())
    assert torch.isfinite(runs[0]).all() and len(runs[0]) == cfg['args']['ranking_size']
    assert torch.equal(runs[0], runs[1])


# ----------------------------------------------------------------------------------------------------------------------
# cfg 2: the headline workload's reconstruction at full size
# ----------------------------------------------------------------------------------------------------------------------
def test_cfg2_build_at_full_size_vs_oracle():
    """SYN(128,16), 16 reference views, GRU fuser: the fused latent volume of the HIP path (16 per-view encodes, camera ->
    object resampling, 15 ConvGRU steps of 9 Winograd launches each) against the CPU oracle's (reference recon/models.py:198-258,
    recon/fusion.py:152-201), then iteration 0 of the adam_quick loop -- losses and camera gradients -- on the ORACLE-built
    volume, so the render parity no longer rests on the GPU build."""
    from lf_oracle import pose as opose
    import lf_oracle as O
    from latentfusion_amd import synth
    from latentfusion_amd.pose import estimation, utils as pu
    S, C, V, N = 128, 16, 16, 8
    model, cks = synth.build_model(S, C, 'gru', seed=0, device=DEV)
    rd, td = synth.make_observation_data(V, seed=100), synth.make_observation_data(1, seed=200)
    z_hip = model.build_latent_object(_observation(rd))
    torch.set_num_threads(min(32, os.cpu_count()))
    om = opose.Model(*cks)
    z_ora = om.build_latent_object(opose.Obs(rd['color'], rd['depth'], rd['mask'], O.Cam.from_extrinsic(rd['intrinsic'], rd['extrinsic'])))
    scale = z_ora.abs().max().item()
    diff = (z_hip.cpu() - z_ora).abs()
    rel_l2 = ((z_hip.cpu() - z_ora).norm() / z_ora.norm()).item()
    assert diff.max().item() <= 1e-3 * scale and rel_l2 <= 1e-4, (diff.max().item(), scale, rel_l2)

    target = _observation(td)
    cfg = estimation._load_toml(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'adam_quick.toml'))
    cfg['args']['num_samples'] = cfg['args']['ranking_size'] = N
    torch.manual_seed(300)
    init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
    est = estimation.load_from_config(cfg, model, converge_patience=10 ** 6)
    st = est.start(z_ora.to(DEV), target, init.zoom(None, model.input_size, model.camera_dist).to(DEV))
    with torch.no_grad():
        l0, g0 = st['engine'].forward_backward(st['cam'], need_grad=True)
    c = dict(cfg, args=dict(cfg['args'], num_iters=1, converge_patience=10 ** 6))
    otarget = opose.Obs(None, td['depth'], td['mask'], O.Cam.from_extrinsic(td['intrinsic'], td['extrinsic']))
    _, first = opose.gradient_estimate(om, z_ora, otarget, O.Cam(init.intrinsic.clone(), init.log_quaternion.clone(),
                                                                 init.translation.clone()), c)
    ref_loss = first['rank_loss'][0]
    ref_grad = torch.cat((first['grad_log_q'][0], first['grad_t'][0], first['grad_viewport'][0]), dim=1)
    close(l0[:, 4], ref_loss, atol=1e-5, rtol=1e-3)
    assert torch.equal(torch.argsort(l0[:, 4].cpu()), torch.argsort(ref_loss))
    gerr = ((g0.cpu() - ref_grad).norm(dim=1) / ref_grad.norm(dim=1).clamp_min(1e-30)).max().item()
    assert gerr < 1e-2, gerr                      # the fp32 noise floor of these gradients at 128^3 (test_oracle_golden)

    # what the north star states: the RENDERED depth / mask of the N = 8 hypotheses, pixel by pixel at the headline size
    # (reference recon/models.py:455-484: logits -> tanh / sigmoid, thresholded mask applied to the depth), within 1e-3 rel
    ocam = O.Cam(init.intrinsic.clone(), init.log_quaternion.clone(), init.translation.clone()).zoom(None, om.input_size, om.camera_dist)
    with torch.no_grad():
        yo, _ = om.render_latent_object(z_ora, ocam, apply_mask=True)
        pred, _ = model.render_latent_object(z_ora.to(DEV), st['cam'], return_latent=True, apply_mask=True)
    for key in ('depth_logits', 'mask_logits'):
        a, b = pred[key].squeeze(0).cpu(), yo[key].squeeze(0)
        assert a.shape == b.shape == (N, 1, S, S)
        close(a, b, atol=1e-4 * max(1.0, b.abs().max().item()), rtol=1e-3)
    # thresholded outputs: identical wherever the oracle's mask logit is not within fp32 noise of the threshold
    mo, mh = yo['mask'].squeeze(0), pred['mask'].squeeze(0).cpu()
    clear = yo['mask_logits'].squeeze(0).abs() > 1e-3
    assert clear.float().mean().item() > 0.99
    assert torch.equal((mh > 0.5)[clear], (mo > 0.5)[clear])
    close(mh, mo, atol=1e-4, rtol=1e-3)
    same = ((mh > 0.5) == (mo > 0.5))
    close(pred['depth'].squeeze(0).cpu()[same], yo['depth'].squeeze(0)[same], atol=1e-4, rtol=1e-3)
    # the four terms of default_pose_loss, not only their weighted sum (reference pose/estimation.py:70-118)
    ld = opose.pose_loss(otarget, ocam.denormalize_depth(yo['depth'].squeeze(0)), yo['mask_logits'].squeeze(0), ocam)
    for i, k in enumerate(('depth', 'ov_depth', 'iou', 'mask')):
        close(l0[:, i], ld[k], atol=1e-5, rtol=1e-3)


def test_mid_size_loop_trace_vs_oracle():
    """adam_quick on SYN(32,16), 4 views, N = 8, TWENTY iterations against the CPU oracle (reference pose/estimation.py:
    618-632, the per-iteration ranking): identical argmin pose index at every iteration, the first iterations' losses to
    1e-4, the rest within the envelope of the 16^3 golden trace (g7) -- Adam turns 1e-5-sized viewport gradients into
    lr-sized steps, so fp32 noise moves late iterations by ~one optimiser step."""
    from lf_oracle import pose as opose
    import lf_oracle as O
    from latentfusion_amd import synth
    from latentfusion_amd.pose import estimation, utils as pu
    S, C, V, N, IT = 32, 16, 4, 8, 20
    model, cks = synth.build_model(S, C, 'gru', seed=2, device=DEV)
    rd, td = synth.make_observation_data(V, seed=110), synth.make_observation_data(1, seed=210)
    om = opose.Model(*cks)
    z_ora = om.build_latent_object(opose.Obs(rd['color'], rd['depth'], rd['mask'], O.Cam.from_extrinsic(rd['intrinsic'], rd['extrinsic'])))
    z_hip = model.build_latent_object(_observation(rd))
    assert (z_hip.cpu() - z_ora).abs().max().item() <= 1e-3 * z_ora.abs().max().item()
    target = _observation(td)
    cfg = estimation._load_toml(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'adam_quick.toml'))
    cfg['args'].update(num_samples=N, ranking_size=N, num_iters=IT, converge_patience=10 ** 6)
    torch.manual_seed(301)
    init = pu.sample_cameras_with_estimate(N, target.camera.to('cpu'))
    otarget = opose.Obs(None, td['depth'], td['mask'], O.Cam.from_extrinsic(td['intrinsic'], td['extrinsic']))
    _, tr = opose.gradient_estimate(om, z_ora, otarget, O.Cam(init.intrinsic.clone(), init.log_quaternion.clone(),
                                                              init.translation.clone()), cfg)
    ref = tr['rank_loss']                                         # (IT, N)
    assert ref.shape == (IT, N)
    for z in (z_ora.to(DEV), z_hip):                               # render parity alone, then end to end
        est = estimation.load_from_config(cfg, model, track_stats=True)
        _, stats = est.estimate(z, target, camera=init)
        got = stats['rank_loss']
        assert got.shape == (IT, N)
        close(got[:3], ref[:3], atol=1e-5, rtol=1e-4)
        close(got, ref, atol=3e-2, rtol=0)
        top2 = torch.sort(ref, dim=1).values[:, :2]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-3 * top2[:, 0].abs()
        assert clear.sum().item() >= IT - 2, clear
        assert torch.equal(torch.argmin(got, dim=1)[clear], torch.argmin(ref, dim=1)[clear])


# ----------------------------------------------------------------------------------------------------------------------
# cfg 5: the training step at its stated shape
# ----------------------------------------------------------------------------------------------------------------------
def test_cfg5_training_step_at_full_shape_bf16():
    """One generator step of tools/train/train_reconstruct.py:421-535 at BASELINE cfg 5's shape -- 32 input + 8 output views,
    SYN(128,16), bf16 autocast (GeneratorStep(use_amp=True)): every loss and gradient finite; the gradient of two runs on the
    same batch bit-identical (fixed-point splat, fixed-order weight-gradient reductions); the loss falls over three
    optimiser steps; the bf16 gradient points where the fp32 gradient of the same batch points (cosine >= 0.95: the
    SYN(16,16) figure against the oracle under torch.autocast is 0.963, tests/test_autocast_gpu.py)."""
    from latentfusion_amd import synth
    from latentfusion_amd.recon import training
    S, C, VI, VO = 128, 16, 32, 8
    model, _ = synth.build_model(S, C, 'gru', seed=0, device=DEV)
    obs_in = model.preprocess_observation(synth.make_observation(VI, seed=1, device=DEV))
    obs_out = model.preprocess_observation(synth.make_observation(VO, seed=2, device=DEV))
    batch = {'in': {'camera': obs_in.camera, 'image': obs_in.color.unsqueeze(0), 'mask': obs_in.mask.unsqueeze(0)},
             'out_gt': {'camera': obs_out.camera, 'depth': obs_out.depth.unsqueeze(0), 'mask': obs_out.mask.unsqueeze(0)}}
    step = training.GeneratorStep(model.sculptor, model.fuser, model.photographer, g_depth_recon_loss_k=S * S // 4, use_amp=True)
    grads = []
    for _ in range(2):
        out = step.run_iteration(batch, is_step=False)
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in out.values())
        grads.append(step.flat.grad.clone())
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1])
    # fp32 gradient of the same batch, same weights
    step.use_amp = False
    step.run_iteration(batch, is_step=False)
    g32 = step.flat.grad.clone()
    cos = torch.nn.functional.cosine_similarity(grads[0].double(), g32.double(), dim=0).item()
    assert cos >= 0.95, cos
    step.use_amp = True
    del grads, g32
    torch.cuda.empty_cache()
    losses = [float(step.run_iteration(batch)['total']) for _ in range(4)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    assert torch.cuda.max_memory_allocated() < 200 * 2 ** 30


# ----------------------------------------------------------------------------------------------------------------------
# cfg 2 against the REFERENCE ITSELF at the headline shape, over the preset's real length (golden g26)
# ----------------------------------------------------------------------------------------------------------------------
def test_cfg2_headline_loop_vs_reference_fixture():
    """tests/golden/g26_headline_trace.pt holds what the imported REFERENCE produced for BASELINE cfg 2 exactly as bench.py runs
    it (oracle/make_golden_headline.py: SYN(128,16), GRU, 16 views, N = 8, all 100 iterations of adam_quick; 49 min on 8 cores).
    HIP end to end against it (tools/headline_trace_probe.py; measured numbers: profiles/r05_headline_trace_vs_reference.json):
      * the reconstructed volume (sub-sampled) and the iteration-0 renders: 1e-3 relative (measured 3e-6 / 2e-6 rel-L2);
      * the loop: fp32 rounding differences are amplified by Adam's first steps (1e-7 -> 4e-3 over iterations 0-8) and then
        stay BOUNDED (<= 5e-3 relative over all 100 iterations; asserted against the reference's OWN thread-count noise,
        fixture g26n, see below); the argmin pose index is identical for the
        first 10 iterations (measured: 14) and at EVERY iteration where the reference separates its best two hypotheses by
        more than twice that iteration's loss deviation; the final best loss agrees to 1e-3 (measured 1e-4).
    The reference's own top-2 gap falls below 1e-3 relative in 66 of the 100 iterations (down to 4e-7): there the index of
    the best hypothesis is decided by rounding in either implementation -- 'identical argmin' is not a property the
    reference has against itself on another BLAS."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('headline_trace_probe', os.path.join(root, 'tools', 'headline_trace_probe.py'))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    res = probe.compare(DEV)
    assert res['fixture']['T'] >= 20 and res['fixture']['S'] == 128 and res['fixture']['V'] == 16
    assert res['volume']['max_abs_diff_over_absmax'] <= 1e-4 and res['volume']['rel_l2'] <= 1e-4, res['volume']
    for k, r in res['iteration0_renders'].items():
        assert r['max_abs_diff'] <= 1e-3 * max(1.0, r['max_abs_ref']) and r['rel_l2'] <= 1e-4, (k, r)
    t = res['trace']
    rows = t['per_iteration']
    assert t['iterations'] == res['fixture']['T']
    # ---- tolerances ANCHORED to the reference's own noise floor (round 6): g26n = the same reference run on the same seeds with
    # 3 instead of 8 threads (oracle/make_golden_headline.py --threads 3; 32 iterations).  The reference deviates from ITSELF by
    # 2e-7 -> 1.1e-3 over iterations 1-9 and up to 2.4e-3 after (its argmin differs at iterations 13, 14, 28).  HIP must stay
    # within 8 x the control's deviation, taken over a +-2-iteration window (Adam amplifies a rounding difference
    # exponentially at first, so single iterations of the control dip: 4e-5 at iteration 7 between 1e-4 and 7e-4);
    # measured: 5.3 x (Winograd kernels; 4.1 x with the direct fp32 convolution, profiles/r06_headline_trace_vs_control.json).
    sd = t['reference_self_deviation']
    ctl = sd['per_iteration']
    assert sd['control_threads'] != res['fixture']['reference_threads'] and len(ctl) >= 30
    for i, r in enumerate(rows):
        floor = max(ctl[max(0, i - 2):i + 3]) if i < len(ctl) else max(ctl)
        assert r['rank_loss_max_rel_diff'] <= 8.0 * floor + 1e-6, (i, r['rank_loss_max_rel_diff'], floor)
    assert sd['max_hip_dev_over_windowed_ref_dev'] <= 8.0
    assert all(r['argmin_equal'] for r in rows[:10]), t['first_iteration_argmin_differs']
    # argmin: asserted wherever BOTH reference runs agree on it and the reference separates its best two hypotheses by more
    # than twice that iteration's HIP deviation.  (Where the two references agree but the gap is BELOW the deviation -- the
    # cross-over of hypotheses 3 and 7, iterations 26-30: gap 1.2e-3 -> 6e-5, the control itself switches one iteration before
    # the fixture -- the index is decided by rounding; those iterations are reported, not asserted.)
    decided = [r for r in rows if r['reference_top2_rel_gap'] > 2.0 * r['rank_loss_max_rel_diff']
               and r.get('references_agree_on_argmin', True)]
    assert len(decided) >= 5 and all(r["argmin_equal"] for r in decided), [r['iteration'] for r in decided if not r['argmin_equal']]
    assert abs(t['final_best_loss_hip'] - t['final_best_loss_reference']) <= 1e-3 * abs(t['final_best_loss_reference'])


@pytest.mark.parametrize('name', ['g26b_headline_trace_seed201', 'g26c_headline_trace_seed202'])
def test_cfg2_headline_loop_vs_reference_further_seeds(name):
    """The same comparison for two further seeds of the target frame and of the initial hypotheses (same object; 30 reference
    iterations each, oracle/make_golden_headline.py 30 <target seed> <init seed> <name>): how the deviation of the loop and the
    argmin agreement behave is a property of the loop, not of g26's seed.  Numbers: profiles/r05_headline_trace_further_seeds.json."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('headline_trace_probe', os.path.join(root, 'tools', 'headline_trace_probe.py'))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    res = probe.compare(DEV, name)
    assert res['fixture']['T'] >= 20 and res['fixture']['S'] == 128 and res['fixture']['V'] == 16
    for k, r in res['iteration0_renders'].items():
        assert r['max_abs_diff'] <= 1e-3 * max(1.0, r['max_abs_ref']) and r['rel_l2'] <= 1e-4, (k, r)
    t = res['trace']
    rows = t['per_iteration']
    assert t['iterations'] == res['fixture']['T']
    assert t['max_rel_diff_first_5'] <= 1e-3 and t['max_rel_diff_all'] <= 2e-2, (t['max_rel_diff_first_5'], t['max_rel_diff_all'])
    decided = [r for r in rows if r['reference_top2_rel_gap'] > 2.0 * r['rank_loss_max_rel_diff']]
    assert len(decided) >= 5 and all(r["argmin_equal"] for r in decided), [r['iteration'] for r in decided if not r['argmin_equal']]
    assert abs(t['final_best_loss_hip'] - t['final_best_loss_reference']) <= 2e-3 * abs(t['final_best_loss_reference'])
