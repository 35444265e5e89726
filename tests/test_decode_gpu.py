"""End-to-end GPU parity of the renderer (Photographer.decode) and the encoder (Sculptor.encode
+ fusers) on the HIP path against golden vectors produced by the real reference.

North-star tolerance: rendered depth / mask within 1e-3 relative of the reference."""
import pytest
import torch

import lf_oracle as O
from lf_oracle import nets

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def close(a, b, atol=1e-4, rtol=1e-3):
    torch.testing.assert_close(a.detach().cpu().contiguous(), b.detach().cpu().contiguous(), atol=atol, rtol=rtol)


def prod_camera(d, device=DEV):
    from latentfusion_amd.modules.geometry import Camera
    return Camera(d['K'].to(device), None, d['z_span'], d['viewport'].to(device), width=d['width'],
                  height=d['height'], log_quaternion=d['log_q'].to(device), translation=d['t'].to(device))


@pytest.mark.parametrize('variant', ['factor', 'sum', 'occlusion'])
def test_g5_decode(golden, variant):
    from latentfusion_amd.recon.models import Photographer
    r = golden('g5_decode')[variant]
    ph = Photographer.from_checkpoint(r['ck']).to(DEV)
    cam = prod_camera(r['cam'])
    for p in (cam.log_quaternion, cam.translation, cam.viewport):
        p.requires_grad_(True)
    y, lat, zd = ph.decode(r['z_obj'].to(DEV), cam, return_latent=True, apply_mask=True)
    for k in ('depth_logits', 'mask_logits', 'depth', 'mask'):
        close(y[k], r['y'][k], atol=1e-4, rtol=1e-3)
    close(lat, r['latent'], atol=1e-4, rtol=1e-3)
    if r['z_depth'] is not None:
        close(zd, r['z_depth'], atol=1e-4)
    ((y['depth_logits'] * r['wd'].to(DEV)).sum() + (y['mask_logits'] * r['wm'].to(DEV)).sum()).backward()
    # Camera gradients: relative L2 error of the 10-vector per sample.  The network is piecewise
    # linear in places (LeakyReLU): when a pre-activation of the reference run is within fp32
    # noise of 0 (|pre| ~ 2e-5 exists in the 'occlusion' fixture) the two runs may take different
    # slopes for that one activation, which moves the summed gradient by ~1e-2 relative.  That is
    # a property of the test point, not of the kernels (layerwise forward agreement is ~2e-6).
    got = torch.cat((cam.log_quaternion.grad, cam.translation.grad, cam.viewport.grad), dim=1).cpu()
    want = torch.cat((r['g_log_q'], r['g_t'], r['g_viewport']), dim=1)
    rel = ((got - want).norm(dim=1) / want.norm(dim=1)).max().item()
    assert rel < (3e-2 if variant == "occlusion" else 1e-2), rel


@pytest.mark.parametrize('tag', ['gru', 'pool_mean'])
def test_g10_encode(golden, tag):
    from latentfusion_amd.recon.models import Sculptor
    from latentfusion_amd.recon import fusion
    g = golden('g10_encode_' + tag)
    sc = Sculptor.from_checkpoint(g['sculptor']).to(DEV)
    fu = fusion.from_checkpoint(g['fuser']).to(DEV)
    o = g['obs_pre']
    cam = prod_camera(o['cam'])
    with torch.no_grad():
        z, _ = sc.encode(fu, cam, o['color'].unsqueeze(0).to(DEV), o['depth'].unsqueeze(0).to(DEV),
                         o['mask'].unsqueeze(0).to(DEV))
    close(z, g['z_obj'], atol=1e-4, rtol=1e-3)


def test_g4_fusers(golden):
    from latentfusion_amd.recon import fusion
    g = golden('g4_fusers')
    z = g['z'].to(DEV)
    cam = prod_camera(g['cam'])
    for pool in ('mean', 'max', 'abs_max', 'median'):
        out, _ = fusion.PoolFuser(pool)(z, None, None, cam)
        close(out, g['pool_' + pool], atol=1e-6)
    for kind in ('gru', 'lstm', 'blend'):
        f = fusion.from_checkpoint(g[kind]['ck']).to(DEV)
        with torch.no_grad():
            mids = [g['blend']['z_cam_mid'].to(DEV)] if kind == 'blend' else None
            out, _ = f(z, mids, None, cam)
        close(out, g[kind]['out'], atol=1e-4, rtol=1e-3)


def test_g11_released_like_structure(golden):
    """The generic kernel paths behind the released architecture's structure: U-Nets with D/U
    rescaling and skip concatenations, channel counts 12/20/24/48 (multi-chunk, non-multiple-of-16),
    camera blocks that change width, 32x32 output from a 16^3 volume -- encode, decode, camera grads."""
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g11_released_like')
    sc = Sculptor.from_checkpoint(g['sculptor']).to(DEV)
    fu = fusion.from_checkpoint(g['fuser']).to(DEV)
    ph = Photographer.from_checkpoint(g['photographer']).to(DEV)
    o = g['obs_pre']
    with torch.no_grad():
        z, _ = sc.encode(fu, prod_camera(o['cam']), o['color'].unsqueeze(0).to(DEV), o['depth'].unsqueeze(0).to(DEV),
                         o['mask'].unsqueeze(0).to(DEV))
    close(z, g['z_obj'], atol=2e-4, rtol=2e-3)
    cam = prod_camera(g['cam'])
    for p in (cam.log_quaternion, cam.translation, cam.viewport):
        p.requires_grad_(True)
    y, lat, _ = ph.decode(g['z_obj'].to(DEV), cam, return_latent=True, apply_mask=True)
    for k in ('depth_logits', 'mask_logits', 'depth', 'mask'):
        close(y[k], g['y'][k], atol=2e-4, rtol=2e-3)
    ((y['depth_logits'] * g['wd'].to(DEV)).sum() + (y['mask_logits'] * g['wm'].to(DEV)).sum()).backward()
    got = torch.cat((cam.log_quaternion.grad, cam.translation.grad, cam.viewport.grad), dim=1).cpu()
    want = torch.cat((g['g_log_q'], g['g_t'], g['g_viewport']), dim=1)
    rel = ((got - want).norm(dim=1) / want.norm(dim=1)).max().item()
    assert rel < 2e-2, rel


def test_g8_cross_entropy_step_on_hip(golden):
    """CrossEntropyPoseEstimator.evaluate_samples (flip augmentation, zoom, render without grad,
    loss) on the HIP path: per-sample losses and their ORDER equal the reference's."""
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g, t7 = golden('g8_ce_step'), golden('g7_adam_trace')
    model = LatentFusionModel(Sculptor.from_checkpoint(t7['sculptor']), fusion.from_checkpoint(t7['fuser']),
                              Photographer.from_checkpoint(t7['photographer']), t7['camera_dist'], DEV)
    est = estimation.CrossEntropyPoseEstimator(model=model, num_samples=24, num_elites=5, num_iters=3,
                                               num_gmm_components=2, learning_rate=0.9, sample_flipped=True,
                                               ranking_size=4, loss_weights=g['weights'])
    tg = t7['target']
    target = Observation(None, tg['depth'], tg['mask'].float(), prod_camera(tg['cam'], 'cpu')).to(DEV)
    cams, loss = est.evaluate_samples(t7['z_obj'].to(DEV), target, prod_camera(g['cams']))
    close(loss, g['loss'], atol=1e-4, rtol=1e-3)
    assert torch.argsort(loss).cpu().tolist() == g['order'].tolist()


def test_g9_ibr_on_hip(golden):
    """render_ibr_basic-style colour rendering: depths from the HIP renderer, reprojection + blend."""
    from latentfusion_amd import ibr
    from latentfusion_amd.recon.models import Photographer
    g = golden('g9_ibr')
    ph = Photographer.from_checkpoint(g['ck']).to(DEV)
    cam_in, cam_out = prod_camera(g['cam_in']), prod_camera(g['cam_out'])
    y, z = ibr.render_latent_ibr2(ph, g['z_obj'].to(DEV), cam_in, cam_out, g['image_in'].to(DEV), p=0.5,
                                  weight_type='cam_dist', apply_mask=True)
    close(y['depth'], g['depth'], atol=2e-4, rtol=2e-3)
    close(y['color'], g['color'], atol=2e-3, rtol=1e-2)
    with torch.no_grad():
        y_in, _, _ = ph.decode(g['z_obj'].to(DEV), cam_in, apply_mask=True)
        y_out, _, _ = ph.decode(g['z_obj'].to(DEV), cam_out, apply_mask=True)
        img_re, dep_re = ibr.reproject_views(g['image_in'][0].to(DEV), y_in['depth'][0], y_out['depth'][0], cam_in, cam_out)
    close(img_re, g['image_reproj'], atol=2e-3, rtol=1e-2)


def test_g23_ibr_generator_on_hip(golden):
    """Row a14 remainder on the device: LatentFusionModel.render_ibr with a loaded generator U-Net (depths from the HIP
    renderer, reprojection + learned flow through lf_grid_sample2d_fwd, the generator through the 2-D conv kernels),
    reproject_views_batch with its camera distances, render_latent_ibr, blend_logits / warp_blend_logits -- against the
    real reference's outputs (ibr.py:96-154,225-249; recon/inference.py:151-217)."""
    from latentfusion_amd import ibr
    from latentfusion_amd.modules.unet import UNet2d
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g23_ibr_generator')
    e = golden('g10_encode_pool_mean')                      # any sculptor / fuser of the same size: render_ibr never encodes
    ph = Photographer.from_checkpoint(g['photographer'])
    model = LatentFusionModel(Sculptor.from_checkpoint(e['sculptor']), fusion.from_checkpoint(e['fuser']), ph,
                              g['camera_dist'], DEV, generator=UNet2d.from_checkpoint(g['generator']))
    cam_in, cam_out = prod_camera(g['cam_in']), prod_camera(g['cam_out'])
    obs = Observation(g['color_in'], g['depth_in'], g['mask_in'], prod_camera(g['cam_in'], 'cpu'), is_zoomed=True,
                      is_prepared=True, is_normalized=True).to(DEV)
    z_obj = g['z_obj'].to(DEV)
    with torch.no_grad():
        y, z_out = model.render_ibr(z_obj, obs, cam_out)
        _, _, img_re, dep_re, _, _, dist_r, dist_t = model._render_reprojections(z_obj, obs.color, cam_in, cam_out)
        col, d_out, m_out, reproj = ibr.render_latent_ibr(model.photographer, z_obj, cam_in, cam_out, obs.color.unsqueeze(0), p=0.5)
        wb = ibr.warp_blend_logits(g['logits'].to(DEV), g['image_reproj'].to(DEV), 5)
        bl = ibr.blend_logits(g['logits'][:, :3].to(DEV), g['image_reproj'].to(DEV))
    for k in ('depth', 'mask', 'depth_logits', 'mask_logits'):
        close(y[k], g['y'][k], atol=2e-4, rtol=2e-3)
    close(img_re, g['image_reproj'], atol=2e-3, rtol=1e-2)
    close(dep_re, g['depth_reproj'], atol=2e-3, rtol=1e-2)
    close(dist_r, g['cam_dist_r'], atol=1e-5, rtol=1e-4)
    close(dist_t, g['cam_dist_t'], atol=1e-5, rtol=1e-4)
    close(y['color'], g['y']['color'], atol=3e-3, rtol=1e-2)
    close(z_out, g['z_out'], atol=3e-4, rtol=3e-3)
    close(col, g['latent_ibr']['color'], atol=2e-3, rtol=1e-2)
    close(d_out, g['latent_ibr']['depth'], atol=2e-4, rtol=2e-3)
    close(reproj, g['latent_ibr']['reproj'], atol=2e-3, rtol=1e-2)
    for a, b in zip(wb, g['warp_blend']):                   # same inputs as the reference: only the sampler differs
        close(a, b, atol=2e-5, rtol=1e-4)
    for a, b in zip(bl, g['blend']):
        close(a, b, atol=1e-5, rtol=1e-4)


def test_g24_tile_projection_on_hip(golden):
    """Row a4: TileProjection2d3d (modules/geometry.py:693-708) -- the module alone and inside a
    Sculptor(projection_type='tile') encode, with the gradient w.r.t. the input images."""
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.models import Sculptor
    g = golden('g24_tile_projection')
    sc = Sculptor.from_checkpoint(g['sculptor']).to(DEV)
    fu = fusion.from_checkpoint(g['fuser']).to(DEV)
    assert sc.projection_type == 'tile' and type(sc.projection_block).__name__ == 'TileProjection2d3d'
    with torch.no_grad():
        lifted = sc.projection_block(g['x2d'].to(DEV))
    close(lifted, g['lifted'], atol=2e-5, rtol=1e-4)
    color = g['color'].to(DEV).requires_grad_(True)
    z, _ = sc.encode(fu, prod_camera(g['cam']), color, None, g['mask'].to(DEV))
    close(z, g['z_obj'], atol=1e-4, rtol=1e-3)
    (z * g['wz'].to(DEV)).sum().backward()
    # The network is piecewise linear: the reference's own fp32 gradient has a pre-activation within rounding of 0 on the
    # other LeakyReLU slope than an exact evaluation (6.8 % of its elements differ from the oracle in fp64 by up to 0.2;
    # tests/test_oracle_golden.py pins oracle == reference bit for bit in fp32).  So the gradient is judged against the fp64
    # evaluation: the HIP path must be at least as close to it as the fp32 reference is, and within 1e-2 of the reference.
    from oracle_util import cast, in_fp64

    def exact():
        c64 = cast(g['color'][0], torch.float64).requires_grad_(True)
        sck = {'args': g['sculptor']['args'], 'state_dict': cast(g['sculptor']['state_dict'], torch.float64)}
        z64 = nets.encode(sck, g['fuser'], O.cam_from_dict(cast(g['cam'], torch.float64)), c64, None, cast(g['mask'][0], torch.float64))
        (z64 * cast(g['wz'], torch.float64)).sum().backward()
        return c64.grad
    g64 = in_fp64(exact)

    def rel(a):
        return ((a.detach().cpu().double() - g64).norm() / g64.norm()).item()
    err_hip, err_ref = rel(color.grad[0]), rel(g['grad_color'][0])
    assert err_hip <= max(1e-4, 1.2 * err_ref), (err_hip, err_ref)
    assert ((color.grad.cpu() - g['grad_color']).norm() / g['grad_color'].norm()).item() < 1e-2


def test_g12_latent_code_on_hip(golden):
    """compute_latent_code (encode the target crop under each candidate camera + decode) and the
    latent term of the pose loss -- the path used by adam_latent / cross_entropy_latent."""
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose.loss import default_pose_loss
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g12_latent_code')
    model = LatentFusionModel(Sculptor.from_checkpoint(g['sculptor']), fusion.from_checkpoint(g['fuser']),
                              Photographer.from_checkpoint(g['photographer']), g['camera_dist'], DEV)
    tg = g['target']
    target = Observation(tg['color_u8'].float() / 255.0, tg['depth'], tg['mask'].float(), prod_camera(tg['cam'], 'cpu')).to(DEV)
    cams = prod_camera(g['cams'])
    with torch.no_grad():
        zt = model.compute_latent_code(target, cams)
        pred, zp = model.render_latent_object(g['z_obj'].to(DEV), cams, return_latent=True)
        ld = default_pose_loss(target, cams.denormalize_depth(pred['depth'].squeeze(0)), pred['mask_logits'].squeeze(0),
                               cams, z_pred_latent=zp, z_target_latent=zt)
    close(zt, g['z_target_latent'], atol=3e-4, rtol=3e-3)
    close(zp, g['z_pred_latent'], atol=3e-4, rtol=3e-3)
    close(ld['latent'], g['latent_loss'], atol=1e-4, rtol=1e-3)


def test_g19_metropolis_on_hip(golden):
    """MetropolisPoseEstimator on the HIP path, replaying the random draws the reference consumed (two randn_like of
    pu.perturb_camera + the rand_like of the acceptance test per step): identical accept / reject decisions at
    every step, per-sample errors within the north-star tolerance, identical final ranking (steps and cameras)."""
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g, m = golden('g19_metropolis'), golden('g12_latent_code')
    model = LatentFusionModel(Sculptor.from_checkpoint(m['sculptor']), fusion.from_checkpoint(m['fuser']),
                              Photographer.from_checkpoint(m['photographer']), m['camera_dist'], DEV)
    tg = m['target']
    target = Observation(tg['color_u8'].float() / 255.0, tg['depth'], tg['mask'].float(), prod_camera(tg['cam'], 'cpu'))
    est = estimation.MetropolisPoseEstimator(model=model, num_samples=g['num_samples'], num_iters=g['num_iters'],
                                             ranking_size=g['ranking_size'], loss_weights=g['weights'],
                                             translation_std=g['translation_std'], quaternion_std=g['quaternion_std'],
                                             return_camera_history=True)
    est.replay_draws = [d for s in g['steps'] for d in (s['noise_t'], s['noise_q'], s['thresholds'])]
    # step by step (the loop body of _estimate), so that every intermediate state is compared
    camera = prod_camera(g['init'])
    error = torch.full((g['num_samples'],), 100.0, device=DEV)
    z_obj = m['z_obj'].to(DEV)
    tdev = target.to(DEV)
    prev_t = g['init']['t']
    for s in g['steps']:
        camera, error, n_acc = est._refine_pose(z_obj, camera.clone(), error.clone(), tdev, s['temperature'])
        assert n_acc == s['num_accepted']
        assert torch.equal((camera.translation.cpu() != prev_t).any(dim=1), (s['t'] != prev_t).any(dim=1))
        close(error, s['error'], atol=1e-4, rtol=1e-3)
        close(camera.translation, s['t'], atol=1e-6)
        close(camera.log_quaternion, s['log_q'], atol=1e-6)
        prev_t = s['t']
    # and the whole estimator from the same sample cameras
    est.replay_draws = [d for s in g['steps'] for d in (s['noise_t'], s['noise_q'], s['thresholds'])]
    best, hist = est.estimate(z_obj, target, cameras=prod_camera(g['init'], 'cpu'))
    assert est.accept_history == [s['num_accepted'] for s in g['steps']]
    close(best.translation, g['ranking_t'], atol=1e-6)
    close(best.log_quaternion, g['ranking_log_q'], atol=1e-6)


@pytest.mark.parametrize('variant', ['factor', 'sum'])
def test_photographer_parameter_gradients(golden, variant):
    """Training-step side: d(loss)/d(every Photographer weight and bias) through the HIP
    weight-gradient kernels against autograd of the CPU oracle on the same checkpoint and inputs."""
    import copy
    from latentfusion_amd.recon.models import Photographer
    r = golden('g5_decode')[variant]
    ck = copy.deepcopy(r['ck'])
    osd = {k: v.clone().requires_grad_(True) for k, v in ck['state_dict'].items()}
    ocam = O.Cam(r['cam']['K'], r['cam']['log_q'], r['cam']['t'], viewport=r['cam']['viewport'], z_span=r['cam']['z_span'],
                 width=r['cam']['width'], height=r['cam']['height'])
    y, _, _ = nets.decode({'args': ck['args'], 'state_dict': osd}, r['z_obj'], ocam, apply_mask=True)
    ((y['depth_logits'] * r['wd']).sum() + (y['mask_logits'] * r['wm']).sum()).backward()
    ph = Photographer.from_checkpoint(r['ck']).to(DEV)
    ph.requires_grad_(True)
    cam = prod_camera(r['cam'])
    yp, _, _ = ph.decode(r['z_obj'].to(DEV), cam, return_latent=True, apply_mask=True)
    ((yp['depth_logits'] * r['wd'].to(DEV)).sum() + (yp['mask_logits'] * r['wm'].to(DEV)).sum()).backward()
    checked = 0
    for name, p in ph.named_parameters():
        want = osd[name].grad
        assert want is not None and p.grad is not None, name
        # relative L2 per tensor (same LeakyReLU slope-flip sensitivity as the camera gradients above)
        rel = ((p.grad.cpu() - want).norm() / want.norm().clamp(min=1e-6)).item()
        assert rel < 1e-2, (name, rel)
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize('tag', ['gru', 'pool_mean'])
def test_generator_parameter_gradients_end_to_end(golden, tag):
    """One generator step's backward: Sculptor.encode + fuser + Photographer.decode,
    L1 on depth + BCE on the mask logits, gradients of EVERY sculptor / fuser / photographer parameter
    on the HIP path (weight-gradient kernels, volume splat, differentiable lift) against autograd of the
    CPU oracle with the same checkpoints and inputs."""
    import copy
    import torch.nn.functional as F
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g10_encode_' + tag)
    pck = golden('g7_adam_trace')['photographer']              # SYN(16,8) photographer, same volume size / channels
    o = g['obs_pre']
    gen = torch.Generator().manual_seed(3)
    tgt_depth = torch.rand(1, 2, 1, 16, 16, generator=gen) * 2 - 1
    tgt_mask = (torch.rand(1, 2, 1, 16, 16, generator=gen) > 0.5).float()

    def objective(y):
        return F.l1_loss(y['depth'], tgt_depth.to(y['depth'].device)) + \
            F.binary_cross_entropy_with_logits(y['mask_logits'], tgt_mask.to(y['depth'].device))

    # oracle
    cks = {k: {**ck, 'state_dict': {n: v.clone().requires_grad_(True) for n, v in ck.get('state_dict', {}).items()}}
           for k, ck in (('s', g['sculptor']), ('f', g['fuser']), ('p', pck))}
    oc = o['cam']
    ocam = O.Cam(oc['K'], oc['log_q'], oc['t'], viewport=oc['viewport'], z_span=oc['z_span'], width=oc['width'],
                 height=oc['height'])
    z = nets.encode(cks['s'], cks['f'], ocam, o['color'], o['depth'], o['mask'])
    rcam = ocam[:2] if hasattr(ocam, '__getitem__') else ocam
    y, _, _ = nets.decode(cks['p'], z, rcam, apply_mask=False)
    objective(y).backward()

    # HIP
    sc = Sculptor.from_checkpoint(g['sculptor']).to(DEV)
    fu = fusion.from_checkpoint(g['fuser']).to(DEV)
    ph = Photographer.from_checkpoint(pck).to(DEV)
    for m in (sc, fu, ph):
        m.requires_grad_(True)
    cam = prod_camera(oc)
    zp, _ = sc.encode(fu, cam, o['color'].unsqueeze(0).to(DEV), o['depth'].unsqueeze(0).to(DEV), o['mask'].unsqueeze(0).to(DEV))
    close(zp, z, atol=1e-4, rtol=1e-3)
    yp, _, _ = ph.decode(zp, cam[:2], return_latent=True, apply_mask=False)
    objective(yp).backward()
    checked = 0
    for key, mod in (('s', sc), ('f', fu), ('p', ph)):
        for name, p in mod.named_parameters():
            want = cks[key]['state_dict'][name].grad
            if want is None:                                     # parameter not on this path (unused head)
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, (key, name)
                continue
            assert p.grad is not None, (key, name)
            rel = ((p.grad.cpu() - want).norm() / want.norm().clamp(min=1e-8)).item()
            assert rel < 2e-2, (key, name, rel)
            checked += 1
    assert checked >= 25


def test_generator_training_step(golden):
    """recon/training.GeneratorStep (loss composition of train_reconstruct.py:491-516, flat Adam with the
    reference's betas): loss terms equal the oracle's on the same batch, the first update moves every
    parameter by lr * g / (|g| + eps) with the oracle's gradient sign, and repeated steps on one batch
    reduce the loss."""
    import torch.nn.functional as F
    from latentfusion_amd import losses as L
    from latentfusion_amd.recon import fusion, training
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g10_encode_gru')
    pck = golden('g7_adam_trace')['photographer']
    o, oc = g['obs_pre'], g['obs_pre']['cam']
    gen = torch.Generator().manual_seed(5)
    tgt_depth = torch.rand(1, 4, 1, 16, 16, generator=gen) * 2 - 1
    tgt_mask = (torch.rand(1, 4, 1, 16, 16, generator=gen) > 0.5).float()
    kw = dict(g_depth_recon_loss_type='hard_smooth_l1', g_depth_recon_loss_k=100, g_depth_recon_loss_weight=25.0,
              g_mask_recon_loss_weight=25.0, g_mask_beta_loss_weight=0.5, g_mask_beta_loss_param=0.01, generator_lr=1e-3)

    # oracle: same objective through the CPU restatement
    cks = {k: {**ck, 'state_dict': {n: v.clone().requires_grad_(True) for n, v in ck.get('state_dict', {}).items()}}
           for k, ck in (('s', g['sculptor']), ('f', g['fuser']), ('p', pck))}
    ocam = O.Cam(oc['K'], oc['log_q'], oc['t'], viewport=oc['viewport'], z_span=oc['z_span'], width=oc['width'],
                 height=oc['height'])
    z = nets.encode(cks['s'], cks['f'], ocam, o['color'], o['depth'], o['mask'])
    y, _, _ = nets.decode(cks['p'], z, ocam, apply_mask=False)
    want = {'depth_recon': L.reduce_loss(L.get_recon_criterion('hard_smooth_l1', 100)(y['depth'], tgt_depth)),
            'mask_recon': L.reduce_loss(L.get_recon_criterion('binary_cross_entropy')(y['mask_logits'], tgt_mask)),
            'mask_beta': L.beta_prior_loss(y['mask'], 0.01, 0.01)}
    want['total'] = 25.0 * want['depth_recon'] + 25.0 * want['mask_recon'] + 0.5 * want['mask_beta']
    want['total'].backward()

    sc = Sculptor.from_checkpoint(g['sculptor']).to(DEV)
    fu = fusion.from_checkpoint(g['fuser']).to(DEV)
    ph = Photographer.from_checkpoint(pck).to(DEV)
    step = training.GeneratorStep(sc, fu, ph, **kw)
    before = [q.detach().clone() for q in step.flat.params]
    cam = prod_camera(oc)
    batch = {'in': {'camera': cam, 'image': o['color'].unsqueeze(0).to(DEV), 'mask': o['mask'].unsqueeze(0).to(DEV),
                    'depth': o['depth'].unsqueeze(0).to(DEV)},
             'out_gt': {'camera': cam, 'depth': tgt_depth.to(DEV), 'mask': tgt_mask.to(DEV)}}
    got = step.run_iteration(batch)
    for k in ('depth_recon', 'mask_recon', 'mask_beta', 'total'):
        close(got[k], want[k], atol=1e-5, rtol=1e-4)
    # first Adam step with betas (0, 0.99): delta = -lr * g / (|g| + eps)
    delta = torch.cat([(q.detach() - b).reshape(-1) for q, b in zip(step.flat.params, before)]).cpu()
    ograd = torch.cat([cks[key]['state_dict'][n].grad.reshape(-1)
                       for key, mod in (('s', sc), ('p', ph), ('f', fu)) for n, _ in mod.named_parameters()])
    assert delta.shape == ograd.shape
    big = ograd.abs() > 1e-5 * ograd.abs().max()
    agree = (torch.sign(delta[big]) == -torch.sign(ograd[big])).float().mean().item()
    assert agree > 0.995, agree
    assert float(delta.abs().max()) <= 1e-3 * 1.0001
    # the updated WEIGHTS (not only the biases, which kernels read in place) must be what the next forward
    # uses: a model rebuilt from the updated state dicts gives the same loss
    after = step.run_iteration(batch, train=False)
    sc2 = Sculptor.from_checkpoint({'args': g['sculptor']['args'], 'state_dict': {k: v.detach().clone() for k, v in sc.state_dict().items()}}).to(DEV)
    fu2 = fusion.from_checkpoint({**g['fuser'], 'state_dict': {k: v.detach().clone() for k, v in fu.state_dict().items()}}).to(DEV)
    ph2 = Photographer.from_checkpoint({'args': pck['args'], 'state_dict': {k: v.detach().clone() for k, v in ph.state_dict().items()}}).to(DEV)
    with torch.no_grad():
        z2, _ = sc2.encode(fu2, cam, batch['in']['image'], None, batch['in']['mask'])
        y2, _, _ = ph2.decode(z2, cam, return_latent=True, apply_mask=False)
    d2 = L.reduce_loss(L.get_recon_criterion('hard_smooth_l1', 100)(y2['depth'], tgt_depth.to(DEV)))
    close(after['depth_recon'], d2, atol=1e-6, rtol=1e-5)
    assert abs(float(after['total']) - float(got['total'])) > 1e-4
    first = float(got['total'])
    for _ in range(4):
        last = step.run_iteration(batch)
    assert float(last['total']) < first


def test_gru_fuser_inference_path_equals_general_path(golden):
    """GRUFuser's inference path (merged update+reset convolution, in-place state record, lf_gru_stage_a/b)
    against the module's general (autograd) path on the same weights and views."""
    from latentfusion_amd.recon import fusion
    g = golden('g10_encode_gru')
    fu = fusion.from_checkpoint(g['fuser']).to(DEV)
    gen = torch.Generator().manual_seed(21)
    z = torch.randn(1, 5, 8, 12, 10, 14, generator=gen).to(DEV)          # (B=1, V=5, C=8, D, H, W), non-cubic
    with torch.no_grad():
        fast, _ = fu(z, None, None, None)
    with torch.enable_grad():
        slow, _ = fu(z.clone().requires_grad_(True), None, None, None)
    assert fast.shape == slow.shape == (1, 1, 8, 12, 10, 14)
    close(fast, slow, atol=2e-6, rtol=1e-5)


def test_gru_fuser_winograd_inference_path():
    """16-channel volumes: the gate convolutions run as sums of 16 -> 16 Winograd convolutions (addend form of
    lf_conv3d_c16_wino; coords part once per object).  Same recurrence as the module's general path
    (modules/gru.py:37-43) up to fp32 summation order; odd sizes exercise partial tiles on every axis."""
    from latentfusion_amd.recon import fusion
    torch.manual_seed(5)
    fu = fusion.GRUFuser(16).to(DEV)
    with torch.no_grad():
        for gate in (fu.gru.update_gate, fu.gru.reset_gate, fu.gru.out_gate):
            gate.bias.normal_(0.0, 0.3)
    gen = torch.Generator().manual_seed(22)
    z = torch.randn(1, 4, 16, 9, 12, 21, generator=gen).to(DEV)
    with torch.no_grad():
        fast, _ = fu(z, None, None, None)
    with torch.enable_grad():
        slow, _ = fu(z.clone().requires_grad_(True), None, None, None)
    assert fast.shape == slow.shape == (1, 1, 16, 9, 12, 21)
    close(fast, slow, atol=2e-5, rtol=1e-4)


def test_gru_fuser_split_gates_match_concatenated_gates_under_autograd():
    """Training path of the 16-channel GRU fuser: gates as sums of 16 -> 16 convolutions over (view, coords, state)
    (ops.conv3x3_sum16: Winograd forward / data gradients, LDS-staged weight gradients, no 35-channel
    concatenation) against the concatenated 35-channel convolutions of the reference formulation
    (modules/gru.py:37-43) -- output, gradients of every gate weight and bias, gradient of the per-view volumes."""
    from latentfusion_amd.recon import fusion
    torch.manual_seed(6)
    fu = fusion.GRUFuser(16).to(DEV)
    for p in fu.parameters():
        p.requires_grad_(True)
    with torch.no_grad():
        for gate in (fu.gru.update_gate, fu.gru.reset_gate, fu.gru.out_gate):
            gate.bias.normal_(0.0, 0.3)
    gen = torch.Generator().manual_seed(23)
    z0 = torch.randn(1, 3, 16, 10, 24, 33, generator=gen).to(DEV)
    gout = torch.randn(1, 1, 16, 10, 24, 33, generator=gen).to(DEV)
    res = {}
    for split in (False, True, 'per-view coords'):
        fu.split_gates = bool(split)
        fu.hoist_coords = split is True              # True: the coordinate share of the gates once per forward (default)
        fu.zero_grad()
        z = z0.clone().requires_grad_(True)
        out, _ = fu(z, None, None, None)
        (out * gout).sum().backward()
        res[split] = (out.detach(), z.grad.clone(), {k: p.grad.clone() for k, p in fu.named_parameters()})
    for form in (True, 'per-view coords'):
        close(res[form][0], res[False][0], atol=3e-5, rtol=1e-4)
        scale = res[False][1].abs().max().item()
        close(res[form][1], res[False][1], atol=2e-5 * scale, rtol=1e-3)
        for k, g in res[False][2].items():
            close(res[form][2][k], g, atol=2e-5 * max(g.abs().max().item(), 1e-3), rtol=1e-3)


def test_skip_connections_quirk(golden):
    """Quirk Q20: the reference's Photographer(skip_connections=True) concatenates a skip tensor in front of camera
    block 0, which was allocated without skip channels (create_blocks(skip_connect_start=True)), so its forward raises
    a channel-mismatch RuntimeError for every configuration (golden g22).  The HIP path builds the same modules,
    performs the same O2C resamples / concatenations and stops at the same point."""
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g22_photographer_skip')
    case = g['cases']['b']
    torch.manual_seed(0)
    sc = Sculptor(in_size=g['in_size'], image_config=g['image_config'], camera_config=case['camera_config'],
                  object_config=case['object_config'], projection_type='factor', scale_mode='nearest').to(DEV)
    ph = Photographer(in_size=g['in_size'], image_config=g['image_config'], camera_config=case['camera_config'],
                      object_config=case['object_config'], projection_type='factor', skip_connections=True,
                      scale_mode='nearest').to(DEV)
    cam = prod_camera(golden('g2_resample')['cam'])[:2]
    with torch.no_grad():
        z, zc, zo = sc(torch.randn(2, 4, g['in_size'], g['in_size'], device=DEV), cam)
        with pytest.raises(RuntimeError, match='channels'):
            ph(z, cam, z_cam_mid=zc, z_obj_mid=zo)


@pytest.mark.parametrize('autocast', [False, True])
def test_gru_fuser_fused_recurrence_matches_per_gate_functions(autocast):
    """ops.gru_fuse (round 5: the whole ConvGRU recurrence as one autograd node, explicit backward, addend-chained data
    gradients, gate-gradient sums kept by the stage kernels, bf16 storage of the per-step tensors under autocast) against
    the per-gate autograd functions it replaces (ops.conv3x3_sum16 / gru_gates / gru_blend): output, gradient of the
    per-view volumes, gradients of every gate weight and bias.  fp32: the two forms run the same kernels on the same
    numbers in another order (tolerance = accumulation order); autocast: the fused form additionally rounds the gate
    pre-activations / gate gradients to bf16 where it stores them (tolerance = bf16 storage of O(1) tensors)."""
    from latentfusion_amd import ops
    from latentfusion_amd.recon import fusion
    torch.manual_seed(6)
    fu = fusion.GRUFuser(16).to(DEV)
    for p in fu.parameters():
        p.requires_grad_(True)
    with torch.no_grad():
        for gate in (fu.gru.update_gate, fu.gru.reset_gate, fu.gru.out_gate):
            gate.bias.normal_(0.0, 0.3)
    gen = torch.Generator().manual_seed(29)
    z0 = torch.randn(1, 4, 16, 10, 24, 33, generator=gen).to(DEV)
    gout = torch.randn(1, 1, 16, 10, 24, 33, generator=gen).to(DEV)
    res = {}
    for fused in (False, True, True):
        fu.fused_recurrence = fused
        fu.zero_grad()
        z = z0.clone().requires_grad_(True)
        with ops.autocast(autocast):
            out, _ = fu(z, None, None, None)
        (out * gout).sum().backward()
        res.setdefault(fused, []).append((out.detach(), z.grad.clone(), {k: p.grad.clone() for k, p in fu.named_parameters()}))
    # run-to-run identical
    a, b = res[True]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
    ref = res[False][0]
    tol = 3e-2 if autocast else 3e-5
    close(a[0], ref[0], atol=tol, rtol=1e-2 if autocast else 1e-4)
    if autocast:
        # round 6: the recurrence on the multi-output ring kernels (ops_train.GRU_RING) against round 5's one-output kernels +
        # stage kernels: the same bf16 storage points, so far closer to each other than either is to the per-gate functions
        from latentfusion_amd import ops_train
        assert ops_train.GRU_RING
        ops_train.GRU_RING = False
        try:
            fu.zero_grad()
            z = z0.clone().requires_grad_(True)
            with ops.autocast(True):
                out, _ = fu(z, None, None, None)
            (out * gout).sum().backward()
            old = (out.detach(), z.grad.clone(), {k: p.grad.clone() for k, p in fu.named_parameters()})
        finally:
            ops_train.GRU_RING = True
        # (a stored pre-activation that lands on the other side of a bf16 rounding boundary moves an output element by up to
        # ulp * |c - h| u (1 - u) ~ 6e-3; a handful of elements do)
        close(a[0], old[0], atol=1.5e-2, rtol=1e-2)
        assert float((a[0] - old[0]).norm() / old[0].norm()) < 1e-3
        cosf = lambda x, y: torch.nn.functional.cosine_similarity(x.reshape(1, -1).double(), y.reshape(1, -1).double()).item()  # noqa: E731
        assert cosf(a[1], old[1]) > 0.99995, cosf(a[1], old[1])
        for k, g in old[2].items():
            assert cosf(a[2][k], g) > 0.9999, (k, cosf(a[2][k], g))
    scale = ref[1].abs().max().item()
    if autocast:
        cos = torch.nn.functional.cosine_similarity(a[1].reshape(1, -1).double(), ref[1].reshape(1, -1).double()).item()
        assert cos > 0.999, cos
        for k, g in ref[2].items():
            cos = torch.nn.functional.cosine_similarity(a[2][k].reshape(1, -1).double(), g.reshape(1, -1).double()).item()
            assert cos > 0.995, (k, cos)
    else:
        close(a[1], ref[1], atol=2e-5 * scale, rtol=1e-3)
        for k, g in ref[2].items():
            close(a[2][k], g, atol=2e-5 * max(g.abs().max().item(), 1e-3), rtol=1e-3)


def test_ring_conv_bf16_storage_variants():
    """lf_conv3d_c16_ring_bf16_io: every storage combination of input / output / addend gives the result of the fp32-storage
    launch on the same (bf16-representable) numbers, rounded once where the output is stored as bf16."""
    from latentfusion_amd import ops
    gen = torch.Generator().manual_seed(3)
    N, D, H, W = 2, 6, 11, 19
    x = ops.cl(ops.round_bf16(torch.randn(N, 16, D, H, W, generator=gen).to(DEV)))
    add = ops.cl(ops.round_bf16(torch.randn(N, 16, D, H, W, generator=gen).to(DEV)))
    w = torch.randn(16, 16, 3, 3, 3, generator=gen).to(DEV)
    bias = torch.randn(16, generator=gen).to(DEV)
    wp = ops.pack_conv3d_c16_ring_bf16(w)
    he = 0.21
    cl3 = torch.channels_last_3d
    for flags, addend, b in ((0, None, None), (3, None, bias), (0, add, None)):
        ref, nref = ops.conv3d_c16_ring_bf16_io(x, wp, b, he, flags, 0, addend=addend)
        for in16 in (False, True):
            for out16 in (False, True):
                for add16 in ((False, True) if addend is not None else (False,)):
                    xi = x.to(torch.bfloat16).contiguous(memory_format=cl3) if in16 else x
                    ai = (addend.to(torch.bfloat16).contiguous(memory_format=cl3) if add16 else addend) if addend is not None else None
                    y, nrm = ops.conv3d_c16_ring_bf16_io(xi, wp, b, he, flags, 0, addend=ai, out_bf16=out16)
                    assert y.dtype == (torch.bfloat16 if out16 else torch.float32)
                    want = ref.to(torch.bfloat16) if out16 else ref
                    assert torch.equal(y, want), (flags, in16, out16, add16, (y.float() - want.float()).abs().max().item())
                    if nref is not None:
                        assert torch.equal(nrm, nref)


def test_wgrad_bf16_storage_variants():
    """lf_conv_bwd_weight_bf16_io: bf16-stored operands give bit-identical weight gradients to fp32-stored ones holding the
    same (bf16-representable) values."""
    from latentfusion_amd import _lib, ops
    gen = torch.Generator().manual_seed(4)
    N, D, H, W = 1, 16, 24, 32
    x = ops.cl(ops.round_bf16(torch.randn(N, 16, D, H, W, generator=gen).to(DEV)))
    gp = ops.cl(ops.round_bf16((torch.randn(N, 16, D, H, W, generator=gen) * 1e-2).to(DEV)))
    L = _lib.lib()
    nb = L.lf_conv_bwd_weight_scratch_bytes(3, N, D, H, W, 16, 16)
    scr = torch.empty(nb // 4 + 1, device=DEV)
    outs = {}
    for io in range(4):
        xi = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d) if io & 1 else x
        gi = gp.to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d) if io & 2 else gp
        gw = torch.empty(27, 16, 16, device=DEV)
        _lib.check(L.lf_conv_bwd_weight_bf16_io(xi.data_ptr(), gi.data_ptr(), gw.data_ptr(), scr.data_ptr(), scr.numel() * 4, 3, N, D, H, W,
                                                16, 16, 0.3, io, torch.cuda.current_stream().cuda_stream), 'wgrad io')
        outs[io] = gw
    for io in (1, 2, 3):
        assert torch.equal(outs[io], outs[0]), io


def test_heads_commute_with_the_final_upsampling():
    """Photographer.decode_features: a 2-D decoder that ends in an up-sampling (the released architecture's last 'U') hands that
    resize to the LOGITS -- the pointwise output blocks (reference recon/models.py:316-327, blocks.py:108-119) run before it.  A
    resize is linear over space with weights that sum to one, the heads are affine over channels: the two orders agree to fp32
    rounding (bilinear, the mode of the reference's 2-D U-Nets), logits and gradients."""
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.recon.models import Photographer
    from latentfusion_amd import synth
    torch.manual_seed(5)
    S, N = 16, 3
    td = synth.make_observation_data(N, seed=3)
    cam = Camera(td['intrinsic'], td['extrinsic']).zoom(None, S, 2.0).to(DEV)
    ph = Photographer(in_size=S, camera_config=[16, 16], object_config=[], image_config=[[16, 'D', 32], [32, 'U', 24, 'U', 12]],
                      predict_color=False, predict_depth=True, predict_mask=True, scale_mode='nearest', cube_size=1.0).to(DEV)
    with torch.no_grad():
        for name, p in ph.named_parameters():
            if name.endswith('bias'):
                p.normal_(0.0, 0.1)
    assert ph.image_decoder.up_blocks[-1].interpolate.mode == 'bilinear'
    z = torch.randn(1, 1, 16, S, S, S, generator=torch.Generator().manual_seed(8)).to(DEV).requires_grad_(True)
    z2d = torch.randn(N, 16, S, S, generator=torch.Generator().manual_seed(9)).to(DEV)
    feats, rescale = ph.decode_features(z2d)
    assert rescale is ph.image_decoder.up_blocks[-1].interpolate
    assert feats.shape[-1] == S and feats.shape[1] == 12 and ph.image_decoder(z2d).shape[-1] == 2 * S
    y, _, _ = ph.decode(z, cam, interpret_logits=False)
    assert y.shape[-2:] == (2 * S, 2 * S)
    wts = torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)
    gz, = torch.autograd.grad((y * wts).sum(), z)
    # the reference order: decoder with its resize, then the output blocks
    z_ = z.detach().clone().requires_grad_(True)
    orig = Photographer.decode_features
    Photographer.decode_features = lambda self, zz: (self.image_decoder(zz), None)
    try:
        y_ref, _, _ = ph.decode(z_, cam, interpret_logits=False)
    finally:
        Photographer.decode_features = orig
    close(y, y_ref, atol=2e-6 * y_ref.abs().max().item(), rtol=1e-5)
    gz_ref, = torch.autograd.grad((y_ref * wts).sum(), z_)
    close(gz, gz_ref, atol=1e-5 * gz_ref.abs().max().item(), rtol=1e-4)
