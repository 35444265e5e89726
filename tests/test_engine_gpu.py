"""GPU parity of the fused pieces of the pose loop: lf_camera_coefs (+Jacobian), the fused pose
loss, the batched Adam kernel, the RenderLoopEngine, and -- end to end -- the gradient pose loop
on the HIP path against the reference's golden iteration trace (G7: per-iteration losses,
argmin indices, camera trajectory)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def close(a, b, atol=1e-5, rtol=1e-4):
    torch.testing.assert_close(a.detach().cpu().contiguous(), b.detach().cpu().contiguous(), atol=atol, rtol=rtol)


def prod_camera(d, device=DEV):
    from latentfusion_amd.modules.geometry import Camera
    return Camera(d['K'].to(device), None, d['z_span'], d['viewport'].to(device), width=d['width'],
                  height=d['height'], log_quaternion=d['log_q'].to(device), translation=d['t'].to(device))


def _target(g, device=DEV):
    from latentfusion_amd.observation import Observation
    tg = g['target']
    return Observation(None, tg['depth'], tg['mask'].float(), prod_camera(tg['cam'], 'cpu')).to(device)


def test_camera_coefs_and_jacobian(golden):
    """lf_camera_coefs == the host (torch fp64) coefficient algebra, and its dual-number Jacobian ==
    autograd of that algebra."""
    from latentfusion_amd.engine import camera_coefs
    from latentfusion_amd.modules.geometry import o2c_coefficients
    cam = prod_camera(golden('g5_decode')['factor']['cam'])
    for p in (cam.log_quaternion, cam.translation, cam.viewport):
        p.requires_grad_(True)
    got = camera_coefs(cam, 1.0, 16, 16)
    want = o2c_coefficients(cam, 1.0)
    close(got[:, :18], want, atol=1e-6, rtol=1e-5)
    vw = cam.viewport[:, 2] - cam.viewport[:, 0]
    vh = cam.viewport[:, 3] - cam.viewport[:, 1]
    extra = torch.stack((16 / vw, -cam.viewport[:, 0] * 16 / vw - 0.5, 16 / vh, -cam.viewport[:, 1] * 16 / vh - 0.5,
                         torch.full_like(vw, 0.51), cam.translation[:, 2]), dim=1)
    close(got[:, 18:], extra, atol=1e-5, rtol=1e-5)
    g = torch.Generator().manual_seed(0)
    wgt = torch.randn(got.shape, generator=g).to(DEV)
    (got * wgt).sum().backward()
    g_got = [p.grad.clone() for p in (cam.log_quaternion, cam.translation, cam.viewport)]
    for p in (cam.log_quaternion, cam.translation, cam.viewport):
        p.grad = None
    (torch.cat((want, extra), dim=1) * wgt).sum().backward()
    for a, b in zip(g_got, (cam.log_quaternion.grad, cam.translation.grad, cam.viewport.grad)):
        close(a, b, atol=1e-4 * b.abs().max().item(), rtol=1e-4)


@pytest.mark.parametrize('crop', [16, 24])
def test_fused_pose_loss_matches_module_loss(golden, crop):
    """Fused loss (head logits -> losses, grads) == interpret_logits + denormalize + uncrop +
    default_pose_loss of the module path (itself pinned to the reference by G6 on the CPU)."""
    from latentfusion_amd.engine import camera_coefs, pose_loss
    # (the yardstick is the reference's EXPRESSIONS -- since round 5 default_pose_loss itself takes fused kernels on device
    # tensors; test_module_path_pose_loss_on_fused_kernels below pins that form against the same expressions)
    from latentfusion_amd.pose.loss import _pose_loss_expressions as default_pose_loss
    g6 = golden('g6_loss')
    target = _target(g6)
    target.depth[:, :, 100:140, 200:260] = 0.0            # invalid pixels inside the mask
    cam = prod_camera(g6['cam'])
    cam.viewport = cam.viewport.clone()
    gen = torch.Generator().manual_seed(crop)
    logits = (torch.randn(3, 2, crop, crop, generator=gen) * 2).to(DEV)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.7, 'mask': 0.5}
    wvec = torch.tensor([weights[k] for k in ('depth', 'ov_depth', 'iou', 'mask')], device=DEV)

    def module_path(lg, cam_):
        dl, ml = lg[:, :1], lg[:, 1:2]
        mask = torch.sigmoid(ml)
        depth = (torch.tanh(dl) + 1) * (mask > 0.5) - 1
        ld = default_pose_loss(target, cam_.denormalize_depth(depth), ml, cam_)
        return ld, sum(weights[k] * v for k, v in ld.items())

    lg1 = logits.clone().requires_grad_(True)
    for p in (cam.translation, cam.viewport):
        p.requires_grad_(True)
    ld, tot = module_path(lg1, cam)
    tot.mean().backward()
    ref_g = (lg1.grad.clone(), cam.translation.grad.clone(), cam.viewport.grad.clone())
    cam.translation.grad = None
    cam.viewport.grad = None

    lg2 = logits.clone().requires_grad_(True)
    coefs = camera_coefs(cam, 1.0, crop, crop)
    total, losses = pose_loss(lg2, coefs, target.depth.reshape(-1).contiguous(), target.mask.reshape(-1).contiguous(),
                              wvec, 480, 640)
    for i, k in enumerate(('depth', 'ov_depth', 'iou', 'mask')):
        close(losses[:, i], ld[k], atol=2e-6, rtol=2e-5)
    close(total, tot, atol=2e-6, rtol=2e-5)
    total.mean().backward()
    close(lg2.grad, ref_g[0], atol=1e-8, rtol=2e-3)
    close(cam.translation.grad, ref_g[1], atol=1e-7, rtol=2e-3)
    close(cam.viewport.grad, ref_g[2], atol=1e-7, rtol=2e-3)


def test_adam_kernel_matches_torch():
    from latentfusion_amd.pose.estimation import BatchedOptimizer
    g = torch.Generator().manual_seed(0)
    n = 6
    init = torch.randn(n, 10, generator=g)
    for name, cls in (('adam', torch.optim.Adam), ('adamw', torch.optim.AdamW)):
        ref = init.clone().requires_grad_(True)
        opt = cls([ref], lr=0.01)
        mine = init.clone().to(DEV).requires_grad_(True)
        bo = BatchedOptimizer(name, [mine])
        for step in range(20):
            gr = torch.randn(n, 10, generator=g)
            ref.grad = gr.clone()
            opt.step()
            mine.grad = gr.to(DEV)
            bo.step([0.01] * n)
        close(mine, ref, atol=2e-7, rtol=1e-5)


def test_engine_matches_module_path(golden):
    """RenderLoopEngine (explicit fused forward/backward) == Photographer.decode + loss through the
    generic autograd modules: losses and d/d(camera parameters)."""
    from latentfusion_amd.engine import RenderLoopEngine
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon.models import Photographer
    g = golden('g7_adam_trace')
    ph = Photographer.from_checkpoint(g['photographer']).to(DEV)
    target = _target(g)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    assert RenderLoopEngine.supports(ph, weights)

    class M:
        device = DEV
        photographer = ph
        input_size, camera_dist = 16, g['camera_dist']

        def render_latent_object(self, z_obj, camera, return_latent=True, apply_mask=True):
            y, z, _ = ph.decode(z_obj, camera, return_latent=return_latent, apply_mask=apply_mask)
            return y, z.squeeze(0)
    est = estimation.GradientPoseEstimator(model=M(), learning_rate=0.01, num_samples=8, num_iters=1, ranking_size=8,
                                           converge_threshold=1e-6, converge_patience=10, optimizer='adam',
                                           loss_weights=weights, use_engine=False)
    cam0 = prod_camera(g['init']).zoom(None, 16, g['camera_dist'])
    z_obj = g['z_obj'].to(DEV)
    st = est.start(z_obj, target, cam0)
    ld, _, rank, _ = est.loss_and_grad(z_obj, target, st['cam'])
    want_g = torch.cat((st['cam'].log_quaternion.grad, st['cam'].translation.grad, st['cam'].viewport.grad), dim=1)
    eng = RenderLoopEngine(ph, z_obj, target, weights)
    losses, gparams = eng.forward_backward(cam0)
    for i, k in enumerate(eng.LOSS_KEYS):
        close(losses[:, i], ld[k], atol=5e-6, rtol=5e-5)
    close(losses[:, 4], rank, atol=5e-6, rtol=5e-5)
    rel = ((gparams - want_g).norm(dim=1) / want_g.norm(dim=1)).max().item()
    assert rel < 2e-3, rel


@pytest.mark.parametrize('use_engine', [True, False])
def test_g7_gradient_loop_on_hip(golden, use_engine):
    """The full adam_quick loop on the HIP path reproduces the reference's iteration trace with
    IDENTICAL argmin pose indices at every iteration.  Losses agree to ~1e-7 relative on the first
    iterations (same inputs); afterwards Adam's gradient normalisation amplifies fp32 noise on
    near-zero gradient components (viewport gradients are ~1e-5 yet move by lr per step, SURVEY
    section 7), so per-sample trajectories drift by up to about one optimiser step (0.01) and the
    losses by < 1e-2 relative over the 10 recorded iterations -- without changing the ranking."""
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g7_adam_trace')
    model = LatentFusionModel(Sculptor.from_checkpoint(g['sculptor']), fusion.from_checkpoint(g['fuser']),
                              Photographer.from_checkpoint(g['photographer']), g['camera_dist'], DEV)
    est = estimation.load_from_config(copy.deepcopy(g['cfg']), model, track_stats=True, return_camera_history=True,
                                      use_engine=use_engine)
    best, stats, hist = est.estimate(g['z_obj'].to(DEV), _target(g, 'cpu'), camera=prod_camera(g['init'], 'cpu'))
    close(stats['rank_loss'][:3], g['rank_loss'][:3], atol=1e-6, rtol=1e-5)
    close(stats['rank_loss'], g['rank_loss'], atol=1e-4, rtol=1e-2)
    assert torch.argmin(stats['rank_loss'], dim=1).tolist() == g['argmin'].tolist()
    close(torch.stack([c.log_quaternion for _, c in hist]), g['hist_log_q'], atol=2e-2, rtol=0)
    close(torch.stack([c.translation for _, c in hist]), g['hist_t'], atol=1e-2, rtol=0)
    close(best.log_quaternion, g['best']['log_q'], atol=2e-2, rtol=0)
    assert len(best) == 8


def test_build_latent_object_end_to_end(golden):
    """Observation preprocessing (zoom / prepare / normalize) + encode on the device."""
    from latentfusion_amd import synth
    model, cks = synth.build_model(16, 8, 'gru', seed=3, device=DEV, bias_std=0.1)
    obs = synth.make_observation(4, seed=7, device=DEV)
    z = model.build_latent_object(obs)
    import lf_oracle as O
    from lf_oracle import pose as opose
    d = synth.make_observation_data(4, seed=7)
    oobs = opose.Obs(d['color'], d['depth'], d['mask'], O.Cam.from_extrinsic(d['intrinsic'], d['extrinsic']))
    want = opose.Model(*cks).build_latent_object(oobs)
    close(z, want, atol=2e-4, rtol=2e-3)


def test_cross_entropy_and_metropolis_estimators_run(golden):
    """Full CrossEntropy (GMM on the host, renders on the device) and Metropolis loops: losses of
    the returned ranking are sorted and no worse than the best initial sample."""
    from latentfusion_amd.pose import estimation, utils as pu
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g7_adam_trace')
    model = LatentFusionModel(Sculptor.from_checkpoint(g['sculptor']), fusion.from_checkpoint(g['fuser']),
                              Photographer.from_checkpoint(g['photographer']), g['camera_dist'], DEV)
    target = _target(g, 'cpu')
    z_obj = g['z_obj'].to(DEV)
    torch.manual_seed(0)
    import numpy as np
    np.random.seed(0)
    ce = estimation.CrossEntropyPoseEstimator(model=model, num_samples=16, num_elites=6, num_iters=3, num_gmm_components=2,
                                              learning_rate=0.9, sample_flipped=True, ranking_size=4,
                                              loss_weights={'depth': 1.0, 'ov_depth': 0.2}, return_camera_history=True)
    init = pu.sample_cameras_with_estimate(12, target.camera)
    cams, hist = ce.estimate(z_obj, target, cameras=init)
    assert len(cams) == 4 and len(hist) >= 1
    _, loss0 = ce.evaluate_samples(z_obj, target.to(DEV), cams.to(DEV))
    assert torch.isfinite(loss0).all()
    mp = estimation.MetropolisPoseEstimator(model=model, num_samples=4, num_iters=2, ranking_size=2,
                                            loss_weights={'depth': 1.0, 'latent': 0.1})
    out = mp.estimate(z_obj, _target_with_color(g), camera=target.camera)
    assert len(out) == 2


def _target_with_color(g):
    from latentfusion_amd.observation import Observation
    tg = g['target']
    color = torch.rand(1, 3, 480, 640, generator=torch.Generator().manual_seed(1))
    return Observation(color, tg['depth'], tg['mask'].float(), prod_camera(tg['cam'], 'cpu'))


def test_split_precision_conv_matches_fp32_and_fp64():
    """lf_conv3d_c16_split (three f16 MFMAs per product) vs the exact-fp32 MFMA kernel and an fp64
    reference: forward with the fused epilogue, and the amax-scaled data gradient on a tiny-valued
    input.  The split kernel must be at least as close to fp64 as the fp32 kernel (x2 slack)."""
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(0)
    N, S = 2, 24
    x = torch.randn(N, 16, S, S, S, generator=g)
    x = x / torch.sqrt((x ** 2).mean(dim=1, keepdim=True))
    w = torch.randn(16, 16, 3, 3, 3, generator=g)
    b = torch.randn(16, generator=g) * 0.1
    he = ops.he_constant(w)
    y64 = torch.nn.functional.conv3d(x.double(), w.double(), None, 1, 1) * he + b.double().view(1, -1, 1, 1, 1)
    y64 = torch.nn.functional.leaky_relu(y64, 0.2)
    y64 = y64 / torch.sqrt((y64 ** 2).mean(dim=1, keepdim=True) + 1e-8)
    xd, wd, bd = ops.cl(x.to(DEV)), w.to(DEV), b.to(DEV)
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    ref, _ = ops._conv3x3_raw(xd, ops.pack_conv3x3(wd), bd, 16, he, flags, True)
    got, _ = ops.conv3d_c16_split(xd, ops.pack_conv3d_c16_split(wd), bd, he, flags)
    e_ref = (ref.cpu().double() - y64).abs().max().item()
    e_got = (got.cpu().double() - y64).abs().max().item()
    assert e_got < max(2 * e_ref, 5e-6), (e_got, e_ref)
    close(got, ref, atol=3e-5, rtol=1e-4)
    gs = xd * 3e-7
    amax = ops.amax_buffer(gs.abs().max(), DEV)
    gref = ops.conv3x3_bwd_data(gs, ops.pack_conv3x3(wd, transpose=True), 16, he, None)
    ggot, _ = ops.conv3d_c16_split(gs, ops.pack_conv3d_c16_split(wd, transpose=True), None, he, 0, amax_in=amax)
    assert (ggot - gref).abs().max().item() < 1e-5 * gref.abs().max().item()


@pytest.mark.parametrize('shape', [(2, 16, 10, 20, 40), (1, 16, 5, 7, 9), (3, 16, 4, 8, 16), (1, 16, 13, 24, 33), (2, 16, 36, 64, 64)])
def test_split_conv3d_matches_fp64(shape):
    """lf_conv3d_c16_split (direct, three f16 MFMAs per product, z-sliding halo ring) on volumes with partial tiles on
    every axis and, for the last shape, several tiles and a column change per workgroup: fused forward, raw data
    gradient, and the data gradient with the fused previous-layer backward on a tiny-valued gradient through the max-abs
    side channel.  Must be as close to fp64 as the direct fp32 MFMA kernel (x2 slack)."""
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(sum(shape))
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    w = torch.randn(16, 16, 3, 3, 3, generator=g)
    b = torch.randn(16, generator=g) * 0.1
    he = ops.he_constant(w)
    xs = torch.randn(*shape, generator=g)
    xs = xs / torch.sqrt((xs ** 2).mean(dim=1, keepdim=True))
    y64 = torch.nn.functional.conv3d(xs.double(), w.double(), None, 1, 1) * he + b.double().view(1, -1, 1, 1, 1)
    y64 = torch.nn.functional.leaky_relu(y64, 0.2)
    n64 = torch.sqrt((y64 ** 2).mean(dim=1, keepdim=True) + 1e-8)
    y64 = y64 / n64
    xd, wd, bd = ops.cl(xs.to(DEV)), w.to(DEV), b.to(DEV)
    sp, spt = ops.pack_conv3d_c16_split(wd), ops.pack_conv3d_c16_split(wd, transpose=True)
    ref, nref = ops._conv3x3_raw(xd, ops.pack_conv3x3(wd), bd, 16, he, flags, True)
    got, ngot = ops.conv3d_c16_split(xd, sp, bd, he, flags)
    e_ref = (ref.cpu().double() - y64).abs().max().item()
    e_got = (got.cpu().double() - y64).abs().max().item()
    assert e_got < max(2 * e_ref, 5e-6), (e_got, e_ref)
    assert (ngot.cpu().double().view(n64.shape) - n64).abs().max().item() < 2e-6
    g64 = torch.nn.functional.conv_transpose3d(xs.double(), w.double(), None, 1, 1) * he
    gw, _ = ops.conv3d_c16_split(xd, spt, None, he, 0)
    gd = ops.conv3x3_bwd_data(xd, ops.pack_conv3x3(wd, transpose=True), 16, he, None)
    assert (gw.cpu().double() - g64).abs().max().item() < max(2 * (gd.cpu().double() - g64).abs().max().item(), 5e-6)
    prev = (ref, nref, flags)
    tiny = xd * 3e-7
    want = ops.conv3x3_bwd_data(tiny, ops.pack_conv3x3(wd, transpose=True), 16, he, prev)
    am_in, am_out = ops.amax_buffer(tiny.abs().max(), DEV), ops.amax_buffer(None, DEV)
    gw2, _ = ops.conv3d_c16_split(tiny, spt, None, he, 0, prev=prev, amax_in=am_in, amax_out=am_out)
    assert (gw2 - want).abs().max().item() < 2e-6 * want.abs().max().item()
    assert abs(am_out.max().item() - gw2.abs().max().item()) == 0.0


def test_split_conv3d_is_run_to_run_identical():
    """The split kernel spreads the previous tile's epilogue over the MFMA phase of the next one; a register hazard between
    the two (seen with SLP-packed fp32 arithmetic in that phase: rare wrong elements in the gradient form) shows up as
    run-to-run differences.  Ten launches of both forms on a volume with many tiles and column changes per workgroup
    must be bit-identical and agree with the all-fp32 Winograd kernel."""
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(5)
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    shape = (4, 16, 76, 96, 96)                                    # 38 x 12 x 6 x 4 = 10,944 tiles: > 20 per workgroup
    w = torch.randn(16, 16, 3, 3, 3, generator=g).to(DEV)
    b = (torch.randn(16, generator=g) * 0.1).to(DEV)
    he = ops.he_constant(w)
    x = torch.randn(*shape, generator=g)
    x = ops.cl((x / torch.sqrt((x ** 2).mean(dim=1, keepdim=True))).to(DEV))
    gin = ops.cl((torch.randn(*shape, generator=g) * 1e-7).to(DEV))
    sp, spt = ops.pack_conv3d_c16_split(w), ops.pack_conv3d_c16_split(w, transpose=True)
    yw, nw = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w), b, he, flags)
    gwant, _ = ops.conv3d_c16_wino(gin, ops.pack_conv3d_c16_wino(w, transpose=True), None, he, 0, prev=(yw, nw, flags))
    am = ops.amax_buffer(gin.abs().max(), DEV)
    y0 = g0 = None
    for rep in range(10):
        y, _ = ops.conv3d_c16_split(x, sp, b, he, flags)
        gg, _ = ops.conv3d_c16_split(gin, spt, None, he, 0, prev=(yw, nw, flags), amax_in=am)
        if rep == 0:
            y0, g0 = y, gg
            assert (y - yw).abs().max().item() < 3e-5
            assert (gg - gwant).abs().max().item() < 5e-6 * gwant.abs().max().item()
        else:
            assert torch.equal(y, y0) and torch.equal(gg, g0), rep


@pytest.mark.parametrize('shape', [(2, 16, 10, 20, 40), (1, 16, 5, 7, 9), (3, 16, 4, 8, 16), (1, 16, 13, 24, 33)])
def test_winograd_conv3d_matches_fp64(shape):
    """lf_conv3d_c16_wino (F(2x2x2,3x3x3), all-fp32) on volumes with partial tiles on every axis:
    fused forward, raw data gradient, data gradient with the fused previous-layer backward and the
    max-abs side output.  Must be as close to fp64 as the direct fp32 MFMA kernel (x2 slack)."""
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(sum(shape))
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    w = torch.randn(16, 16, 3, 3, 3, generator=g)
    b = torch.randn(16, generator=g) * 0.1
    he = ops.he_constant(w)
    xs = torch.randn(*shape, generator=g)
    xs = xs / torch.sqrt((xs ** 2).mean(dim=1, keepdim=True))
    y64 = torch.nn.functional.conv3d(xs.double(), w.double(), None, 1, 1) * he + b.double().view(1, -1, 1, 1, 1)
    y64 = torch.nn.functional.leaky_relu(y64, 0.2)
    n64 = torch.sqrt((y64 ** 2).mean(dim=1, keepdim=True) + 1e-8)
    y64 = y64 / n64
    xd, wd, bd = ops.cl(xs.to(DEV)), w.to(DEV), b.to(DEV)
    up, upt = ops.pack_conv3d_c16_wino(wd), ops.pack_conv3d_c16_wino(wd, transpose=True)
    ref, nref = ops._conv3x3_raw(xd, ops.pack_conv3x3(wd), bd, 16, he, flags, True)
    got, ngot = ops.conv3d_c16_wino(xd, up, bd, he, flags)
    e_ref = (ref.cpu().double() - y64).abs().max().item()
    e_got = (got.cpu().double() - y64).abs().max().item()
    assert e_got < max(2 * e_ref, 5e-6), (e_got, e_ref)
    assert (ngot.cpu().double().view(n64.shape) - n64).abs().max().item() < 2e-6
    g64 = torch.nn.functional.conv_transpose3d(xs.double(), w.double(), None, 1, 1) * he
    gw, _ = ops.conv3d_c16_wino(xd, upt, None, he, 0)
    gd = ops.conv3x3_bwd_data(xd, ops.pack_conv3x3(wd, transpose=True), 16, he, None)
    assert (gw.cpu().double() - g64).abs().max().item() < max(2 * (gd.cpu().double() - g64).abs().max().item(), 5e-6)
    prev = (ref, nref, flags)
    want = ops.conv3x3_bwd_data(xd, ops.pack_conv3x3(wd, transpose=True), 16, he, prev)
    amax = ops.amax_buffer(None, DEV)
    gw2, _ = ops.conv3d_c16_wino(xd, upt, None, he, 0, prev=prev, amax_out=amax)
    assert (gw2 - want).abs().max().item() < 2e-6 * want.abs().max().item()
    assert amax.max().item() == gw2.abs().max().item()
    # split-precision products (three f16 MFMAs per Winograd-domain product): same bars, plus a tiny-valued
    # gradient tensor through the amax pre-scaling
    us, ust = ops.pack_conv3d_c16_wino_split(wd), ops.pack_conv3d_c16_wino_split(wd, transpose=True)
    gs, ngs = ops.conv3d_c16_wino_split(xd, us, bd, he, flags)
    assert (gs.cpu().double() - y64).abs().max().item() < max(2 * e_ref, 5e-6)
    assert (ngs.cpu().double().view(n64.shape) - n64).abs().max().item() < 2e-6
    tiny = xd * 3e-7
    am_in, am_out = ops.amax_buffer(tiny.abs().max(), DEV), ops.amax_buffer(None, DEV)
    gt, _ = ops.conv3d_c16_wino_split(tiny, ust, None, he, 0, prev=prev, amax_in=am_in, amax_out=am_out)
    want_t = ops.conv3x3_bwd_data(tiny, ops.pack_conv3x3(wd, transpose=True), 16, he, prev)
    assert (gt - want_t).abs().max().item() < 1e-5 * want_t.abs().max().item()
    assert am_out.max().item() == gt.abs().max().item()


def test_engine_winograd_matches_module_path(golden):
    """RenderLoopEngine with the Winograd conv kernels == Photographer.decode + loss through the generic
    autograd modules (direct conv kernels) on a SYN(16,16) model: losses and camera gradients."""
    from latentfusion_amd import synth
    from latentfusion_amd.experimental import RenderLoopEngineX as RenderLoopEngine   # (variants beyond the product engine)
    from latentfusion_amd.pose import estimation
    model, _ = synth.build_model(16, 16, 'pool:mean', seed=4, device=DEV, bias_std=0.05)
    g = golden('g7_adam_trace')
    target = _target(g)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    gen = torch.Generator().manual_seed(9)
    z_obj = torch.randn(1, 1, 16, 16, 16, 16, generator=gen).to(DEV)
    est = estimation.GradientPoseEstimator(model=model, learning_rate=0.01, num_samples=8, num_iters=1, ranking_size=8,
                                           converge_threshold=1e-6, converge_patience=10, optimizer='adam',
                                           loss_weights=weights, use_engine=False)
    cam0 = prod_camera(g['init']).zoom(None, model.input_size, model.camera_dist)
    st = est.start(z_obj, target, cam0)
    ld, _, rank, _ = est.loss_and_grad(z_obj, target, st['cam'])
    want_g = torch.cat((st['cam'].log_quaternion.grad, st['cam'].translation.grad, st['cam'].viewport.grad), dim=1)
    for mode in ('winograd', 'fp32', 'f16x3', 'winograd_f16x3'):
        eng = RenderLoopEngine(model.photographer, z_obj, target, weights, conv_mode=mode)
        assert eng.conv_mode == mode
        losses, gparams = eng.forward_backward(cam0)
        for i, k in enumerate(eng.LOSS_KEYS):
            close(losses[:, i], ld[k], atol=5e-6, rtol=5e-5)
        close(losses[:, 4], rank, atol=5e-6, rtol=5e-5)
        rel = ((gparams - want_g).norm(dim=1) / want_g.norm(dim=1)).max().item()
        assert rel < 2e-3, (mode, rel)
    assert RenderLoopEngine(model.photographer, z_obj, target, weights).conv_mode == 'winograd'


def test_g7_gradient_loop_split_precision(golden):
    """The adam_quick loop with the Winograd and the f16x3 conv kernels against the direct fp32 kernels:
    same argmin indices at every iteration, losses within the same envelope."""
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    from latentfusion_amd import synth
    # G7's model has C = 8; the split kernels are built for C = 16, so use a SYN(16,16) model and compare
    # the two engine modes against each other over a 10-iteration run
    model, _ = synth.build_model(16, 16, 'pool:mean', seed=4, device=DEV, bias_std=0.05)
    g = golden('g7_adam_trace')
    gen = torch.Generator().manual_seed(9)
    z_obj = torch.randn(1, 1, 16, 16, 16, 16, generator=gen).to(DEV)
    out = {}
    for mode in ('fp32', 'winograd', 'f16x3', 'winograd_f16x3'):
        est = estimation.load_from_config(copy.deepcopy(g['cfg']), model, track_stats=True, conv_mode=mode)
        _, stats = est.estimate(z_obj, _target(g, 'cpu'), camera=prod_camera(g['init'], 'cpu'))
        out[mode] = stats['rank_loss']
    for mode in ('winograd', 'f16x3', 'winograd_f16x3'):
        close(out[mode][:3], out['fp32'][:3], atol=1e-6, rtol=2e-5)
        # later iterations: Adam amplifies last-bit differences across voxel-cell boundaries (DESIGN Q17);
        # the fp32 kernels drift by the same order against the CPU oracle (test_g7_gradient_loop_on_hip)
        close(out[mode], out['fp32'], atol=3e-2, rtol=0)
        # same best hypothesis wherever the direct-kernel run separates best and runner-up by more than that
        top2 = torch.sort(out['fp32'], dim=1).values[:, :2]
        clear = (top2[:, 1] - top2[:, 0]) > 6e-2
        assert (torch.argmin(out[mode], dim=1) == torch.argmin(out['fp32'], dim=1))[clear].all()
        assert torch.argmin(out[mode][0]) == torch.argmin(out['fp32'][0])


def test_conv3d_c16_kernels_on_random_shapes():
    """Winograd / split-Winograd / weight-gradient kernels against the direct fp32 kernels on random small and
    ragged volumes (single-plane, narrower than a tile, odd sizes): guards the tile-boundary handling."""
    import random
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_LRELU, LF_EPI_PIXELNORM
    rnd = random.Random(7)
    g = torch.Generator().manual_seed(7)
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    shapes = [(1, 1, 1, 1), (1, 1, 7, 3), (2, 3, 1, 17), (1, 2, 8, 16), (3, 5, 9, 33)]
    shapes += [(rnd.choice([1, 2, 3]), rnd.randint(1, 21), rnd.randint(1, 37), rnd.randint(1, 45)) for _ in range(10)]
    for N, D, H, W in shapes:
        x = ops.cl(torch.randn(N, 16, D, H, W, generator=g).to(DEV))
        w = torch.randn(16, 16, 3, 3, 3, generator=g).to(DEV)
        b = (torch.randn(16, generator=g) * 0.1).to(DEV)
        he = ops.he_constant(w)
        ref, nref = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, flags, True)
        got, ngot = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w), b, he, flags)
        gs, _ = ops.conv3d_c16_wino_split(x, ops.pack_conv3d_c16_wino_split(w), b, he, flags)
        prev = (ref, nref, flags)
        gref = ops.conv3x3_bwd_data(x, ops.pack_conv3x3(w, transpose=True), 16, he, prev)
        gw, _ = ops.conv3d_c16_wino(x, ops.pack_conv3d_c16_wino(w, transpose=True), None, he, 0, prev=prev)
        assert (got - ref).abs().max().item() < 5e-5, (N, D, H, W)
        assert (gs - ref).abs().max().item() < 5e-5, (N, D, H, W)
        assert (ngot - nref).abs().max().item() < 5e-5, (N, D, H, W)
        assert (gw - gref).abs().max().item() < 5e-5 * max(gref.abs().max().item(), 1.0), (N, D, H, W)
        gwt, _ = ops.conv_bwd_weight(x, ref, 3, 16, he)
        want = torch.nn.grad.conv3d_weight(x.double().cpu(), (16, 16, 3, 3, 3), ref.double().cpu(), padding=1) * he
        gotw = gwt.reshape(3, 3, 3, 16, 16).permute(3, 4, 0, 1, 2).double().cpu()
        assert (gotw - want).abs().max().item() < 1e-4 * max(want.abs().max().item(), 1e-3), (N, D, H, W)


@pytest.mark.parametrize('shape', [(4, 16, 14, 48, 96), (3, 16, 22, 64, 64), (1, 16, 30, 40, 200), (8, 16, 32, 64, 64),
                                   (2, 16, 2, 128, 256), (2, 16, 3, 96, 160)])
def test_winograd_sliding_halo_and_addend(shape):
    """Volumes with more tiles than resident workgroups, so every workgroup walks several tiles of a column and
    the halo slides along z (LDS -> LDS move + two fetched planes), with tile ranges that start and end in the
    middle of columns; both precisions, forward and fused backward, against the direct fp32 kernels.  Also the
    addend form (prev_flags = LF_EPI_ADD): conv(x) * he + bias + addend, then the epilogue."""
    from latentfusion_amd import ops
    from latentfusion_amd._lib import LF_EPI_ADD, LF_EPI_LRELU, LF_EPI_PIXELNORM
    g = torch.Generator().manual_seed(sum(shape))
    flags = LF_EPI_LRELU | LF_EPI_PIXELNORM
    x = ops.cl(torch.randn(*shape, generator=g).to(DEV))
    w = torch.randn(16, 16, 3, 3, 3, generator=g).to(DEV)
    b = (torch.randn(16, generator=g) * 0.1).to(DEV)
    he = ops.he_constant(w)
    up, upt = ops.pack_conv3d_c16_wino(w), ops.pack_conv3d_c16_wino(w, transpose=True)
    ref, nref = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, flags, True)
    got, ngot = ops.conv3d_c16_wino(x, up, b, he, flags)
    assert (got - ref).abs().max().item() < 5e-5
    assert (ngot - nref).abs().max().item() < 5e-5
    gs, _ = ops.conv3d_c16_wino_split(x, ops.pack_conv3d_c16_wino_split(w), b, he, flags)
    assert (gs - ref).abs().max().item() < 5e-5
    prev = (ref, nref, flags)
    gref = ops.conv3x3_bwd_data(x, ops.pack_conv3x3(w, transpose=True), 16, he, prev)
    gw, _ = ops.conv3d_c16_wino(x, upt, None, he, 0, prev=prev)
    assert (gw - gref).abs().max().item() < 5e-5 * max(gref.abs().max().item(), 1.0)
    # addend form: plain (no activation) and with the full epilogue
    addend = ops.cl(torch.randn(*shape, generator=g).to(DEV))
    raw, _ = ops._conv3x3_raw(x, ops.pack_conv3x3(w), b, 16, he, 0, False)
    plain, _ = ops.conv3d_c16_wino(x, up, b, he, 0, prev=(addend, None, LF_EPI_ADD))
    assert (plain - (raw + addend)).abs().max().item() < 5e-5
    act = torch.nn.functional.leaky_relu(raw + addend, 0.2)
    norm = torch.sqrt((act ** 2).mean(dim=1, keepdim=True) + 1e-8)
    full, nfull = ops.conv3d_c16_wino(x, up, b, he, flags, prev=(addend, None, LF_EPI_ADD))
    assert (full - act / norm).abs().max().item() < 5e-5
    assert (nfull.view(norm.shape) - norm).abs().max().item() < 5e-5


def test_engine_hypothesis_groups_on_streams_are_bit_identical(golden):
    """RenderLoopEngine.set_streams(k): the N hypotheses evaluated as k groups on k HIP streams (the small-kernel stretch of
    one group overlaps the volume kernels of another): the SAME losses bit for bit as one group on one stream, gradients equal
    to rounding (the deterministic reductions partition by group size), run-to-run identical; a whole adam loop ranks alike."""
    from latentfusion_amd import synth
    from latentfusion_amd.experimental import RenderLoopEngineX as RenderLoopEngine   # (variants beyond the product engine)
    from latentfusion_amd.pose import estimation
    model, _ = synth.build_model(32, 16, 'pool:mean', seed=4, device=DEV, bias_std=0.05)
    g = golden('g7_adam_trace')
    target = _target(g)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    gen = torch.Generator().manual_seed(9)
    z_obj = torch.randn(1, 1, 16, 32, 32, 32, generator=gen).to(DEV)
    cam0 = prod_camera(g['init']).zoom(None, model.input_size, model.camera_dist)          # 8 hypotheses
    eng = RenderLoopEngine(model.photographer, z_obj, target, weights)
    l1, g1 = eng.forward_backward(cam0)
    for k in (2, 4):
        eng.set_streams(k)
        for _ in range(3):                                       # repeated: stream hand-offs, allocator reuse across streams
            lk, gk = eng.forward_backward(cam0)
            torch.cuda.synchronize()
            assert torch.equal(lk, l1), k
            # (the coefficient gradient's block partition follows the group size: its fixed-order sums associate differently;
            # components that are small through cancellation move by a few 1e-7 absolute)
            close(gk, g1, atol=1e-6 * g1.abs().max().item(), rtol=2e-5)
    eng.set_streams(3)                                           # 8 -> 2 + 3 + 3: factors 2/8, 3/8
    l3, g3 = eng.forward_backward(cam0)
    assert torch.equal(l3, l1)
    close(g3, g1, atol=1e-6 * g1.abs().max().item(), rtol=2e-5)
    cam5 = cam0[:5]
    eng.set_streams(1)
    l5, g5 = eng.forward_backward(cam5)
    eng.set_streams(2)                                           # 5 -> 2 + 3
    l52, g52 = eng.forward_backward(cam5)
    assert torch.equal(l52, l5)
    close(g52, g5, atol=1e-6 * g5.abs().max().item(), rtol=2e-5)
    runs = []
    for k in (1, 2):
        est = estimation.GradientPoseEstimator(model=model, learning_rate=0.01, num_samples=8, num_iters=6, ranking_size=8,
                                               converge_threshold=1e-9, converge_patience=100, optimizer='adam',
                                               loss_weights=weights, engine_streams=k, return_camera_history=True)
        best, hist = est.estimate(z_obj, target, camera=prod_camera(g['init']))
        runs.append((torch.cat((best.log_quaternion, best.translation), dim=1).cpu(), torch.stack([h[0] for h in hist])))
    # iteration 0 is the same evaluation; later iterations drift apart at the rate any last-bit gradient difference does
    # (Adam turns ~1e-5-sized viewport gradients into lr-sized steps: SURVEY 7 "chaotic sensitivity") -- within one
    # optimiser step (lr = 0.01) per iteration, and the same best hypothesis at the end
    assert torch.equal(runs[0][1][0], runs[1][1][0])
    close(runs[0][1][:3], runs[1][1][:3], atol=0.0, rtol=2e-2)
    close(runs[0][1], runs[1][1], atol=0.0, rtol=1e-1)
    close(runs[0][0][0], runs[1][0][0], atol=6 * 0.01, rtol=0.0)          # the best hypothesis (lower ranks may swap near-ties)
    assert int(torch.argmin(runs[0][1][-1])) == int(torch.argmin(runs[1][1][-1]))


def test_engine_latent_term_matches_module_path(golden):
    """adam_latent-style weights (latent = 0.2): the fused engine with the latent term (cosine distance between the projected
    latent of every hypothesis and the target's latent code under that hypothesis, reference pose/estimation.py:112-116,606-608)
    == Photographer.decode + default_pose_loss through the autograd modules: every loss term, the ranking loss, the camera
    gradients; the estimator takes the engine for this preset and its first iterations follow the module path's."""
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g12_latent_code')
    model = LatentFusionModel(Sculptor.from_checkpoint(g['sculptor']), fusion.from_checkpoint(g['fuser']),
                              Photographer.from_checkpoint(g['photographer']), g['camera_dist'], DEV)
    tg = g['target']
    target = Observation(tg['color_u8'].float() / 255.0, tg['depth'], tg['mask'].float(), prod_camera(tg['cam'], 'cpu')).to(DEV)
    z_obj = g['z_obj'].to(DEV)
    cam0 = prod_camera(g['cams'])
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.0, 'mask': 0.0, 'latent': 0.2}
    kw = dict(model=model, learning_rate=0.01, num_samples=len(cam0), num_iters=3, ranking_size=len(cam0), converge_threshold=1e-9,
              converge_patience=100, optimizer='adam', loss_weights=weights, return_camera_history=True)
    ref = estimation.GradientPoseEstimator(use_engine=False, **kw)
    with model.frozen():
        st = ref.start(z_obj, target, cam0)
        with torch.no_grad():                                    # as the estimators do (reference :606-608)
            zt = model.compute_latent_code(target, st['cam'])
        ld, _, rank, _ = ref.loss_and_grad(z_obj, target, st['cam'], 0, zt)
        want_g = torch.cat((st['cam'].log_quaternion.grad, st['cam'].translation.grad, st['cam'].viewport.grad), dim=1)
        eng_est = estimation.GradientPoseEstimator(**kw)
        st2 = eng_est.start(z_obj, target, cam0)
        assert 'engine' in st2 and st2['engine'].w_latent == 0.2
        losses, gparams = st2['engine'].forward_backward(st2['cam'], z_target_latent=zt)
    for i, k in enumerate(('depth', 'ov_depth', 'iou', 'mask')):
        close(losses[:, i], ld[k], atol=5e-6, rtol=1e-4)
    close(losses[:, 5], ld['latent'], atol=2e-6, rtol=1e-4)
    close(losses[:, 4], rank, atol=5e-6, rtol=1e-4)
    rel = ((gparams - want_g).norm(dim=1) / want_g.norm(dim=1)).max().item()
    assert rel < 5e-3, rel
    a, ha = ref.estimate(z_obj, target, camera=prod_camera(g['cams']))
    b, hb = eng_est.estimate(z_obj, target, camera=prod_camera(g['cams']))
    la, lb = torch.stack([h[0] for h in ha]), torch.stack([h[0] for h in hb])
    close(lb[0], la[0], atol=5e-6, rtol=1e-4)
    close(lb, la, atol=0.0, rtol=2e-2)


def test_cross_entropy_scoring_runs_on_the_engine_and_matches_the_module_path(golden):
    """CrossEntropyPoseEstimator.evaluate_samples on the fused engine (forward only, lf_pose_loss_fwd_masked: the crop depth is
    multiplied by the crop's sigmoid mask before the uncrop, reference pose/estimation.py:207-216) against the same estimator
    on the generic module path and against the reference's golden losses / order (g8); then with a latent term."""
    from latentfusion_amd import synth
    from latentfusion_amd.engine import RenderLoopEngine
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g, t7 = golden('g8_ce_step'), golden('g7_adam_trace')
    model = LatentFusionModel(Sculptor.from_checkpoint(t7['sculptor']), fusion.from_checkpoint(t7['fuser']),
                              Photographer.from_checkpoint(t7['photographer']), t7['camera_dist'], DEV)
    tg = t7['target']
    target = Observation(None, tg['depth'], tg['mask'].float(), prod_camera(tg['cam'], 'cpu')).to(DEV)
    z_obj = t7['z_obj'].to(DEV)
    kw = dict(model=model, num_samples=24, num_elites=5, num_iters=3, num_gmm_components=2, learning_rate=0.9,
              sample_flipped=True, ranking_size=4, loss_weights=g['weights'])
    on, off = estimation.CrossEntropyPoseEstimator(**kw), estimation.CrossEntropyPoseEstimator(use_engine=False, **kw)
    _, l_on = on.evaluate_samples(z_obj, target, prod_camera(g['cams']))
    _, l_off = off.evaluate_samples(z_obj, target, prod_camera(g['cams']))
    assert isinstance(on._ranking_engine(z_obj, target), RenderLoopEngine) and off._ranking_engine(z_obj, target) is None
    close(l_on, l_off, atol=2e-6, rtol=2e-5)
    close(l_on, g['loss'], atol=1e-4, rtol=1e-3)
    assert torch.argsort(l_on).cpu().tolist() == g['order'].tolist()
    # the masked form is NOT the gradient estimator's form: the two must differ where the mask is soft
    eng = on._ranking_engine(z_obj, target)
    zc = prod_camera(g['cams']).zoom(None, model.input_size, model.camera_dist).to(DEV)
    plain, _ = eng.forward_backward(zc, need_grad=False)
    masked, _ = eng.forward_backward(zc, need_grad=False, masked_depth=True)
    assert not torch.equal(plain[:, 0], masked[:, 0]) and torch.equal(plain[:, 2:4], masked[:, 2:4])
    with pytest.raises(ValueError):
        eng.forward_backward(zc, need_grad=True, masked_depth=True)

    # SYN(16,16) (Winograd kernels, fused projection) with a latent term: engine == modules, per sample and in order
    model2, _ = synth.build_model(16, 16, 'pool:mean', seed=4, device=DEV, bias_std=0.05)
    gen = torch.Generator().manual_seed(9)
    z2 = torch.randn(1, 1, 16, 16, 16, 16, generator=gen).to(DEV)
    tcol = _target_with_color(t7).to(DEV)
    w2 = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4, 'latent': 0.5}
    kw2 = dict(model=model2, num_samples=16, num_elites=4, num_iters=2, num_gmm_components=2, learning_rate=0.9,
               sample_flipped=True, ranking_size=4, loss_weights=w2)
    cams = prod_camera(t7['init'])[:4]
    a_ = estimation.CrossEntropyPoseEstimator(**kw2).evaluate_samples(z2, tcol, cams)[1]
    b_ = estimation.CrossEntropyPoseEstimator(use_engine=False, **kw2).evaluate_samples(z2, tcol, cams)[1]
    assert a_.shape == (16,)
    close(a_, b_, atol=5e-6, rtol=5e-5)
    assert torch.equal(torch.argsort(a_), torch.argsort(b_))
    # a latent weight without the target's code is an error, not a silently shorter loss
    eng2 = RenderLoopEngine(model2.photographer, z2, tcol, w2)
    with pytest.raises(ValueError):
        eng2.forward_backward(cams.zoom(None, model2.input_size, model2.camera_dist).to(DEV), need_grad=False)


@pytest.mark.parametrize('variant', ['factor', 'sum', 'occlusion'])
def test_engine_renders_the_sum_and_occlusion_variants(golden, variant):
    """RenderLoopEngine on the three renderer variants of the reference (recon/models.py:378-395,427-437: 'factor' projection,
    'sum' composite, occlusion softmax over the depth column) against Photographer.decode + default_pose_loss through the
    generic modules on the golden fixture's network (g5): the four loss terms, their weighted sum and the camera gradients;
    the forward-only ranking form as well."""
    from latentfusion_amd.engine import RenderLoopEngine
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon.models import Photographer
    r = golden('g5_decode')[variant]
    ph = Photographer.from_checkpoint(r['ck']).to(DEV)
    for p in ph.parameters():
        p.requires_grad_(False)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    assert RenderLoopEngine.supports(ph, weights)
    target = _target(golden('g7_adam_trace'))
    z_obj = r['z_obj'].to(DEV)

    class _M:                                                     # the facade surface the estimator touches
        photographer, device, input_size, camera_dist = ph, torch.device(DEV), ph.in_size, 1.0

        @staticmethod
        def render_latent_object(z, cam, return_latent=True, apply_mask=True):
            y, zl, _ = ph.decode(z, cam, return_latent=return_latent, apply_mask=apply_mask)
            return y, (zl.squeeze(0) if return_latent else zl)
    cam0 = prod_camera(r['cam'])
    est = estimation.GradientPoseEstimator(model=_M, learning_rate=0.01, num_samples=len(cam0), num_iters=1, ranking_size=len(cam0),
                                           converge_threshold=1e-6, converge_patience=10, optimizer='adam', loss_weights=weights,
                                           use_engine=False)
    st = est.start(z_obj, target, cam0)
    assert 'engine' not in st
    ld, _, rank, _ = est.loss_and_grad(z_obj, target, st['cam'])
    want_g = torch.cat((st['cam'].log_quaternion.grad, st['cam'].translation.grad, st['cam'].viewport.grad), dim=1)
    eng = RenderLoopEngine(ph, z_obj, target, weights)
    assert eng.generic_tail == (variant != 'factor')
    losses, gparams = eng.forward_backward(cam0)
    for i, k in enumerate(eng.LOSS_KEYS):
        close(losses[:, i], ld[k], atol=5e-6, rtol=5e-5)
    close(losses[:, 4], rank, atol=5e-6, rtol=5e-5)
    rel = ((gparams - want_g).norm(dim=1) / want_g.norm(dim=1)).max().item()
    assert rel < 2e-3, (variant, rel)
    fwd_only, none = eng.forward_backward(cam0, need_grad=False)
    assert none is None and torch.equal(fwd_only, losses)
    # the estimator now takes the engine for these renderers
    est2 = estimation.GradientPoseEstimator(model=_M, learning_rate=0.01, num_samples=len(cam0), num_iters=1, ranking_size=len(cam0),
                                            converge_threshold=1e-6, converge_patience=10, optimizer='adam', loss_weights=weights)
    assert 'engine' in est2.start(z_obj, target, cam0)


def test_engine_first_call_on_several_streams(golden):
    """a FRESH engine (fresh model: no weight pack exists yet) whose very first evaluation is multi-stream must give
    the single-stream result -- the packs are built on the current stream before any side stream reads them."""
    from latentfusion_amd import synth
    from latentfusion_amd.experimental import RenderLoopEngineX as RenderLoopEngine   # (variants beyond the product engine)
    g = golden('g7_adam_trace')
    target = _target(g)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    gen = torch.Generator().manual_seed(9)
    z_obj = torch.randn(1, 1, 16, 32, 32, 32, generator=gen).to(DEV)
    outs = []
    for k in (1, 4, 2):
        model, _ = synth.build_model(32, 16, 'pool:mean', seed=4, device=DEV, bias_std=0.05)     # new parameters: no cached packs
        model.freeze()
        cam0 = prod_camera(g['init']).zoom(None, model.input_size, model.camera_dist)
        eng = RenderLoopEngine(model.photographer, z_obj, target, weights).set_streams(k)
        first = eng.forward_backward(cam0)
        second = eng.forward_backward(cam0)                       # this one IS multi-stream
        torch.cuda.synchronize()
        outs.append((first, second))
    for first, second in outs:
        assert torch.equal(first[0], outs[0][0][0]) and torch.equal(second[0], outs[0][0][0])
        close(first[1], outs[0][0][1], atol=1e-6 * outs[0][0][1].abs().max().item(), rtol=2e-5)
        close(second[1], outs[0][0][1], atol=1e-6 * outs[0][0][1].abs().max().item(), rtol=2e-5)


def test_engine_graph_replay_equals_eager(golden):
    """RenderLoopEngine.forward_backward_graph (the evaluation captured once into a hipGraph and replayed) against the eager
    evaluation: bit-identical losses and camera gradients at every replay, also after the parameters were updated in place;
    and a whole adam loop with engine_graph=True ranks exactly like the eager loop."""
    from latentfusion_amd import synth
    from latentfusion_amd.engine import camera_params
    from latentfusion_amd.experimental import RenderLoopEngineX as RenderLoopEngine
    from latentfusion_amd.pose import estimation
    model, _ = synth.build_model(32, 16, 'pool:mean', seed=4, device=DEV, bias_std=0.05)
    model.freeze()
    g = golden('g7_adam_trace')
    target = _target(g)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    gen = torch.Generator().manual_seed(9)
    z_obj = torch.randn(1, 1, 16, 32, 32, 32, generator=gen).to(DEV)
    cam0 = prod_camera(g['init']).zoom(None, model.input_size, model.camera_dist)
    eng = RenderLoopEngine(model.photographer, z_obj, target, weights)
    P = camera_params(cam0).detach().clone().contiguous()
    cam = cam0._like(log_quaternion=P[:, 0:3], translation=P[:, 3:6], viewport=P[:, 6:10])
    for it in range(4):
        le, ge = eng.forward_backward(cam, params=P)
        lg, gg = eng.forward_backward_graph(cam, P)
        torch.cuda.synchronize()
        assert torch.equal(lg, le) and torch.equal(gg, ge), it
        P.sub_(0.01 * ge)                                          # in-place update, as the optimiser does
    runs = []
    for graph in (False, True):
        est = estimation.GradientPoseEstimator(model=model, learning_rate=0.01, num_samples=8, num_iters=8, ranking_size=8,
                                               converge_threshold=1e-9, converge_patience=100, optimizer='adam',
                                               loss_weights=weights, engine_graph=graph, return_camera_history=True)
        best, hist = est.estimate(z_obj, target, camera=prod_camera(g['init']))
        runs.append((torch.cat((best.log_quaternion, best.translation), dim=1).cpu(), torch.stack([h[0] for h in hist])))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize('crop', [16, 33])
def test_module_path_pose_loss_on_fused_kernels(golden, crop):
    """default_pose_loss on device tensors (the module path: GradientPoseEstimator(use_engine=False), custom estimators) runs
    lf_pose_loss_fwd_depth / _bwd_depth (pose/loss.py:_PoseLossTerms) -- no ATen reduction over the frame -- and equals the
    reference's expressions (pinned by G6 on the CPU): the four terms, and the gradients of an arbitrary weighting w.r.t. the
    depth crop, the mask-logit crop and the camera's viewport."""
    from latentfusion_amd import ops
    from latentfusion_amd.pose.loss import _pose_loss_expressions, default_pose_loss
    g6 = golden('g6_loss')
    target = _target(g6)
    target.depth[:, :, 100:140, 200:260] = 0.0
    gen = torch.Generator().manual_seed(crop + 1)
    depth0 = (torch.rand(3, 1, crop, crop, generator=gen) * 0.6 + 0.7).to(DEV)
    logit0 = (torch.randn(3, 1, crop, crop, generator=gen) * 2).to(DEV)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.7, 'mask': 0.5}
    res = []
    for fn in (_pose_loss_expressions, default_pose_loss):
        cam = prod_camera(g6['cam'])
        cam.viewport = cam.viewport.clone().requires_grad_(True)
        d, l = depth0.clone().requires_grad_(True), logit0.clone().requires_grad_(True)
        ops.KERNEL_TIMER = []
        try:
            ld = fn(target, d, l, cam)
            tags = {str(n) for n, _, _ in ops.KERNEL_TIMER}
        finally:
            ops.KERNEL_TIMER = None
        (sum(weights[k] * v for k, v in ld.items()) * torch.tensor([1.0, 0.5, 2.0], device=DEV)).sum().backward()
        res.append((ld, d.grad.clone(), l.grad.clone(), cam.viewport.grad.clone(), tags))
    assert 'pose_loss_terms' in res[1][4] and 'pose_loss_terms' not in res[0][4]
    for k in ('depth', 'ov_depth', 'iou', 'mask'):
        close(res[1][0][k], res[0][0][k], atol=2e-6, rtol=2e-5)
    close(res[1][1], res[0][1], atol=1e-8, rtol=2e-3)
    close(res[1][2], res[0][2], atol=1e-8, rtol=2e-3)
    close(res[1][3], res[0][3], atol=1e-6, rtol=5e-3)


def test_gradient_estimator_module_path_uses_the_fused_loss(golden):
    """GradientPoseEstimator(use_engine=False): the loss of every iteration comes from the fused kernels (tag 'pose_loss_terms'),
    and the 10-iteration reference trace (g7) still holds on that path."""
    from latentfusion_amd import ops
    from latentfusion_amd.pose import estimation
    import copy
    from latentfusion_amd.recon import fusion
    from latentfusion_amd.recon.inference import LatentFusionModel
    from latentfusion_amd.recon.models import Photographer, Sculptor
    g = golden('g7_adam_trace')
    model = LatentFusionModel(Sculptor.from_checkpoint(g['sculptor']), fusion.from_checkpoint(g['fuser']),
                              Photographer.from_checkpoint(g['photographer']), g['camera_dist'], DEV)
    target = _target(g, 'cpu')
    z_obj = g['z_obj'].to(DEV)
    est = estimation.load_from_config(copy.deepcopy(g['cfg']), model, track_stats=True, use_engine=False)
    ops.KERNEL_TIMER, ops.KERNEL_TIMER_TAGS = [], {'pose_loss_terms'}
    try:
        _, stats = est.estimate(z_obj, target, camera=prod_camera(g['init'], 'cpu'))
        n_loss = len(ops.KERNEL_TIMER)
    finally:
        ops.KERNEL_TIMER, ops.KERNEL_TIMER_TAGS = None, None
    assert n_loss == g['rank_loss'].shape[0], n_loss
    close(stats['rank_loss'][:3], g['rank_loss'][:3], atol=1e-5, rtol=1e-4)
    assert torch.equal(torch.argmin(stats['rank_loss'], dim=1), g['argmin'])


def test_engine_with_object_blocks_matches_module_path(golden):
    """a Photographer WITH object-frame blocks (reference recon/models.py:410-415) -- the engine evaluates them once per
    object at construction and renders from the cached result; losses and camera gradients equal Photographer.decode +
    default_pose_loss through the modules, and a host-resident target is accepted (the engine moves its buffers)."""
    from latentfusion_amd.engine import RenderLoopEngine
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon.models import Photographer
    torch.manual_seed(11)
    ph = Photographer(in_size=16, image_config=[[16, 32], [32, 16]], camera_config=[16, 16], object_config=[16, 16],
                      projection_type='factor', predict_color=False, predict_depth=True, predict_mask=True, scale_mode='nearest',
                      cube_size=1.0).to(DEV)
    with torch.no_grad():
        for n_, p in ph.named_parameters():
            if n_.endswith('bias'):
                p.normal_(0, 0.1)
    for p in ph.parameters():
        p.requires_grad_(False)
    assert len(ph.object_blocks) == 1
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    g = golden('g7_adam_trace')
    target = _target(g)
    z_obj = torch.randn(1, 1, 16, 16, 16, 16, generator=torch.Generator().manual_seed(12)).to(DEV)

    class _M:
        photographer, device, input_size, camera_dist = ph, torch.device(DEV), 16, g['camera_dist']

        @staticmethod
        def render_latent_object(z, cam, return_latent=True, apply_mask=True):
            y, zl, _ = ph.decode(z, cam, return_latent=return_latent, apply_mask=apply_mask)
            return y, (zl.squeeze(0) if return_latent else zl)
    cam0 = prod_camera(g['init']).zoom(None, 16, g['camera_dist'])
    est = estimation.GradientPoseEstimator(model=_M, learning_rate=0.01, num_samples=len(cam0), num_iters=1, ranking_size=len(cam0),
                                           converge_threshold=1e-6, converge_patience=10, optimizer='adam', loss_weights=weights,
                                           use_engine=False)
    st = est.start(z_obj, target, cam0)
    ld, _, rank, _ = est.loss_and_grad(z_obj, target, st['cam'])
    want_g = torch.cat((st['cam'].log_quaternion.grad, st['cam'].translation.grad, st['cam'].viewport.grad), dim=1)
    assert RenderLoopEngine.supports(ph, weights)
    for tgt in (target, _target(g, 'cpu')):
        eng = RenderLoopEngine(ph, z_obj, tgt, weights)
        losses, gparams = eng.forward_backward(cam0)
        for i, k in enumerate(eng.LOSS_KEYS):
            close(losses[:, i], ld[k], atol=5e-6, rtol=5e-5)
        close(losses[:, 4], rank, atol=5e-6, rtol=5e-5)
        rel = ((gparams - want_g).norm(dim=1) / want_g.norm(dim=1)).max().item()
        assert rel < 2e-3, rel


def test_occlusion_seventeenth_channel_kernels():
    """lf_occ_input_fwd / _bwd and lf_occ_conv17_fwd / _bwd against the expressions they stand for (reference
    recon/models.py:381-384 over the U-Net's input block, modules/blocks.py:78-91, and the first 3x3x3 convolution,
    blocks.py:152-158): LeakyReLU(conv1x1(cat(z, depth coordinate)) * he + b) with its 17 outputs written as a 16-channel volume
    and a scalar volume, the 1 -> 16 convolution of the latter, and their data gradients; fp64 expressions as the yardstick.
    Ragged sizes (W not a multiple of 4, one-plane volumes)."""
    from latentfusion_amd import _lib, ops
    from latentfusion_amd.recon.utils import get_normalized_voxel_depth
    F = torch.nn.functional
    L = _lib.lib()
    gen = torch.Generator().manual_seed(5)
    s = torch.cuda.current_stream().cuda_stream
    for (n, D, H, W) in ((3, 12, 6, 10), (2, 1, 5, 7), (1, 4, 4, 64)):
        z = ops.cl(torch.randn(n, 16, D, H, W, generator=gen).to(DEV))
        w = torch.randn(17, 17, generator=gen).to(DEV) * 0.4
        b = torch.randn(17, generator=gen).to(DEV) * 0.2
        wi = torch.zeros(17, 20, device=DEV)
        wi[:, :17] = w
        bi = torch.zeros(20, device=DEV)
        bi[:17] = b
        ta, t16 = ops.empty_cl(z.shape, DEV), torch.empty(n, 1, D, H, W, device=DEV)
        _lib.check(L.lf_occ_input_fwd(z.data_ptr(), wi.data_ptr(), bi.data_ptr(), ta.data_ptr(), t16.data_ptr(), n, D, H * W, 0.2, s), 'fwd')
        x17 = torch.cat((z, get_normalized_voxel_depth(z)), dim=1).double().requires_grad_(True)
        t = F.leaky_relu(torch.einsum('oc,ncdhw->nodhw', w.double(), x17) + b.double().view(1, 17, 1, 1, 1), 0.2)
        close(ta, t[:, :16].float(), atol=2e-6, rtol=1e-5)
        close(t16, t[:, 16:].float(), atol=2e-6, rtol=1e-5)
        # 1 -> 16 convolution of the scalar volume and its data gradient
        w2 = torch.randn(16, 1, 3, 3, 3, generator=gen).to(DEV) * 0.3
        w27 = w2.reshape(16, 27).t().contiguous()
        pre = ops.empty_cl(z.shape, DEV)
        _lib.check(L.lf_occ_conv17_fwd(t16.data_ptr(), w27.data_ptr(), pre.data_ptr(), n, D, H, W, s), 'conv17 fwd')
        t16d = t16.double().requires_grad_(True)
        want_pre = F.conv3d(t16d, w2.double(), padding=1)
        close(pre, want_pre.float(), atol=5e-6, rtol=1e-5)
        g = ops.cl(torch.randn(n, 16, D, H, W, generator=gen).to(DEV))
        gp16 = torch.empty_like(t16)
        _lib.check(L.lf_occ_conv17_bwd(g.data_ptr(), t16.data_ptr(), w27.data_ptr(), gp16.data_ptr(), n, D, H, W, 0.2, s), 'conv17 bwd')
        gt16, = torch.autograd.grad(want_pre, t16d, g.double())
        want_gp16 = gt16 * torch.where(t16d > 0, 1.0, 0.2)
        close(gp16, want_gp16.float(), atol=1e-5, rtol=1e-5)
        # input block backward with and without the direct term of the scaling
        gta = ops.cl(torch.randn(n, 16, D, H, W, generator=gen).to(DEV))
        g_zs = ops.cl(torch.randn(n, 16, D, H, W, generator=gen).to(DEV))
        wocc = torch.rand(n, 1, D, H, W, generator=gen).to(DEV)
        gz = ops.empty_cl(z.shape, DEV)
        _lib.check(L.lf_occ_input_bwd(gta.data_ptr(), ta.data_ptr(), gp16.data_ptr(), wi.data_ptr(), g_zs.data_ptr(), wocc.data_ptr(),
                                      gz.data_ptr(), n * D * H * W, 0.2, None, None, 0, s), 'bwd')
        g17 = torch.cat((gta.double(), gt16), dim=1)
        want, = torch.autograd.grad(t, x17, g17)
        close(gz, (want[:, :16] + g_zs.double() * wocc.double()).float(), atol=1e-5, rtol=1e-5)
        gz2 = ops.empty_cl(z.shape, DEV)
        _lib.check(L.lf_occ_input_bwd(gta.data_ptr(), ta.data_ptr(), gp16.data_ptr(), wi.data_ptr(), None, None,
                                      gz2.data_ptr(), n * D * H * W, 0.2, None, None, 0, s), 'bwd')
        close(gz2, want[:, :16].float(), atol=1e-5, rtol=1e-5)
        # ... followed by the epilogue backward of the layer that produced z (z = PixelNorm(LeakyReLU(pre)))
        prez = torch.randn(n, 16, D, H, W, generator=gen).to(DEV).double().requires_grad_(True)
        a_ = F.leaky_relu(prez, 0.2)
        nrm = (a_.pow(2).mean(dim=1, keepdim=True) + 1e-8).sqrt()
        yz = a_ / nrm
        want3, = torch.autograd.grad(yz, prez, want[:, :16])
        yz32, nrm32 = ops.cl(yz.detach().float()), nrm.detach().float().reshape(-1).contiguous()
        gz3 = ops.empty_cl(z.shape, DEV)
        _lib.check(L.lf_occ_input_bwd(gta.data_ptr(), ta.data_ptr(), gp16.data_ptr(), wi.data_ptr(), None, None, gz3.data_ptr(),
                                      n * D * H * W, 0.2, yz32.data_ptr(), nrm32.data_ptr(), 3, s), 'bwd')
        close(gz3, want3.float(), atol=2e-5, rtol=2e-5)


def test_occlusion_tail_round6_kernels():
    """The round-6 kernels around the occlusion module against the launches they replace and fp64 expressions (reference
    recon/models.py:378-395,427-430): the factor projection that scales its operand (lf_conv1x1_fwd_scaled == lf_column_scale_fwd +
    lf_conv1x1_fwd, same bits), LF_EPI_DOT of lf_conv1x1_bwd_data, output block + depth softmax in one pass
    (lf_column_softmax_head_fwd), the output block's backward (lf_occ_head_bwd vs lf_conv1x1_bwd_data with K = 1), the
    weights' gradient with the projection's data gradient recomputed (lf_occ_weight_grad, lf_occ_weight_grad_softmax_bwd) and the
    input block's backward in that form (lf_occ_input_bwd_proj)."""
    from latentfusion_amd import _lib, ops
    F = torch.nn.functional
    L = _lib.lib()
    gen = torch.Generator().manual_seed(11)
    s = torch.cuda.current_stream().cuda_stream
    for (n, D, H, W) in ((2, 8, 8, 8), (1, 5, 4, 12), (3, 16, 16, 16)):
        P = H * W
        zc = ops.cl(torch.randn(n, 16, D, H, W, generator=gen).to(DEV))
        wocc = torch.softmax(torch.randn(n, 1, D, H, W, generator=gen), dim=2).to(DEV).contiguous()
        pw = (torch.randn(16, 16 * D, 1, 1, generator=gen) * 0.2).to(DEV)          # projection weight [cout][C * D] (channel-major K)
        pb = (torch.randn(16, generator=gen) * 0.1).to(DEV)
        phe = ops.he_constant(pw)
        ppack = ops.pack_conv1x1(pw.reshape(16, 16, D).permute(0, 2, 1).reshape(16, D * 16))
        ppack_t = ops.pack_conv1x1(pw.reshape(16, 16, D).permute(2, 1, 0).reshape(D * 16, 16))
        flags = _lib.LF_EPI_LRELU | _lib.LF_EPI_PIXELNORM
        # 1. scaled projection == scale pass + projection, bit for bit
        zs = ops.empty_cl(zc.shape, DEV)
        _lib.check(L.lf_column_scale_fwd(zc.data_ptr(), wocc.data_ptr(), zs.data_ptr(), n * D * P, 16, s), 'scale')
        zp_a, zp_b = ops.empty_cl((n, 16, H, W), DEV), ops.empty_cl((n, 16, H, W), DEV)
        na = ops._conv1x1_raw(zs, ppack, pb, n, P, 16, D, D * P * 16, P * 16, 16, zp_a, phe, flags)
        nb = ops._conv1x1_raw(zc, ppack, pb, n, P, 16, D, D * P * 16, P * 16, 16, zp_b, phe, flags, xscale=wocc)
        assert torch.equal(zp_a, zp_b) and torch.equal(na, nb)
        # 2. data gradient of the projection + the per-voxel sums g_zs . zc (LF_EPI_DOT), with and without the gradient volume
        gp = ops.cl(torch.randn(n, 16, H, W, generator=gen).to(DEV))
        g_ref = ops.empty_cl(zc.shape, DEV)
        ops._conv1x1_raw(gp, ppack_t, None, n, P, 16, 1, P * 16, 0, D * 16, g_ref, phe, 0, yaddr=(D * P * 16, 16, 16, P * 16))
        g_dot, gw = ops.empty_cl(zc.shape, DEV), torch.empty(n, 1, D, H, W, device=DEV)
        _lib.check(L.lf_conv1x1_bwd_data(gp.data_ptr(), ppack_t.data_ptr(), g_dot.data_ptr(), n, P, 16, D * 16, D * P * 16, 16, 16, P * 16,
                                         phe, zc.data_ptr(), gw.data_ptr(), _lib.LF_EPI_DOT, 0.2, None, s), 'dot')
        assert torch.equal(g_dot, g_ref)
        want_gw = (g_ref.double() * zc.double()).sum(1, keepdim=True)
        close(gw, want_gw.float(), atol=2e-5, rtol=1e-5)
        gw2 = torch.empty_like(gw)
        _lib.check(L.lf_conv1x1_bwd_data(gp.data_ptr(), ppack_t.data_ptr(), None, n, P, 16, D * 16, D * P * 16, 16, 16, P * 16,
                                         phe, zc.data_ptr(), gw2.data_ptr(), _lib.LF_EPI_DOT, 0.2, None, s), 'dot only')
        assert torch.equal(gw2, gw)
        # 3. output block (16 -> 1, no activation) + softmax over the depth column
        hw = (torch.randn(16, generator=gen) * 0.5).to(DEV).contiguous()
        hb = (torch.randn(1, generator=gen) * 0.3).to(DEV)
        hhe = 0.35
        w_out = torch.empty(n, 1, D, H, W, device=DEV)
        _lib.check(L.lf_column_softmax_head_fwd(zc.data_ptr(), hw.data_ptr(), hb.data_ptr(), hhe, w_out.data_ptr(), None, n, D, P, s), 'head softmax')
        logits = (zc.double() * hw.double().view(1, 16, 1, 1, 1)).sum(1, keepdim=True) * hhe + hb.double()
        close(w_out, torch.softmax(logits, dim=2).float(), atol=2e-6, rtol=1e-5)
        # 4. output block backward with the producer's epilogue backward vs the K = 1 pointwise data gradient
        nrm = (torch.rand(n * D * P, generator=gen) + 0.5).to(DEV)
        gl = torch.randn(n, 1, D, H, W, generator=gen).to(DEV)
        hpkt = ops.pack_conv1x1(hw.reshape(16, 1))
        g_a, g_b = ops.empty_cl(zc.shape, DEV), ops.empty_cl(zc.shape, DEV)
        _lib.check(L.lf_conv1x1_bwd_data(gl.data_ptr(), hpkt.data_ptr(), g_a.data_ptr(), n, D * P, 1, 16, D * P * 16, 16, 16, 0, hhe,
                                         zc.data_ptr(), nrm.data_ptr(), flags, 0.2, None, s), 'head bwd (pointwise)')
        _lib.check(L.lf_occ_head_bwd(gl.data_ptr(), hw.data_ptr(), hhe, zc.data_ptr(), nrm.data_ptr(), flags, 0.2, g_b.data_ptr(), n * D * P, s),
                   'head bwd')
        close(g_b, g_a, atol=1e-6, rtol=1e-5)                       # (the same expression; the compiler contracts it differently: last-bit differences)
        if P % 16:
            continue                                              # (the matrix-pipe forms take whole groups of 16 voxels per depth plane)
        # 5. the weights' gradient with g_zs recomputed; followed by the softmax backward
        gw3 = torch.empty_like(gw)
        _lib.check(L.lf_occ_weight_grad(zc.data_ptr(), gp.data_ptr(), ppack_t.data_ptr(), phe, gw3.data_ptr(), n, D, P, s), 'weight grad')
        close(gw3, want_gw.float(), atol=2e-5, rtol=1e-5)
        glz = torch.empty_like(gw)
        _lib.check(L.lf_occ_weight_grad_softmax_bwd(zc.data_ptr(), gp.data_ptr(), ppack_t.data_ptr(), phe, wocc.data_ptr(), glz.data_ptr(), n, D, P, s),
                   'weight grad + softmax bwd')
        wd = wocc.double()
        want_gl = wd * (want_gw - (wd * want_gw).sum(2, keepdim=True))
        close(glz, want_gl.float(), atol=2e-5, rtol=2e-5)
        # 6. input block backward with the direct term recomputed == the form that reads g_zs (ta = NULL in both)
        wi = torch.zeros(17, 20, device=DEV)
        wi[:, :17] = torch.randn(17, 17, generator=gen).to(DEV) * 0.4
        gta = ops.cl(torch.randn(n, 16, D, H, W, generator=gen).to(DEV))
        gp16 = torch.randn(n, 1, D, H, W, generator=gen).to(DEV)
        for prev in (None, (zc, nrm, flags)):
            gz_a, gz_b = ops.empty_cl(zc.shape, DEV), ops.empty_cl(zc.shape, DEV)
            py, pn, pf = (prev[0].data_ptr(), prev[1].data_ptr(), prev[2]) if prev else (None, None, 0)
            _lib.check(L.lf_occ_input_bwd(gta.data_ptr(), None, gp16.data_ptr(), wi.data_ptr(), g_ref.data_ptr(), wocc.data_ptr(), gz_a.data_ptr(),
                                          n * D * P, 0.2, py, pn, pf, s), 'input bwd')
            _lib.check(L.lf_occ_input_bwd_proj(gta.data_ptr(), gp16.data_ptr(), wi.data_ptr(), gp.data_ptr(), ppack_t.data_ptr(), phe, wocc.data_ptr(),
                                               gz_b.data_ptr(), n, D, P, 0.2, py, pn, pf, s), 'input bwd proj')
            close(gz_b, gz_a, atol=3e-5, rtol=2e-5)


@pytest.mark.parametrize('projection', ['factor', 'sum'])
def test_engine_occlusion_module_on_explicit_kernels(golden, projection):
    """The 16-channel occlusion renderer (UNet3d(17, 1, [[17, 16], [16, 16]]), reference recon/models.py:305-306,378-395,427-430)
    sequenced by the engine on explicit kernels -- input block without the 17-channel concatenation, the 17 -> 16 convolution as
    two 16-channel Winograd launches, every data gradient with the previous layer's epilogue backward folded in -- against (a) the
    same engine with the occlusion module run as modules under autograd and (b) the estimator's module path."""
    from latentfusion_amd.engine import RenderLoopEngine
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon.models import Photographer
    g = golden('g7_adam_trace')
    target = _target(g)
    S = 32
    torch.manual_seed(11)
    ph = Photographer(in_size=S, camera_config=[16, 16], object_config=[16, 16], occlusion_config=[[17, 16], [16, 16]],
                      image_config=[[16, 32], [32, 16]], projection_type=projection, predict_color=False, predict_depth=True,
                      predict_mask=True, scale_mode='nearest', cube_size=1.0).to(DEV)
    with torch.no_grad():
        for name, p in ph.named_parameters():
            if name.endswith('bias'):
                p.normal_(0.0, 0.1)
    for p in ph.parameters():
        p.requires_grad_(False)
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.2, 'mask': 0.4}
    z_obj = torch.randn(1, 1, 16, S, S, S, generator=torch.Generator().manual_seed(3)).to(DEV)
    cam0 = prod_camera(g['init']).zoom(None, S, 1.0)
    eng = RenderLoopEngine(ph, z_obj, target, weights)
    assert eng.occ is not None and eng.generic_tail and (eng.dec is not None) == (projection == 'factor')
    losses, gparams = eng.forward_backward(cam0)
    again = eng.forward_backward(cam0)
    assert torch.equal(again[0], losses) and torch.equal(again[1], gparams)
    RenderLoopEngine.EXPLICIT_OCCLUSION = False
    try:
        ref_eng = RenderLoopEngine(ph, z_obj, target, weights)
    finally:
        RenderLoopEngine.EXPLICIT_OCCLUSION = True
    assert ref_eng.occ is None
    want_l, want_g = ref_eng.forward_backward(cam0)
    close(losses[:, :5], want_l[:, :5], atol=5e-6, rtol=5e-5)
    rel = ((gparams - want_g).norm(dim=1) / want_g.norm(dim=1)).max().item()
    assert rel < 1e-3, rel

    class _M:
        photographer, device, input_size, camera_dist = ph, torch.device(DEV), S, 1.0

        @staticmethod
        def render_latent_object(z, cam, return_latent=True, apply_mask=True):
            y, zl, _ = ph.decode(z, cam, return_latent=return_latent, apply_mask=apply_mask)
            return y, (zl.squeeze(0) if return_latent else zl)
    est = estimation.GradientPoseEstimator(model=_M, learning_rate=0.01, num_samples=len(cam0), num_iters=1, ranking_size=len(cam0),
                                           converge_threshold=1e-6, converge_patience=10, optimizer='adam', loss_weights=weights,
                                           use_engine=False)
    st = est.start(z_obj, target, cam0.to(DEV))
    ld, _, rank, _ = est.loss_and_grad(z_obj, target, st['cam'])
    mod_g = torch.cat((st['cam'].log_quaternion.grad, st['cam'].translation.grad, st['cam'].viewport.grad), dim=1)
    close(losses[:, 4], rank, atol=5e-6, rtol=5e-5)
    rel = ((gparams - mod_g).norm(dim=1) / mod_g.norm(dim=1)).max().item()
    assert rel < 2e-3, rel
    fwd_only, none = eng.forward_backward(cam0, need_grad=False)
    assert none is None and torch.equal(fwd_only, losses)


@pytest.mark.parametrize('proj', ['factor', 'sum'])
def test_g28_occlusion16_loop_on_hip(golden, proj):
    """The explicit-kernel occlusion renderer against the REFERENCE ITSELF (golden g28: the reference's Photographer with
    UNet3d(17, 1, [[17, 16], [16, 16]]) under its GradientPoseEstimator, 6 hypotheses x 6 iterations of adam_quick at 32^3 x 16):
    iteration-0 logits, the rank-loss trace, identical argmin at every iteration."""
    from latentfusion_amd.engine import RenderLoopEngine
    from latentfusion_amd.pose import estimation
    from latentfusion_amd.recon.models import Photographer
    g = golden('g28_occlusion16')
    r = g['variants'][proj]
    ph = Photographer.from_checkpoint(r['photographer']).to(DEV)
    for p in ph.parameters():
        p.requires_grad_(False)
    S = g['S']

    class _M:
        photographer, device, input_size, camera_dist = ph, torch.device(DEV), S, g['camera_dist']

        @staticmethod
        def render_latent_object(z, cam, return_latent=True, apply_mask=True):
            y, zl, _ = ph.decode(z, cam, return_latent=return_latent, apply_mask=apply_mask)
            return y, (zl.squeeze(0) if return_latent else zl)
    z_obj = g['z_obj'].to(DEV)
    target = _target(g, 'cpu')
    init = prod_camera(g['init'], 'cpu')
    with torch.no_grad():
        y0, _ = _M.render_latent_object(z_obj, init.zoom(None, S, g['camera_dist']).to(DEV))
    for k in ('depth_logits', 'mask_logits'):
        close(y0[k].squeeze(0), r['iter0'][k], atol=1e-4, rtol=1e-3)
    eng = RenderLoopEngine(ph, z_obj, target.to(DEV), dict(g['cfg']['loss_weights']))
    assert eng.occ is not None and (eng.dec is not None) == (proj == 'factor')
    est = estimation.load_from_config(copy.deepcopy(g['cfg']), _M, track_stats=True, return_camera_history=True)
    best, stats, hist = est.estimate(z_obj, target, camera=init)
    # (same inputs at iteration 0; afterwards Adam's normalised steps amplify fp32 rounding, as in g7 / g26)
    close(stats['rank_loss'][:1], r['rank_loss'][:1], atol=1e-6, rtol=2e-5)
    close(stats['rank_loss'][:3], r['rank_loss'][:3], atol=1e-5, rtol=1e-3)
    close(stats['rank_loss'], r['rank_loss'], atol=1e-4, rtol=1e-2)
    assert torch.argmin(stats['rank_loss'], dim=1).tolist() == r['argmin'].tolist()
    close(torch.stack([c.log_quaternion for _, c in hist]), r['hist_log_q'], atol=2e-2, rtol=0)
