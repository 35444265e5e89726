"""Generates tests/golden/*.pt from the REAL reference (build container only).

    python oracle/make_golden.py            # rewrites every fixture

Each fixture is plain data: inputs, reference-format checkpoints (args + state_dict) and the
outputs / gradients the reference produced.  Tests then check (a) the oracle restatement and
(b) the HIP product path against them, so the GPU box never needs /root/reference.
Fixture ids follow SURVEY.md section 8c (G1..G10).
"""
import copy
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refharness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
INTRINSIC = [[615.1436, 0.0, 315.3623, 0.0], [0.0, 615.4991, 251.5415, 0.0], [0.0, 0.0, 1.0, 0.0]]


def save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.pt')
    torch.save(obj, path)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def cam_dict(c):
    return {'K': c.intrinsic.detach().clone(), 'viewport': c.viewport.detach().clone(),
            'log_q': c.log_quaternion.detach().clone(), 't': c.translation.detach().clone(),
            'z_span': c.z_span, 'width': c.width, 'height': c.height}


def rand_cameras(lf, n, zoomed_size=None, dist=None, seed=0):
    from latentfusion.modules.geometry import Camera
    g = torch.Generator().manual_seed(seed)
    log_q = torch.randn(n, 3, generator=g) * 0.7
    t = torch.cat((torch.randn(n, 2, generator=g) * 0.05, 1.0 + 0.2 * torch.rand(n, 1, generator=g)), 1)
    K = torch.tensor(INTRINSIC).unsqueeze(0).expand(n, -1, -1).clone()
    cam = Camera(K, None, log_quaternion=log_q, translation=t)
    if zoomed_size is not None:
        cam = cam.zoom(None, zoomed_size, dist)
    return cam


def syn_ckpts(lf, S, C, fuser='gru', seed=0, **ph_kw):
    from latentfusion.recon.models import Sculptor, Photographer
    from latentfusion.recon import fusion
    torch.manual_seed(seed)
    img = [[16, 32], [32, 16]]
    sc = Sculptor(in_size=S, image_config=img, camera_config=[C, C], object_config=[C, C],
                  projection_type='factor', input_color=True, input_depth=False, input_mask=True,
                  scale_mode='nearest').eval()
    kw = dict(in_size=S, image_config=img, camera_config=[C, C], object_config=[], projection_type='factor',
              predict_depth=True, predict_mask=True, scale_mode='nearest')
    kw.update(ph_kw)
    ph = Photographer(**kw).eval()
    fu = fusion.get_fuser(fuser, C, 1.0).eval()
    # perturb the zero-initialised biases so that bias handling is actually pinned
    for m in (sc, ph, fu):
        for k, p in m.named_parameters():
            if k.endswith('bias'):
                p.data.normal_(0, 0.1)
    return sc, fu, ph


def ck(module):
    d = module.create_checkpoint()
    d = {k: (copy.deepcopy(v) if k != 'state_dict' else {n: t.clone() for n, t in v.items()})
         for k, v in d.items()}
    return d


def synth_obs(lf, V, seed):
    """SURVEY section 8d synthetic observation."""
    from latentfusion.modules.geometry import Camera
    from latentfusion.observation import Observation
    from latentfusion import three
    torch.manual_seed(seed)
    q = three.orientation.evenly_distributed_quats(V)
    t = torch.tensor([[0.0, 0.0, 1.0]]).expand(V, -1)
    E = three.to_extrinsic_matrix(t, q)
    K = torch.tensor(INTRINSIC).unsqueeze(0).expand(V, -1, -1).clone()
    cam = Camera(K, E)
    color = torch.rand(V, 3, 480, 640)
    yy, xx = torch.meshgrid(torch.arange(480.0), torch.arange(640.0), indexing='ij')
    disc = (((xx - 315) ** 2 + (yy - 251) ** 2) <= 150 ** 2).float()
    mask = disc.view(1, 1, 480, 640).expand(V, -1, -1, -1).clone()
    depth = (1.0 + 0.1 * torch.rand(V, 1, 480, 640)) * mask
    return Observation(color, depth, mask, cam)


def obs_dict(o, lite=False):
    """lite: drop the colour frame (unused by the pose loss) and store the mask as bool."""
    if lite:
        return {'color': None, 'depth': o.depth.clone(), 'mask': o.mask.bool(), 'cam': cam_dict(o.camera)}
    return {'color': o.color.clone(), 'depth': o.depth.clone(), 'mask': o.mask.clone(), 'cam': cam_dict(o.camera)}


# ---------------------------------------------------------------------------------------------
def g1_camera(lf):
    from latentfusion.three import quaternion as Q
    from latentfusion import three
    cam = rand_cameras(lf, 5, seed=1)
    cam.viewport = torch.tensor([[100., 80, 420, 400], [0, 0, 640, 480], [50.5, 20.25, 300, 270],
                                 [-30, -40, 500, 490], [200, 100, 456, 356]])
    g = torch.Generator().manual_seed(2)
    qa, qb = torch.randn(6, 4, generator=g), torch.randn(6, 4, generator=g)
    rot = Q.quat_to_mat(qa)
    depth = 0.5 + torch.rand(5, 1, 6, 7, generator=g)
    zcam = cam.zoom(None, 32, 2.5)
    small = torch.rand(5, 1, 16, 16, generator=g)
    un_n, _ = zcam.uncrop(small, scale_mode='nearest')
    un_b, _ = zcam.uncrop(small, scale_mode='bilinear')
    torch.manual_seed(7)
    edq = three.orientation.evenly_distributed_quats(8)
    torch.manual_seed(7)
    edq_hu = three.orientation.evenly_distributed_quats(6, hemisphere=True, upright=True)
    save('g1_camera', {
        'cam': cam_dict(cam), 'quaternion': cam.quaternion.clone(), 'R': cam.rotation_matrix.clone(),
        'obj_to_cam': cam.obj_to_cam.clone(), 'cam_to_obj': cam.cam_to_obj.clone(),
        'znear': cam.znear.clone(), 'zfar': cam.zfar.clone(), 'position': cam.position.clone(),
        'zoom_viewport': zcam.viewport.clone(), 'depth': depth,
        'depth_norm': cam.normalize_depth(depth), 'depth_denorm': cam.denormalize_depth(depth * 2 - 1),
        'qa': qa, 'qb': qb, 'qmul': Q.qmul(qa, qb), 'qa_mat': rot, 'mat_to_quat': Q.mat_to_quat(rot),
        'qlog': Q.qlog(qa), 'qexp3': Q.qexp(qa[:, 1:]), 'angdist': Q.angular_distance(qa, qb),
        'small': small, 'uncrop_nearest': un_n[:, :, ::4, ::4].clone(), 'uncrop_bilinear': un_b[:, :, ::4, ::4].clone(),
        'edq_seed': 7, 'edq8': edq, 'edq6_hemi_upright': edq_hu,
    })
    # full-resolution crop fixture separately (input image is large): use a low-res frame instead
    from latentfusion.modules.geometry import Camera
    cam2 = Camera(cam.intrinsic * torch.tensor([[[0.1], [0.1], [1.0]]]), None, log_quaternion=cam.log_quaternion,
                  translation=cam.translation, width=64, height=48)
    img2 = torch.rand(5, 2, 48, 64, generator=g)
    c_b, zc2 = cam2.zoom(img2, 16, 2.5, scale_mode='bilinear')
    c_n, _ = cam2.zoom(img2, 16, 2.5, scale_mode='nearest')
    save('g1_zoom', {'cam': cam_dict(cam2), 'img': img2, 'crop_bilinear': c_b, 'crop_nearest': c_n,
                     'zoom_viewport': zc2.viewport.clone(), 'target_size': 16, 'target_dist': 2.5})


def g2_resample(lf):
    from latentfusion.modules.geometry import CameraToObjectTransform, ObjectToCameraTransform
    B, C, S = 3, 3, 8
    cam = rand_cameras(lf, B, zoomed_size=S, dist=2.0, seed=3)
    g = torch.Generator().manual_seed(4)
    out = {'cam': cam_dict(cam), 'cube_size': 1.0}
    for name, T in (('o2c', ObjectToCameraTransform(1.0)), ('c2o', CameraToObjectTransform(1.0))):
        vol = torch.randn(B, C, S, S, S, generator=g, requires_grad=True)
        w = torch.randn(B, C, S, S, S, generator=g)
        # The reference's C2O divides pixel coords in place (geometry.py:636), so it is NOT
        # differentiable w.r.t. the camera; only O2C gets camera gradients.
        want_cam_grad = name == 'o2c'
        cam.log_quaternion = cam.log_quaternion.detach().requires_grad_(want_cam_grad)
        cam.translation = cam.translation.detach().requires_grad_(want_cam_grad)
        cam.viewport = cam.viewport.detach().requires_grad_(want_cam_grad)
        y = T(vol, cam)
        (y * w).sum().backward()
        out[name] = {'vol': vol.detach().clone(), 'w': w, 'out': y.detach().clone(), 'g_vol': vol.grad.clone()}
        if want_cam_grad:
            out[name].update({'g_log_q': cam.log_quaternion.grad.clone(), 'g_t': cam.translation.grad.clone(),
                              'g_viewport': cam.viewport.grad.clone()})
    save('g2_resample', out)


def g3_block(lf):
    from latentfusion.modules.blocks import Block
    from latentfusion.modules import EqualizedConv2d, EqualizedConv3d
    g = torch.Generator().manual_seed(5)
    out = {}
    for dims, conv in ((3, EqualizedConv3d), (2, EqualizedConv2d)):
        for mode, factor in (('nearest', 1.0), ('nearest', 2.0), ('bilinear', 0.5), ('bilinear', 2.0)):
            torch.manual_seed(6)
            m = mode
            if dims == 3 and mode == 'bilinear':
                m = 'trilinear'
            blk = Block(5, 7, conv_module=conv, scale_factor=factor, scale_mode=m).eval()
            for k, p in blk.named_parameters():
                if k.endswith('bias'):
                    p.data.normal_(0, 0.1)
            x = torch.randn(2, 5, *([8] * dims), generator=g, requires_grad=True)
            y = blk(x)
            w = torch.randn(y.shape, generator=g)
            (y * w).sum().backward()
            out[f'{dims}d_{mode}_{factor}'] = {
                'sd': {k: v.detach().clone() for k, v in blk.state_dict().items()}, 'x': x.detach().clone(),
                'w': w, 'y': y.detach().clone(), 'g_x': x.grad.clone(),
                'g_w1': blk.conv1.module.weight.grad.clone(), 'g_b1': blk.conv1.bias.grad.clone(),
                'scale': factor, 'mode': mode}
    save('g3_block', out)


def g4_fusers(lf):
    from latentfusion.recon import fusion
    from latentfusion.modules.geometry import Camera
    g = torch.Generator().manual_seed(8)
    V, C, S = 3, 4, 8
    z = torch.randn(1, V, C, S, S, S, generator=g)
    cam = rand_cameras(lf, V, zoomed_size=S, dist=2.0, seed=9)
    out = {'z': z, 'cam': cam_dict(cam)}
    for pool in ('mean', 'max', 'abs_max', 'median'):
        f = fusion.get_fuser('pool:' + pool, C, 1.0)
        out['pool_' + pool] = f(z, None, None, cam)[0].clone()
    out['concat'] = fusion.get_fuser('concat', C, 1.0)(z, None, None, cam)[0].clone()
    for kind in ('gru', 'lstm'):
        torch.manual_seed(10)
        f = fusion.get_fuser(kind, C, 1.0).eval()
        for k, p in f.named_parameters():
            if k.endswith('bias'):
                p.data.normal_(0, 0.1)
        with torch.no_grad():
            out[kind] = {'ck': ck(f), 'out': f(z, None, None, cam)[0].clone()}
    torch.manual_seed(11)
    f = fusion.get_fuser('blend', C, 1.0, block_config=[[5, 8], [8, 4]]).eval()
    zmid = torch.randn(1, V, C, S, S, S, generator=g)
    with torch.no_grad():
        out['blend'] = {'ck': ck(f), 'z_cam_mid': zmid, 'out': f(z, [zmid], None, cam)[0].clone()}
    save('g4_fusers', out)


def g5_decode(lf):
    S, C, N = 16, 8, 3
    out = {}
    variants = {'factor': {}, 'sum': {'projection_type': 'sum'},
                'occlusion': {'occlusion_config': [[9, 8], [8, 8]], 'object_config': [C, C]}}
    for name, kw in variants.items():
        _, _, ph = syn_ckpts(lf, S, C, seed=12, **kw)
        if name == 'sum':
            # 'sum' leaves C channels for the decoder; image_config[0][0] must equal C
            from latentfusion.recon.models import Photographer
            torch.manual_seed(12)
            ph = Photographer(in_size=S, image_config=[[C, 16], [16, 8]], camera_config=[C, C], object_config=[],
                              projection_type='sum', scale_mode='nearest').eval()
        dist = lf.recon.utils.optimal_camera_dist(615.4991, S, 0.5, slack=128 / S)
        cam = rand_cameras(lf, N, zoomed_size=S, dist=dist, seed=13)
        cam.log_quaternion = cam.log_quaternion.detach().requires_grad_(True)
        cam.translation = cam.translation.detach().requires_grad_(True)
        cam.viewport = cam.viewport.detach().requires_grad_(True)
        g = torch.Generator().manual_seed(14)
        z_obj = torch.randn(1, 1, C, S, S, S, generator=g)
        y, lat, zd = ph.decode(z_obj, cam, return_latent=True, apply_mask=True)
        wd = torch.randn(y['depth_logits'].shape, generator=g)
        wm = torch.randn(y['mask_logits'].shape, generator=g)
        ((y['depth_logits'] * wd).sum() + (y['mask_logits'] * wm).sum()).backward()
        out[name] = {'ck': ck(ph), 'cam': cam_dict(cam), 'z_obj': z_obj, 'wd': wd, 'wm': wm,
                     'y': {k: v.detach().clone() for k, v in y.items()}, 'latent': lat.detach().clone(),
                     'z_depth': None if zd is None else zd.detach().clone(),
                     'g_log_q': cam.log_quaternion.grad.clone(), 'g_t': cam.translation.grad.clone(),
                     'g_viewport': cam.viewport.grad.clone()}
    save('g5_decode', out)


def g6_loss(lf):
    from latentfusion.pose.estimation import default_pose_loss
    from latentfusion.observation import Observation
    N, S = 3, 16
    target = synth_obs(lf, 1, seed=15)
    # make some invalid pixels: mask>0.1 but depth==0
    target.depth[:, :, 240:260, 300:340] = 0.0
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S, 0.5, slack=128 / S)
    cam = rand_cameras(lf, N, zoomed_size=S, dist=dist, seed=16)
    cam.viewport = (cam.viewport * torch.tensor([[1.0, 1.0, 0.6, 0.6]]) + torch.tensor([[150.0, 100, 0, 0]])) \
        .detach().requires_grad_(True)
    cam.translation = cam.translation.detach().requires_grad_(True)
    g = torch.Generator().manual_seed(17)
    depth_n = (torch.rand(N, 1, S, S, generator=g) * 2 - 1).requires_grad_(True)
    logits = (torch.randn(N, 1, S, S, generator=g) * 3).requires_grad_(True)
    z_depth = cam.denormalize_depth(depth_n)
    ld = default_pose_loss(target, z_depth, logits, cam)
    wts = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.7, 'mask': 0.5}
    total = sum(wts[k] * v for k, v in ld.items())
    total.mean().backward()
    save('g6_loss', {'target': obs_dict(target, lite=True), 'cam': cam_dict(cam), 'depth_n': depth_n.detach().clone(),
                     'logits': logits.detach().clone(), 'weights': wts,
                     'loss': {k: v.detach().clone() for k, v in ld.items()},
                     'g_depth_n': depth_n.grad.clone(), 'g_logits': logits.grad.clone(),
                     'g_viewport': cam.viewport.grad.clone(), 'g_t': cam.translation.grad.clone()})


def _model(lf, S, C, fuser, seed):
    from latentfusion.recon.inference import LatentFusionModel
    sc, fu, ph = syn_ckpts(lf, S, C, fuser=fuser, seed=seed)
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S, 0.5, slack=128 / S)
    return LatentFusionModel(sc, fu, ph, dist, 'cpu'), (ck(sc), ck(fu), ck(ph)), dist


def g7_g10_loop(lf):
    from latentfusion.pose import estimation, utils as pu
    import tomli
    S, C, V, N = 16, 8, 4, 8
    for fuser in ('gru', 'pool:mean'):
        tag = fuser.replace(':', '_')
        model, cks, dist = _model(lf, S, C, fuser, seed=20)
        ref_obs = synth_obs(lf, V, seed=21)
        target = synth_obs(lf, 1, seed=22)
        z_obj = model.build_latent_object(ref_obs)
        pre = model.preprocess_observation(ref_obs)
        # G10: encode
        save(f'g10_encode_{tag}', {'sculptor': cks[0], 'fuser': cks[1], 'camera_dist': dist,
                                   'obs_pre': {'color': pre.color.clone(), 'depth': pre.depth.clone(),
                                               'mask': pre.mask.clone(), 'cam': cam_dict(pre.camera)},
                                   'z_obj': z_obj.clone()})
        if fuser != 'gru':
            continue
        # raw->preprocessed pin for Observation.zoom/prepare/normalize (small crop of inputs is enough)
        with open('/root/reference/configs/adam_quick.toml', 'rb') as f:
            cfg = tomli.load(f)
        cfg['args']['num_iters'] = 10
        torch.manual_seed(23)
        init = pu.sample_cameras_with_estimate(N, target.camera)
        est = estimation.load_from_config(copy.deepcopy(cfg), model, track_stats=True, return_camera_history=True)
        best, stats, hist = est.estimate(z_obj, target, camera=init)
        save('g7_adam_trace', {
            'sculptor': cks[0], 'fuser': cks[1], 'photographer': cks[2], 'camera_dist': dist, 'cfg': cfg,
            'z_obj': z_obj.clone(), 'target': obs_dict(target, lite=True), 'init': cam_dict(init), 'init_seed': 23,
            'rank_loss': stats['rank_loss'].clone(), 'depth_loss': stats['depth_loss'].clone(),
            'ov_depth_loss': stats['ov_depth_loss'].clone(), 'iou_loss': stats['iou_loss'].clone(),
            'mask_loss': stats['mask_loss'].clone(),
            'argmin': torch.argmin(stats['rank_loss'], dim=1),
            'hist_log_q': torch.stack([c.log_quaternion for _, c in hist]),
            'hist_t': torch.stack([c.translation for _, c in hist]),
            'best': cam_dict(best)})
        # G8: one cross-entropy refine step with injected sample cameras
        torch.manual_seed(24)
        cams = pu.sample_cameras_with_estimate(6, target.camera, hemisphere=True, upright=True)
        ce = estimation.CrossEntropyPoseEstimator(
            model=model, num_samples=24, num_elites=5, num_iters=3, num_gmm_components=2, learning_rate=0.9,
            sample_flipped=True, ranking_size=4, loss_weights={'depth': 1.0, 'ov_depth': 0.2, 'iou': 0.1, 'mask': 0.3})
        from latentfusion.modules.geometry import Camera
        allc = Camera.cat([cams, pu.flip_camera(cams, axis=(0.0, 0.0, 1.0)), pu.flip_camera(cams, axis=(0.0, 1.0, 0.0)),
                           pu.flip_camera(cams, axis=(1.0, 0.0, 0.0))])
        zd, zl, _, zc = ce._render_observation(z_obj, allc)
        ld = ce.loss_func(target, zd, zl, zc)
        loss = sum(estimation.weigh_losses(ld, ce.loss_weights).values())
        save('g8_ce_step', {'cams': cam_dict(cams), 'all_cams': cam_dict(allc), 'zoom_viewport': zc.viewport.clone(),
                            'loss': loss.clone(), 'order': torch.argsort(loss), 'weights': dict(ce.loss_weights),
                            'depth_crop': zd.clone(), 'mask_logits_crop': zl.clone()})


def g9_ibr(lf):
    from latentfusion import ibr
    S, C = 16, 8
    _, _, ph = syn_ckpts(lf, S, C, seed=30)
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S, 0.5, slack=128 / S)
    cam_in = rand_cameras(lf, 3, zoomed_size=S, dist=dist, seed=31)
    cam_out = rand_cameras(lf, 2, zoomed_size=S, dist=dist, seed=32)
    g = torch.Generator().manual_seed(33)
    z_obj = torch.randn(1, 1, C, S, S, S, generator=g)
    img = torch.rand(1, 3, 3, S, S, generator=g)
    with torch.no_grad():
        y, lat = ibr.render_latent_ibr2(ph, z_obj, cam_in, cam_out, img, p=0.5, apply_mask=True)
        y_in, _, _ = ph.decode(z_obj, cam_in, apply_mask=True)
        y_out, _, _ = ph.decode(z_obj, cam_out, apply_mask=True)
        reproj, dreproj = ibr.reproject_views(img[0], y_in['depth'][0], y_out['depth'][0], cam_in, cam_out)
    save('g9_ibr', {'ck': ck(ph), 'cam_in': cam_dict(cam_in), 'cam_out': cam_dict(cam_out), 'z_obj': z_obj,
                    'image_in': img, 'color': y['color'].clone(), 'depth': y['depth'].clone(),
                    'image_reproj': reproj.clone(), 'depth_reproj': dreproj.clone()})


def g0_preprocess(lf):
    """Observation.zoom/prepare/normalize on a low-resolution frame (observation.py:225-273)."""
    from latentfusion.modules.geometry import Camera
    from latentfusion.observation import Observation
    g = torch.Generator().manual_seed(40)
    V = 2
    K = torch.tensor(INTRINSIC).unsqueeze(0).expand(V, -1, -1).clone()
    K[:, :2] *= 0.125
    cam = Camera(K, None, log_quaternion=torch.randn(V, 3, generator=g) * 0.5,
                 translation=torch.tensor([[0.02, -0.01, 1.0], [0.0, 0.03, 1.1]]), width=80, height=60)
    color = torch.rand(V, 3, 60, 80, generator=g)
    mask = (torch.rand(V, 1, 60, 80, generator=g) > 0.4).float()
    depth = (0.9 + 0.3 * torch.rand(V, 1, 60, 80, generator=g))
    obs = Observation(color, depth, mask, cam)
    z = obs.zoom(2.0, 16)
    p = z.prepare()
    n = p.normalize()
    save('g0_preprocess', {'obs': obs_dict(obs), 'target_dist': 2.0, 'target_size': 16,
                           'zoom': obs_dict(z), 'prepare': obs_dict(p), 'normalize': obs_dict(n)})


def g11_released_like(lf):
    """A scaled-down copy of the RELEASED architecture's structure (tools/train/train.sh:28-66):
    U-Nets with D/U tokens, bilinear rescaling and skip connections, channel counts that are not
    multiples of 16, camera blocks that change width, tanh-free heads.  Pins the generic paths."""
    from latentfusion.recon.models import Sculptor, Photographer
    from latentfusion.recon import fusion
    from latentfusion.recon.inference import LatentFusionModel
    torch.manual_seed(50)
    S_in = 16
    sc = Sculptor(in_size=S_in, image_config=[[8, 'D', 24, 'D', 24], [24, 'U', 24, 'U', 12]],
                  camera_config=[4, 8, 20], object_config=[20, 20], projection_type='factor',
                  input_color=True, input_depth=False, input_mask=True, scale_mode='nearest').eval()
    S = sc.out_size
    ph = Photographer(in_size=S, image_config=[[12, 'D', 24, 'D', 24], [24, 'U', 24, 'U', 12, 'U', 8]],
                      camera_config=[20, 20], object_config=[], projection_type='factor',
                      predict_depth=True, predict_mask=True, scale_mode='nearest').eval()
    fu = fusion.get_fuser('gru', 20, 1.0).eval()
    for m in (sc, ph, fu):
        for k, p in m.named_parameters():
            if k.endswith('bias'):
                p.data.normal_(0, 0.1)
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S_in, 0.5, slack=0.5)
    model = LatentFusionModel(sc, fu, ph, dist, 'cpu')
    ref_obs = synth_obs(lf, 3 + 1, seed=51)
    pre = model.preprocess_observation(ref_obs)
    z_obj = model.build_latent_object(ref_obs)
    cam = rand_cameras(lf, 2, zoomed_size=S_in, dist=dist, seed=52)
    cam.log_quaternion = cam.log_quaternion.detach().requires_grad_(True)
    cam.translation = cam.translation.detach().requires_grad_(True)
    cam.viewport = cam.viewport.detach().requires_grad_(True)
    y, lat, _ = ph.decode(z_obj, cam, return_latent=True, apply_mask=True)
    g = torch.Generator().manual_seed(53)
    wd = torch.randn(y['depth_logits'].shape, generator=g)
    wm = torch.randn(y['mask_logits'].shape, generator=g)
    ((y['depth_logits'] * wd).sum() + (y['mask_logits'] * wm).sum()).backward()
    save('g11_released_like', {
        'sculptor': ck(sc), 'fuser': ck(fu), 'photographer': ck(ph), 'camera_dist': dist,
        'obs_pre': {'color': pre.color.clone(), 'depth': pre.depth.clone(), 'mask': pre.mask.clone(), 'cam': cam_dict(pre.camera)},
        'z_obj': z_obj.clone(), 'cam': cam_dict(cam), 'wd': wd, 'wm': wm,
        'y': {k: v.detach().clone() for k, v in y.items()}, 'latent': lat.detach().clone(),
        'g_log_q': cam.log_quaternion.grad.clone(), 'g_t': cam.translation.grad.clone(),
        'g_viewport': cam.viewport.grad.clone(), 'sizes': {'sculptor_out': S, 'photographer_out': ph.out_size}})


def g12_latent_code(lf):
    """compute_latent_code (inference.py:86-99 -> autoencode, models.py:73-81) and the latent term of
    default_pose_loss (estimation.py:111-116): the path behind adam_latent / cross_entropy_latent."""
    from latentfusion.pose.estimation import default_pose_loss
    model, cks, dist = _model(lf, 16, 8, 'gru', seed=60)
    ref_obs = synth_obs(lf, 4, seed=61)
    target = synth_obs(lf, 1, seed=62)
    target.color = torch.round(target.color * 255.0) / 255.0          # 8-bit colours: stored as uint8 below
    z_obj = model.build_latent_object(ref_obs)
    cams = rand_cameras(lf, 3, seed=63)
    zcams = cams.zoom(None, model.input_size, model.camera_dist)
    with torch.no_grad():
        z_target = model.compute_latent_code(target, zcams)
        pred, z_pred = model.render_latent_object(z_obj, zcams, return_latent=True)
        zd = zcams.denormalize_depth(pred['depth'].squeeze(0))
        ld = default_pose_loss(target, zd, pred['mask_logits'].squeeze(0), zcams, z_pred_latent=z_pred,
                               z_target_latent=z_target)
    save('g12_latent_code', {'sculptor': cks[0], 'fuser': cks[1], 'photographer': cks[2], 'camera_dist': dist,
                             'z_obj': z_obj.clone(),
                             'target': {'color_u8': torch.round(target.color * 255.0).to(torch.uint8), 'depth': target.depth.clone(),
                                        'mask': target.mask.bool(), 'cam': cam_dict(target.camera)},
                             'cams': cam_dict(zcams),
                             'z_target_latent': z_target.clone(), 'z_pred_latent': z_pred.clone(),
                             'latent_loss': ld['latent'].clone()})


def g13_metrics(lf):
    """Pose metrics (pose/metrics.py:19-109) on random model points and camera pairs."""
    from latentfusion.pose import metrics
    g = torch.Generator().manual_seed(70)
    points = (torch.rand(1500, 3, generator=g) - 0.5) * torch.tensor([0.6, 0.4, 0.3])
    gt = rand_cameras(lf, 3, seed=71)
    ev = rand_cameras(lf, 3, seed=72)
    ev.log_quaternion = gt.log_quaternion + 0.1 * torch.randn(3, 3, generator=g)
    ev.translation = gt.translation + 0.02 * torch.randn(3, 3, generator=g)
    out = []
    for i in range(3):
        m = metrics.camera_metrics(gt[i], ev[i], points, 0.35)
        out.append({k: float(v) for k, v in m.items()})
    save('g13_metrics', {'points': points, 'gt': cam_dict(gt), 'ev': cam_dict(ev), 'scale': 0.35, 'metrics': out})


def g14_initial_pose(lf):
    """estimate_initial_pose (pose/initialization.py:59-99).  skimage is absent from this image: its
    binary_erosion / disk (published algorithm: scipy.ndimage.binary_erosion with border_value=True over a
    Euclidean disk) are supplied from scipy for the duration of the call; the reference discards the eroded
    mask anyway (initialization.py:41-42)."""
    import numpy as np
    import scipy.ndimage as ndi
    from latentfusion.pose import initialization as ini

    class _Morph:
        @staticmethod
        def disk(r):
            L = np.arange(-r, r + 1)
            X, Y = np.meshgrid(L, L)
            return (X ** 2 + Y ** 2 <= r ** 2).astype(np.uint8)

        @staticmethod
        def binary_erosion(img, selem=None):
            return ndi.binary_erosion(img, structure=selem, border_value=True)
    old, ini.morphology = ini.morphology, _Morph
    try:
        obs = synth_obs(lf, 2, seed=80)
        depth = obs.depth[:, :, ::2, ::2].clone()                       # 240 x 320 keeps the fixture small
        mask = obs.mask[:, :, ::2, ::2].clone()
        g = torch.Generator().manual_seed(81)
        # outliers for the MAD rejection and holes (depth 0 inside the mask)
        depth[:, :, 120:123, 150:154] = 3.0
        depth[:, :, 100:106, 160:166] = 0.0
        depth = depth + 0.01 * torch.randn(depth.shape, generator=g) * mask
        cam = ini.estimate_initial_pose(depth, mask, obs.camera.intrinsic, 320, 240)
        save('g14_initial_pose', {'depth': depth, 'mask': mask.bool(), 'K': obs.camera.intrinsic.clone(),
                                  'width': 320, 'height': 240,
                                  'translation': cam.translation.clone(), 'log_q': cam.log_quaternion.clone(),
                                  'extrinsic': cam.extrinsic.clone(),
                                  'viewports': ini._masks_to_viewports(mask, 10.0).clone()})
    finally:
        ini.morphology = old


def g15_losses(lf):
    """Training losses (latentfusion/losses.py:33-100) on random maps.  trainutils.get_recon_criterion
    (trainutils.py:114-133) cannot be imported here (tensorboard is absent); its five non-VGG branches are
    one-liners over torch.nn and losses.HardPixelLoss and are written out below."""
    from torch import nn
    from latentfusion import losses

    def get_recon_criterion(name, k):
        return {'l1': lambda: nn.L1Loss(), 'smooth_l1': lambda: nn.SmoothL1Loss(),
                'hard_l1': lambda: losses.HardPixelLoss(nn.L1Loss, k=k),
                'hard_smooth_l1': lambda: losses.HardPixelLoss(nn.SmoothL1Loss, k=k),
                'binary_cross_entropy': lambda: nn.BCEWithLogitsLoss(reduction='none')}[name]()
    g = torch.Generator().manual_seed(90)
    x = torch.randn(2, 3, 1, 12, 12, generator=g)
    y = torch.randn(2, 3, 1, 12, 12, generator=g)
    m = torch.rand(2, 3, 1, 12, 12, generator=g)
    out = {}
    for name in ('l1', 'smooth_l1', 'hard_l1', 'hard_smooth_l1', 'binary_cross_entropy'):
        crit = get_recon_criterion(name, 37)
        out[name] = losses.reduce_loss(crit(x, (y > 0).float() if name == 'binary_cross_entropy' else y)).clone()
    out['beta_0.01'] = losses.beta_prior_loss(m, 0.01, 0.01).clone()
    out['beta_2_3_sum'] = losses.beta_prior_loss(m, 2.0, 3.0, reduction='sum').clone()
    save('g15_losses', {'x': x, 'y': y, 'm': m, 'k': 37, 'out': out})


def make_bop_fixture(root):
    """A tiny scene in the BOP `lm` layout (48 x 36 images, 5 frames, two objects per frame) written with
    PIL/json only: data for both readers, no reference code involved."""
    import json
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(3)
    ds, scene = os.path.join(root, 'lm'), os.path.join(root, 'lm', 'test', '000002')
    for d in ('models', 'models_eval'):
        os.makedirs(os.path.join(ds, d), exist_ok=True)
    for d in ('rgb', 'depth', 'mask_visib'):
        os.makedirs(os.path.join(scene, d), exist_ok=True)
    info = {'2': {'diameter': 247.5, 'min_x': -107.8, 'min_y': -60.9, 'min_z': -109.7, 'size_x': 215.7, 'size_y': 121.9,
                  'size_z': 219.4},
            '5': {'diameter': 201.4, 'min_x': -50.4, 'min_y': -90.9, 'min_z': -96.9, 'size_x': 100.8, 'size_y': 181.8,
                  'size_z': 193.7}}
    json.dump(info, open(os.path.join(ds, 'models_eval', 'models_info.json'), 'w'))
    verts = rng.uniform(-100, 100, size=(12, 3))
    with open(os.path.join(ds, 'models_eval', 'obj_000002.ply'), 'w') as f:
        f.write('ply\nformat ascii 1.0\nelement vertex 12\nproperty float x\nproperty float y\nproperty float z\nend_header\n')
        for v in verts:
            f.write('%.4f %.4f %.4f\n' % tuple(v))
    cam, gt = {}, {}
    for fi in range(5):
        cam[str(fi)] = {'cam_K': [572.4, 0.0, 24.0 + fi, 0.0, 573.6, 18.0, 0.0, 0.0, 1.0], 'depth_scale': 1.0 if fi % 2 else 0.1}
        objs = []
        for oid in ((5, 2) if fi % 2 else (2, 5)):
            a = rng.uniform(-3, 3, size=3)
            q = np.concatenate(([np.cos(np.linalg.norm(a) / 2)], np.sin(np.linalg.norm(a) / 2) * a / np.linalg.norm(a)))
            w, x, y, z = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            objs.append({'obj_id': oid, 'cam_R_m2c': [float(v) for v in R.reshape(-1)],
                         'cam_t_m2c': [float(v) for v in rng.uniform(-80, 80, size=2)] + [float(rng.uniform(600, 1100))]})
        gt[str(fi)] = objs
        Image.fromarray(rng.randint(0, 256, size=(36, 48, 3)).astype(np.uint8)).save(os.path.join(scene, 'rgb', '%06d.png' % fi))
        Image.fromarray(rng.randint(0, 20000, size=(36, 48)).astype(np.uint16)).save(os.path.join(scene, 'depth', '%06d.png' % fi))
        for oi in range(2):
            m = np.zeros((36, 48), np.uint8)
            m[8 + oi * 3:26, 10 + fi:30 + oi * 5] = 255
            Image.fromarray(m).save(os.path.join(scene, 'mask_visib', '%06d_%06d.png' % (fi, oi)))
    json.dump(cam, open(os.path.join(scene, 'scene_camera.json'), 'w'))
    json.dump(gt, open(os.path.join(scene, 'scene_gt.json'), 'w'))


def g16_bop_reader(lf):
    """latentfusion/datasets/bop.py on the fixture tree tests/golden/bop_fixture (the reference's `np.bool` is
    restored for the call, it was removed from numpy)."""
    import numpy as np
    from pathlib import Path
    root = os.path.join(OUT, 'bop_fixture')
    if not os.path.isdir(root):
        make_bop_fixture(root)
    had = hasattr(np, 'bool')
    if not had:
        np.bool = bool
    try:
        from latentfusion.datasets.bop import BOPDataset
        out = {}
        for center in (False, True):
            ds = BOPDataset(Path(root) / 'lm', Path(root) / 'lm' / 'test' / '000002', object_id=2, center_object=center)
            items = [ds[i] for i in range(len(ds))]
            out[center] = {'len': len(ds), 'ids': ds.get_ids(), 'object_scale': ds.object_scale,
                           'centroid': ds.centroid.clone(), 'quaternions': ds.quaternions.clone(),
                           'items': [{k: v.clone() for k, v in it.items()} for it in items],
                           'sample_evenly_3': ds.sample_evenly(3).clone(),
                           'denorm_E': ds.denormalize_extrinsic(items[1]['extrinsic']).clone()}
        save('g16_bop_reader', out)
    finally:
        if not had:
            del np.bool


def g17_api_helpers(lf):
    """Small host-side helpers of the API surface: Camera lattices / back-projection, object lattice, point-set
    statistics, spherical helpers, rigid-matrix edits, batch/view concat, distances, functional."""
    import latentfusion.functional as LF
    from latentfusion import distances, three
    from latentfusion.modules.geometry import CameraToObjectTransform
    g = torch.Generator().manual_seed(110)
    cam = rand_cameras(lf, 2, zoomed_size=8, dist=1.3, seed=111)
    depth = 0.8 + 0.4 * torch.rand(2, 1, 5, 6, generator=g)
    pts = torch.randn(40, 3, generator=g)
    th, ph = torch.rand(7, generator=g) * 3, torch.rand(7, generator=g) * 3
    E = cam.extrinsic.clone()
    a, b = torch.randn(6, 4, generator=g), torch.randn(6, 4, generator=g)
    t4 = torch.randn(4, 3, 5, 5, generator=g)
    out = {
        'cam': cam_dict(cam), 'depth': depth, 'pts': pts, 'th': th, 'ph': ph, 'E': E, 'a': a, 'b': b, 't4': t4,
        'fov_u': cam.fov_u.clone(), 'fov_v': cam.fov_v.clone(),
        'uvz': [t.clone() for t in cam.pixel_coords_uvz((3, 4, 5))], 'uv': [t.clone() for t in cam.pixel_coords_uv((4, 5))],
        'cc': [t.clone() for t in cam.camera_coords(4)], 'dcc': [t.clone() for t in cam.depth_camera_coords(depth)],
        'doc': [t.clone() for t in cam.depth_object_coords(depth)],
        'obj_coords': CameraToObjectTransform(1.0).get_obj_coords(4).clone(),
        'bound': three.points_bound(pts), 'radius': three.points_radius(pts), 'diameter': three.points_diameter(pts),
        'centroid': three.points_centroid(pts), 'bsize': three.points_bounding_size(pts),
        's2c': three.spherical_to_cartesian(th, ph, 2.0), 'qsph': three.quaternion.from_spherical(th, ph, 2.0),
        'scale_m': three.scale_matrix(E, 1.7), 'translate_m': three.translate_matrix(E, torch.tensor([0.1, -0.2, 0.3])),
        'e2q': three.extrinsic_to_quat(E),
        'vcat': three.vcat((torch.arange(12.).view(6, 2), torch.arange(100., 106.).view(3, 2)), batch_size=3),
        'vsplit': [t.clone() for t in three.vsplit(torch.arange(18.).view(9, 2), [1, 2])],
        'cosd': distances.cosine_distance(a, b), 'pair_c': distances.pairwise_distance(a, b),
        'pair_e': distances.pairwise_distance(a, b, metric='euclidean'), 'dist_c': distances.distance(a, b, dim=1),
        'dist_e': distances.distance(a, b, metric='euclidean', dim=1),
        'outer': {m: distances.outer_distance(a, b, metric=m) for m in ('cosine', 'euclidean', 'inner', 'ols_coef')},
        'norm': LF.normalize(t4, (0.1, 0.2, 0.3), (0.5, 0.6, 0.7)), 'denorm': LF.denormalize(t4, (0.1, 0.2, 0.3), (0.5, 0.6, 0.7)),
        'unit': LF.unit_normalize(t4, 1), 'amp': LF.absolute_max_pool(t4, 0),
    }
    save('g17_api_helpers', out)


def g18_training_prep(lf):
    """Training-side host helpers: process_batch (recon/utils.py:68-127) on a 2 x 2-view synthetic batch and the
    seeded orientation samplers (three/orientation.py:9-123); torch's global RNG is seeded before each call."""
    from latentfusion import three
    from latentfusion.recon import utils as RU
    obs = synth_obs(lf, 4, seed=120)
    one = {'extrinsic': obs.camera.extrinsic.view(2, 2, 4, 4).clone(), 'intrinsic': obs.camera.intrinsic.view(2, 2, 3, 4).clone(),
           'mask': obs.mask.view(2, 2, 480, 640)[..., ::8, ::8].clone(), 'render': obs.color.view(2, 2, 3, 480, 640)[..., ::8, ::8].clone(),
           'depth': obs.depth.view(2, 2, 480, 640)[..., ::8, ::8].clone()}
    one['intrinsic'][..., :2, :] /= 8.0                                       # 60 x 80 frames keep the fixture small
    batch = {'in': one, 'out_gt': one}                                       # (stored once)
    torch.manual_seed(7)
    out = RU.process_batch(batch, 1.0, 1.5, 24, 'cpu')
    res = {k: {'image': v['image'].clone(), 'mask': v['mask'].clone(), 'depth': v['depth'].clone(), 'cam': cam_dict(v['camera'])}
           for k, v in out.items()}
    samp = {}
    for name, args in (('sample_hemisphere_rays', (9, (0., 0., 1.))), ('sample_segment_rays', (9, (0., 0., 1.), 0.2, 1.0)),
                       ('sample_segment_quats', (9, (0., 1., 0.), 0.2, 1.0))):
        torch.manual_seed(5)
        samp[name] = getattr(three.orientation, name)(*args).clone()
    samp['spiral_orbit'] = three.orientation.spiral_orbit(7).clone()
    save('g18_training_prep', {'batch_views': one, 'out': res, 'samplers': samp})


def g19_metropolis(lf):
    """MetropolisPoseEstimator (pose/estimation.py:219-295): four steps of the loop of `_estimate` driven through the
    reference's own `_refine_pose` / `_track_best_items`, with every random draw (the two randn_like of
    pu.perturb_camera, the rand_like of the acceptance test) recorded so that the oracle and the HIP path can replay
    them.  `_estimate` itself is not callable here (`initial_pose` needs skimage), so its loop body (:257-270) is
    driven from given sample cameras.  Model / object / target (with its colour frame: the latent term needs it) are those of g12_latent_code (same seeds)."""
    from latentfusion.pose import estimation, utils as pu
    from latentfusion.utils import ExponentialScheduler
    S, C, V, N, K = 16, 8, 4, 5, 6
    model, cks, dist = _model(lf, S, C, 'gru', seed=60)
    ref_obs = synth_obs(lf, V, seed=61)
    target = synth_obs(lf, 1, seed=62)
    target.color = torch.round(target.color * 255.0) / 255.0          # 8-bit colours, as stored by g12
    z_obj = model.build_latent_object(ref_obs)
    g12 = torch.load(os.path.join(OUT, 'g12_latent_code.pt'), weights_only=False)
    assert torch.equal(g12['z_obj'], z_obj), 'g19 must share the model / object / target of g12_latent_code'
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.1, 'mask': 0.2, 'latent': 0.5}
    est = estimation.MetropolisPoseEstimator(model=model, num_samples=N, num_iters=K, ranking_size=3, loss_weights=weights,
                                             translation_std=0.004, quaternion_std=3.0 / 180.0 * math.pi)
    torch.manual_seed(25)
    camera = pu.sample_cameras_with_estimate(N, target.camera)
    init = cam_dict(camera)
    error = torch.full((N,), 100.0)
    temp_weight = 1.0 / target.camera.translation[:, -1].mean().item()
    sched = ExponentialScheduler(temp_weight * 0.1, temp_weight * 0.005, num_steps=K)
    log = []
    real_randn, real_rand = torch.randn_like, torch.rand_like

    def randn_like(t, *a, **k):
        r = real_randn(t, *a, **k)
        log.append(('randn', r.clone()))
        return r

    def rand_like(t, *a, **k):
        r = real_rand(t, *a, **k)
        log.append(('rand', r.clone()))
        return r
    steps, ranking = [], []
    torch.manual_seed(26)
    torch.randn_like, torch.rand_like = randn_like, rand_like
    try:
        for step in range(K):
            T = sched.get(step)
            del log[:]
            with torch.no_grad():
                camera, error, n_acc = est._refine_pose(z_obj, camera.clone(), error.clone(), target_obs=target, temperature=T)
            kinds = [k for k, _ in log]
            assert kinds == ['randn', 'randn', 'rand'], kinds
            est._track_best_items(ranking, step, camera, error)
            steps.append({'temperature': T, 'noise_t': log[0][1], 'noise_q': log[1][1], 'thresholds': log[2][1],
                          'error': error.clone(), 'num_accepted': n_acc, 't': camera.translation.clone(),
                          'log_q': camera.log_quaternion.clone()})
    finally:
        torch.randn_like, torch.rand_like = real_randn, real_rand
    print('g19 accepted per step:', [s_['num_accepted'] for s_ in steps])
    save('g19_metropolis', {'model_fixture': 'g12_latent_code', 'weights': weights, 'num_samples': N, 'num_iters': K,
                            'ranking_size': 3, 'translation_std': 0.004, 'quaternion_std': 3.0 / 180.0 * math.pi,
                            'init': init, 'steps': steps,
                            'ranking_error': torch.tensor([float(e) for _, e, _ in ranking]),
                            'ranking_step': torch.tensor([s_ for _, _, s_ in ranking]),
                            'ranking_t': torch.cat([c.translation for c, _, _ in ranking]),
                            'ranking_log_q': torch.cat([c.log_quaternion for c, _, _ in ranking])})


def g20_released_width(lf):
    """BASELINE cfg 3 at released WIDTH: the structure of tools/train/train.sh:28-66 with the resolution scaled down
    (64^2 images, 16^3 volume) but 64-96 channel 2-D and 3-D blocks, i.e. the layers that take the wide
    (Winograd-GEMM) kernels on the HIP path.  Pins: encode, decode + camera gradients, the pose loss + its camera
    gradients (what RenderLoopEngine computes), and one cross-entropy evaluation (flip augmentation, loss order).
    The target frame is g7_adam_trace's (same generator and seed), not stored again."""
    from latentfusion.recon.models import Sculptor, Photographer
    from latentfusion.recon import fusion
    from latentfusion.recon.inference import LatentFusionModel
    from latentfusion.pose import estimation, utils as pu
    from latentfusion.modules.geometry import Camera
    torch.manual_seed(130)
    S_in = 64
    sc = Sculptor(in_size=S_in, image_config=[[64, 'D', 64, 'D', 96], [96, 64]], camera_config=[8, 64],
                  object_config=[64, 64], projection_type='factor', input_color=True, input_depth=False, input_mask=True,
                  scale_mode='nearest').eval()
    S = sc.out_size
    ph = Photographer(in_size=S, image_config=[[64, 'D', 96], [96, 'U', 64, 'U', 64, 'U', 64]], camera_config=[64, 64],
                      object_config=[], projection_type='factor', predict_depth=True, predict_mask=True,
                      scale_mode='nearest').eval()
    fu = fusion.get_fuser('pool:mean', 64, 1.0).eval()
    for m in (sc, ph):
        for k, p in m.named_parameters():
            if k.endswith('bias'):
                p.data.normal_(0, 0.1)
    assert S == 16 and ph.out_size == S_in, (S, ph.out_size)
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S_in, 0.5, slack=0.5)
    model = LatentFusionModel(sc, fu, ph, dist, 'cpu')
    ref_obs = synth_obs(lf, 3, seed=131)
    pre = model.preprocess_observation(ref_obs)
    z_obj = model.build_latent_object(ref_obs)
    # decode + camera gradients of random functionals of the logits
    cam = rand_cameras(lf, 3, zoomed_size=S_in, dist=dist, seed=132)
    for name in ('log_quaternion', 'translation', 'viewport'):
        setattr(cam, name, getattr(cam, name).detach().requires_grad_(True))
    y, lat, _ = ph.decode(z_obj, cam, return_latent=True, apply_mask=True)
    g = torch.Generator().manual_seed(133)
    wd = torch.randn(y['depth_logits'].shape, generator=g)
    wm = torch.randn(y['mask_logits'].shape, generator=g)
    ((y['depth_logits'] * wd).sum() + (y['mask_logits'] * wm).sum()).backward()
    out = {'sculptor': ck(sc), 'fuser': ck(fu), 'photographer': ck(ph), 'camera_dist': dist,
           'obs_pre': {'color': pre.color.clone(), 'depth': pre.depth.clone(), 'mask': pre.mask.clone(), 'cam': cam_dict(pre.camera)},
           'z_obj': z_obj.clone(), 'cam': cam_dict(cam), 'wd': wd, 'wm': wm,
           'y': {k: v.detach().clone() for k, v in y.items()}, 'latent': lat.detach().clone(),
           'g_log_q': cam.log_quaternion.grad.clone(), 'g_t': cam.translation.grad.clone(), 'g_viewport': cam.viewport.grad.clone()}
    # pose loss of N samples around the target pose + gradients of the mean weighted loss (estimation.py:596-617)
    target = synth_obs(lf, 1, seed=22)
    g7 = torch.load(os.path.join(OUT, 'g7_adam_trace.pt'), weights_only=False)
    assert torch.equal(g7['target']['depth'], target.depth), 'g20 reuses the target frame of g7_adam_trace'
    torch.manual_seed(134)
    init = pu.sample_cameras_with_estimate(4, target.camera)
    zc = init.zoom(None, model.input_size, model.camera_dist)
    for name in ('log_quaternion', 'translation', 'viewport'):
        setattr(zc, name, getattr(zc, name).detach().requires_grad_(True))
    weights = {'depth': 1.0, 'ov_depth': 0.3, 'iou': 0.1, 'mask': 0.2}
    pred, _ = model.render_latent_object(z_obj, zc, return_latent=True)
    ld = estimation.default_pose_loss(target, zc.denormalize_depth(pred['depth'].squeeze(0)), pred['mask_logits'].squeeze(0), zc)
    total = sum(weights[k] * v for k, v in ld.items())
    total.mean().backward()
    out['loss'] = {'init': cam_dict(init), 'zoomed': cam_dict(zc), 'weights': weights,
                   'components': {k: v.detach().clone() for k, v in ld.items()}, 'total': total.detach().clone(),
                   'g_log_q': zc.log_quaternion.grad.clone(), 'g_t': zc.translation.grad.clone(),
                   'g_viewport': zc.viewport.grad.clone()}
    # one cross-entropy evaluation with the linemod preset's weights (configs/cross_entropy_linemod.toml) + a mask term
    torch.manual_seed(135)
    cams = pu.sample_cameras_with_estimate(6, target.camera, hemisphere=True, upright=True)
    ce = estimation.CrossEntropyPoseEstimator(
        model=model, num_samples=24, num_elites=8, num_iters=1, num_gmm_components=2, learning_rate=0.9, sample_flipped=True,
        ranking_size=4, loss_weights={'depth': 1.0, 'ov_depth': 0.0, 'iou': 0.0, 'mask': 0.0})
    allc = Camera.cat([cams, pu.flip_camera(cams, axis=(0.0, 0.0, 1.0)), pu.flip_camera(cams, axis=(0.0, 1.0, 0.0)),
                       pu.flip_camera(cams, axis=(1.0, 0.0, 0.0))])
    with torch.no_grad():                                   # the body of _refine_pose (estimation.py:382-400)
        zd, zl, _, zcam = ce._render_observation(z_obj, allc)
        ld2 = ce.loss_func(target, zd, zl, zcam)
        loss = sum(estimation.weigh_losses(ld2, ce.loss_weights).values())
    # ... and through _refine_pose itself with the sampler stubbed: elites come back sorted by the same losses
    ce._sample_poses = lambda gmm, n: torch.cat([cams.translation, cams.log_quaternion], dim=-1)
    with torch.no_grad():
        elite_cams, elite_loss = ce._refine_pose(z_obj, target, None, None, 8, cams[0])
    assert torch.allclose(elite_loss, torch.sort(loss)[0][:8])
    out['ce'] = {'cams': cam_dict(cams), 'all_cams': cam_dict(allc), 'loss': loss.clone(), 'order': torch.argsort(loss),
                 'weights': dict(ce.loss_weights), 'elite_loss': elite_loss.clone(),
                 'elite_log_q': elite_cams.log_quaternion.clone()}
    save('g20_released_width', out)


def g21_bop_scene(lf):
    """The committed BOP-layout fixture scene through the reference's own flow (tools/poserbpf_comparison.py:195-215,
    111-124): BOPDataset -> Observation.from_dataset(sample_evenly views) -> preprocess -> build_latent_object, then one
    cross-entropy evaluation (estimation.py:382-400) of injected sample cameras against a held-out frame of the
    scene.  Network = g7_adam_trace's SYN(16,8) checkpoints (same seeds)."""
    import numpy as np
    from pathlib import Path
    from latentfusion.observation import Observation
    from latentfusion.pose import estimation, utils as pu
    from latentfusion.modules.geometry import Camera
    root = os.path.join(OUT, 'bop_fixture')
    had = hasattr(np, 'bool')
    if not had:
        np.bool = bool
    try:
        from latentfusion.datasets.bop import BOPDataset
        model, cks, dist = _model(lf, 16, 8, 'gru', seed=20)
        g7 = torch.load(os.path.join(OUT, 'g7_adam_trace.pt'), weights_only=False)
        assert all(torch.equal(v, g7['photographer']['state_dict'][k]) for k, v in cks[2]['state_dict'].items())
        ds = BOPDataset(Path(root) / 'lm', Path(root) / 'lm' / 'test' / '000002', object_id=2, center_object=True)
        inds = ds.sample_evenly(3)
        input_obs = Observation.from_dataset(ds, inds=inds)
        held = [i for i in range(len(ds)) if i not in inds.tolist()]
        target_obs = Observation.from_dataset(ds, inds=torch.tensor(held[:1]))
        pre = model.preprocess_observation(input_obs)
        with torch.no_grad():
            z_obj = model.build_latent_object(input_obs)
        torch.manual_seed(140)
        cams = pu.sample_cameras_with_estimate(4, target_obs.camera, hemisphere=True, upright=True)
        ce = estimation.CrossEntropyPoseEstimator(
            model=model, num_samples=16, num_elites=5, num_iters=1, num_gmm_components=2, learning_rate=0.9, sample_flipped=True,
            ranking_size=4, loss_weights={'depth': 1.0, 'ov_depth': 0.2, 'iou': 0.1, 'mask': 0.3})
        allc = Camera.cat([cams, pu.flip_camera(cams, axis=(0.0, 0.0, 1.0)), pu.flip_camera(cams, axis=(0.0, 1.0, 0.0)),
                           pu.flip_camera(cams, axis=(1.0, 0.0, 0.0))])
        with torch.no_grad():
            zd, zl, _, zcam = ce._render_observation(z_obj, allc)
            ld = ce.loss_func(target_obs, zd, zl, zcam)
            loss = sum(estimation.weigh_losses(ld, ce.loss_weights).values())
        save('g21_bop_scene', {
            'model_fixture': 'g7_adam_trace', 'object_id': 2, 'input_inds': inds.clone(), 'target_ind': held[0],
            'input': obs_dict(input_obs), 'target': obs_dict(target_obs),
            'pre': {'color': pre.color.clone(), 'depth': pre.depth.clone(), 'mask': pre.mask.clone(), 'cam': cam_dict(pre.camera)},
            'z_obj': z_obj.clone(), 'cams': cam_dict(cams), 'all_cams': cam_dict(allc),
            'weights': dict(ce.loss_weights), 'components': {k: v.clone() for k, v in ld.items()}, 'loss': loss.clone(),
            'order': torch.argsort(loss)})
        print('g21 losses', loss)
    finally:
        if not had:
            del np.bool


def g22_photographer_skip(lf):
    """Photographer(skip_connections=True) (recon/models.py:296-313,400-425): parameter names / shapes the reference
    allocates, and the fact that its forward cannot run (quirk Q20: camera block 0 is concatenated with a skip
    tensor it has no channels for -- create_blocks(skip_connect_start=True))."""
    from latentfusion.recon.models import Sculptor, Photographer
    S, C = 8, 4
    img = [[8, 16], [16, 8]]
    out = {}
    for name, oc, cc in (('a', [C, C], [C, C]), ('b', [C, 8, C], [C, 8, C]), ('c', [], [C, C])):
        torch.manual_seed(150)
        sc = Sculptor(in_size=S, image_config=img, camera_config=cc, object_config=oc if oc else [C, C],
                      projection_type='factor', scale_mode='nearest').eval()
        ph = Photographer(in_size=S, image_config=img, camera_config=cc, object_config=oc, projection_type='factor',
                          skip_connections=True, scale_mode='nearest').eval()
        cam = rand_cameras(lf, 2, zoomed_size=S, dist=2.0, seed=151)
        x = torch.randn(2, 4, S, S)
        err = None
        with torch.no_grad():
            z, zc, zo = sc(x, cam)
            try:
                ph(z, cam, z_cam_mid=zc, z_obj_mid=zo)
            except Exception as e:                                  # noqa: BLE001
                err = (type(e).__name__, str(e)[:160])
        out[name] = {'object_config': oc, 'camera_config': cc, 'shapes': {k: tuple(v.shape) for k, v in ph.state_dict().items()},
                     'error': err}
        print('g22', name, err)
    save('g22_photographer_skip', {'in_size': S, 'image_config': img, 'cases': out})


def g23_ibr_generator(lf):
    """a14 remainder: reproject_views_batch, render_latent_ibr, blend_logits / warp_blend_logits and the facade's
    generator-driven LatentFusionModel.render_ibr (ibr.py:96-154,225-249; recon/inference.py:151-217)."""
    from latentfusion import ibr
    from latentfusion.modules import unet
    from latentfusion.observation import Observation
    from latentfusion.recon.inference import LatentFusionModel
    S, C, VI, VO = 16, 8, 3, 2
    sc, fu, ph = syn_ckpts(lf, S, C, fuser='pool:mean', seed=230)
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S, 0.5, slack=128 / S)
    torch.manual_seed(231)
    gen = unet.UNet2d(in_channels=1 + 5 * VI, out_channels=(VI, VI, VI), block_config=[[12, 'D', 16, 'D', 16], [16, 'U', 16, 'U', 12]]).eval()
    for k, prm in gen.named_parameters():
        if k.endswith('bias'):
            prm.data.normal_(0, 0.1)
    model = LatentFusionModel(sc, fu, ph, dist, 'cpu')
    model.generator = gen
    cam_in = rand_cameras(lf, VI, zoomed_size=S, dist=dist, seed=232)
    cam_out = rand_cameras(lf, VO, zoomed_size=S, dist=dist, seed=233)
    g = torch.Generator().manual_seed(234)
    z_obj = torch.randn(1, 1, C, S, S, S, generator=g)
    color = torch.rand(VI, 3, S, S, generator=g) * 2 - 1              # a preprocessed (zoomed, normalised) observation
    depth = torch.rand(VI, 1, S, S, generator=g) * 2 - 1
    mask = (torch.rand(VI, 1, S, S, generator=g) > 0.3).float()
    obs = Observation(color, depth, mask, cam_in, is_zoomed=True, is_prepared=True, is_normalized=True)
    with torch.no_grad():
        y, z_out = model.render_ibr(z_obj, obs, cam_out)
        y_out, _, img_re, dep_re, mask_o, depth_o, dist_r, dist_t = model._render_reprojections(z_obj, color, cam_in, cam_out)
        col, d_out, m_out, reproj = ibr.render_latent_ibr(ph, z_obj, cam_in, cam_out, color.unsqueeze(0), p=0.5)
        logits = torch.randn(VO, 3 * VI, S, S, generator=g)
        wb = ibr.warp_blend_logits(logits, img_re, 5)
        bl = ibr.blend_logits(logits[:, :VI], img_re)
    save('g23_ibr_generator', {
        'photographer': ck(ph), 'generator': ck(gen), 'camera_dist': dist, 'z_obj': z_obj,
        'cam_in': cam_dict(cam_in), 'cam_out': cam_dict(cam_out), 'color_in': color, 'depth_in': depth, 'mask_in': mask,
        'y': {k: v.clone() for k, v in y.items()}, 'z_out': z_out.clone(),
        'image_reproj': img_re.clone(), 'depth_reproj': dep_re.clone(), 'cam_dist_r': dist_r.clone(), 'cam_dist_t': dist_t.clone(),
        'latent_ibr': {'color': col.clone(), 'depth': d_out.clone(), 'mask': m_out.clone(), 'reproj': reproj.clone()},
        'logits': logits, 'warp_blend': [t.clone() for t in wb], 'blend': [t.clone() for t in bl]})


def g24_tile_projection(lf):
    """a4: TileProjection2d3d (modules/geometry.py:693-708) inside a Sculptor(projection_type='tile') encode, SYN(16,8);
    outputs and the gradient of a scalar w.r.t. the images (the lift's data gradient)."""
    from latentfusion.recon.models import Sculptor
    from latentfusion.recon import fusion
    S, C, V = 16, 8, 3
    torch.manual_seed(240)
    sc = Sculptor(in_size=S, image_config=[[16, 32], [32, 16]], camera_config=[C, C], object_config=[C, C],
                  projection_type='tile', input_color=True, input_depth=False, input_mask=True, scale_mode='nearest').eval()
    fu = fusion.get_fuser('pool:mean', C, 1.0).eval()
    for k, prm in sc.named_parameters():
        if k.endswith('bias'):
            prm.data.normal_(0, 0.1)
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S, 0.5, slack=128 / S)
    cam = rand_cameras(lf, V, zoomed_size=S, dist=dist, seed=241)
    g = torch.Generator().manual_seed(242)
    color = (torch.rand(1, V, 3, S, S, generator=g) * 2 - 1).requires_grad_(True)
    mask = (torch.rand(1, V, 1, S, S, generator=g) > 0.3).float()
    wz = torch.randn(1, 1, C, S, S, S, generator=g)
    x2d = torch.randn(2, 16, S, S, generator=g)
    z_obj, _ = sc.encode(fu, cam, color, None, mask)
    (z_obj * wz).sum().backward()
    with torch.no_grad():
        lifted = sc.projection_block(x2d)
    save('g24_tile_projection', {'sculptor': ck(sc), 'fuser': ck(fu), 'cam': cam_dict(cam), 'color': color.detach().clone(),
                                 'mask': mask, 'wz': wz, 'z_obj': z_obj.detach().clone(), 'grad_color': color.grad.clone(),
                                 'x2d': x2d, 'lifted': lifted.clone()})


def g25_released_arch(lf):
    """BASELINE cfg 3 at the TRUE released architecture (tools/train/train.sh:28-66: 256^2 inputs, 16^3 x 256-channel volume,
    512-channel U-Net levels, GRU fuser; 68 M parameters) evaluated by the REAL reference: 8-view reconstruction and one
    cross_entropy_linemod-style evaluation of 4 x 4 flipped cameras.  The network is pinned by a SEED -- weights from
    latentfusion_amd.synth.seeded_state_dict (sorted-key order, plain torch RNG calls), observations from the seeded SURVEY 8d
    generator (asserted identical to the product's synth.make_observation_data) -- so the fixture holds outputs only."""
    from latentfusion.recon.models import Sculptor, Photographer
    from latentfusion.recon import fusion
    from latentfusion.recon.inference import LatentFusionModel
    from latentfusion.pose import estimation, utils as pu
    from latentfusion.modules.geometry import Camera
    sys.path.insert(0, os.path.dirname(HERE))
    from latentfusion_amd import synth                                   # seeded inputs only: no product compute is used
    seed, V = 2500, 8
    sc, ph = Sculptor(**synth.RELEASED_SCULPTOR).eval(), Photographer(**synth.RELEASED_PHOTOGRAPHER).eval()
    fu = fusion.get_fuser('gru', 256, 1.0).eval()
    for i, m in enumerate((sc, fu, ph)):
        m.load_state_dict(synth.seeded_state_dict(m.state_dict(), seed + i, 0.1))
    dist = lf.recon.utils.optimal_camera_dist(615.4991, 256, 0.5, slack=0.5)
    model = LatentFusionModel(sc, fu, ph, dist, 'cpu')
    ref_obs, target = synth_obs(lf, V, seed + 10), synth_obs(lf, 1, seed + 20)
    for o, sd_ in ((ref_obs, seed + 10), (target, seed + 20)):           # the product's generator yields the same tensors
        d = synth.make_observation_data(len(o), sd_)
        assert torch.equal(d['color'], o.color) and torch.equal(d['depth'], o.depth) and torch.equal(d['mask'], o.mask)
        assert torch.allclose(d['extrinsic'], o.camera.extrinsic, atol=1e-6)      # (the camera re-derives it from log_q, t)
    with torch.no_grad():
        z_obj = model.build_latent_object(ref_obs)
        torch.manual_seed(seed + 30)
        cams = pu.sample_cameras_with_estimate(4, target.camera, hemisphere=True, upright=True)
        w = {'depth': 1.0, 'ov_depth': 0.0, 'iou': 0.0, 'mask': 0.0}     # configs/cross_entropy_linemod.toml
        ce = estimation.CrossEntropyPoseEstimator(model=model, num_samples=16, num_elites=6, num_iters=1, num_gmm_components=2,
                                                  learning_rate=0.9, sample_flipped=True, ranking_size=4, loss_weights=w)
        allc = Camera.cat([cams, pu.flip_camera(cams, axis=(0.0, 0.0, 1.0)), pu.flip_camera(cams, axis=(0.0, 1.0, 0.0)),
                           pu.flip_camera(cams, axis=(1.0, 0.0, 0.0))])
        zd, zl, _, zc = ce._render_observation(z_obj, allc)
        ld = ce.loss_func(target, zd, zl, zc)
        loss = sum(estimation.weigh_losses(ld, w).values())
    n_par = sum(p.numel() for m in (sc, fu, ph) for p in m.parameters())
    save('g25_released_arch', {
        'seed': seed, 'views': V, 'camera_dist': dist, 'params': n_par,
        'z_obj_sub': z_obj[..., ::2, ::2, ::2].clone(), 'z_obj_absmax': z_obj.abs().max(), 'z_obj_mean': z_obj.mean(),
        'z_obj_channel_mean': z_obj.mean(dim=(0, 1, 3, 4, 5)).clone(), 'z_obj_channel_sq': (z_obj ** 2).mean(dim=(0, 1, 3, 4, 5)).clone(),
        'cams': cam_dict(cams), 'all_cams': cam_dict(allc), 'zoom_viewport': zc.viewport.clone(),
        'depth_crop_sub': zd[..., ::4, ::4].clone(), 'mask_logits_crop_sub': zl[..., ::4, ::4].clone(),
        'loss_terms': {k: v.clone() for k, v in ld.items()}, 'loss': loss.clone(), 'order': torch.argsort(loss), 'weights': w})
    print('g25: params %.1f M, loss' % (n_par / 1e6), loss)


def g27_training_gradients(lf):
    """BASELINE cfg 5 at a size the reference finishes in seconds: ONE generator step of the trainer (tools/train/train_reconstruct.py:
    455-516: Sculptor.encode over the input views with the GRU fuser, Photographer.decode of the output views, hard smooth-L1 depth +
    BCE mask reconstruction losses x 25) by the REAL reference modules on SYN(32,16), 4 input + 2 output views -- loss terms and the
    gradient of every parameter, in fp32 and under torch.autocast(cpu, bfloat16) (the harness's stand-in for the trainer's
    `autocast()`, which is a CUDA context).  The network and the observations are pinned by seeds (latentfusion_amd.synth: plain
    RNG calls, no product compute), so the fixture holds reference outputs only."""
    from torch import nn
    from latentfusion import losses
    from latentfusion.recon import fusion
    from latentfusion.recon.inference import LatentFusionModel
    from latentfusion.recon.models import Photographer, Sculptor
    sys.path.insert(0, os.path.dirname(HERE))
    from latentfusion_amd import synth
    S, C, VI, VO, seed = 32, 16, 4, 2, 2700
    sck, fck, pck, dist = synth.make_syn_checkpoints(S, C, 'gru', seed, bias_std=0.1)
    model = LatentFusionModel(Sculptor.from_checkpoint(copy.deepcopy(sck)), fusion.from_checkpoint(copy.deepcopy(fck)),
                              Photographer.from_checkpoint(copy.deepcopy(pck)), dist, 'cpu')
    model.train(True)
    obs_in = model.preprocess_observation(synth_obs(lf, VI, seed + 1))
    obs_out = model.preprocess_observation(synth_obs(lf, VO, seed + 2))
    mods = {'s': model.sculptor, 'f': model.fuser, 'p': model.photographer}
    depth_crit = losses.HardPixelLoss(nn.SmoothL1Loss, k=S * S // 4)
    mask_crit = nn.BCEWithLogitsLoss(reduction='none')

    def run(autocast):
        for m in mods.values():
            m.zero_grad()
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
            z_obj, _ = model.sculptor.encode(model.fuser, obs_in.camera, obs_in.color.unsqueeze(0), None, obs_in.mask.unsqueeze(0))
            y, _, _ = model.photographer.decode(z_obj, obs_out.camera, interpret_logits=True)
            l_depth = losses.reduce_loss(depth_crit(y['depth'], obs_out.depth.unsqueeze(0)))
            l_mask = losses.reduce_loss(mask_crit(y['mask_logits'], obs_out.mask.unsqueeze(0)))
            total = 25.0 * l_depth + 25.0 * l_mask
        total.backward()
        grads = {k + '.' + n: p.grad.detach().float().clone() for k, m in mods.items() for n, p in m.named_parameters()}
        return {'depth_recon': l_depth.detach().float().clone(), 'mask_recon': l_mask.detach().float().clone(),
                'total': total.detach().float().clone()}, grads
    l32, g32 = run(False)
    l16, g16 = run(True)
    save('g27_training_gradients', {'S': S, 'C': C, 'views_in': VI, 'views_out': VO, 'seed': seed, 'bias_std': 0.1, 'camera_dist': dist,
                                    'loss_fp32': l32, 'loss_autocast_bf16': l16, 'grad_fp32': g32, 'grad_autocast_bf16': g16})
    import torch.nn.functional as F
    a = torch.cat([g16[k].reshape(-1) for k in g32]).double()
    b = torch.cat([g32[k].reshape(-1) for k in g32]).double()
    print('g27: total fp32 %.5f bf16 %.5f; cos(bf16 grad, fp32 grad) = %.4f' % (float(l32['total']), float(l16['total']),
                                                                               float(F.cosine_similarity(a, b, dim=0))))


def g28_occlusion16(lf):
    """The 16-channel OCCLUSION renderer (reference recon/models.py:305-306,378-395,427-430: UNet3d(17, 1, [[17, 16], [16, 16]]) over
    cat(z, depth coordinate), softmax over D, z * weights) under the adam_quick loop, 'factor' and 'sum' composites: what the
    reference's own Photographer + GradientPoseEstimator produce -- the render-loop engine sequences exactly this architecture on
    explicit kernels since round 5 (engine._plan_occlusion), so it gets its own reference-made fixture."""
    from latentfusion.pose import estimation, utils as pu
    from latentfusion.recon.inference import LatentFusionModel
    import tomli
    S, C, N, T = 32, 16, 6, 6
    dist = lf.recon.utils.optimal_camera_dist(615.4991, S, 0.5, slack=128 / S)
    z_obj = torch.randn(1, 1, C, S, S, S, generator=torch.Generator().manual_seed(31))
    target = synth_obs(lf, 1, seed=32)
    torch.manual_seed(33)
    init = pu.sample_cameras_with_estimate(N, target.camera)
    with open('/root/reference/configs/adam_quick.toml', 'rb') as f:
        cfg = tomli.load(f)
    cfg['args']['num_iters'] = T
    cfg['args']['num_samples'] = cfg['args']['ranking_size'] = N
    out = {'S': S, 'C': C, 'camera_dist': dist, 'cfg': cfg, 'z_obj': z_obj.clone(), 'target': obs_dict(target, lite=True),
           'init': cam_dict(init), 'variants': {}}
    for proj in ('factor', 'sum'):
        sc, fu, ph = syn_ckpts(lf, S, C, seed=30, occlusion_config=[[17, 16], [16, 16]], object_config=[C, C], projection_type=proj)
        model = LatentFusionModel(sc, fu, ph, dist, 'cpu')
        for m in (sc, fu, ph):
            for p_ in m.parameters():
                p_.requires_grad_(False)
        zoomed = init.zoom(None, model.input_size, model.camera_dist)
        with torch.no_grad():
            y0, _ = model.render_latent_object(z_obj, zoomed, return_latent=True)
        est = estimation.load_from_config(copy.deepcopy(cfg), model, track_stats=True, return_camera_history=True)
        best, stats, hist = est.estimate(z_obj, target, camera=init)
        out['variants'][proj] = {
            'photographer': ck(ph),
            'iter0': {k: y0[k].squeeze(0).clone() for k in ('depth_logits', 'mask_logits')},
            'rank_loss': stats['rank_loss'].clone(), 'depth_loss': stats['depth_loss'].clone(),
            'ov_depth_loss': stats['ov_depth_loss'].clone(), 'iou_loss': stats['iou_loss'].clone(), 'mask_loss': stats['mask_loss'].clone(),
            'argmin': torch.argmin(stats['rank_loss'], dim=1),
            'hist_log_q': torch.stack([c.log_quaternion for _, c in hist]), 'hist_t': torch.stack([c.translation for _, c in hist]),
            'best': cam_dict(best)}
    save('g28_occlusion16', out)


def main():
    lf = refharness.load_reference()
    import latentfusion.recon.utils  # noqa
    torch.set_num_threads(8)
    gens = [g0_preprocess, g1_camera, g2_resample, g3_block, g4_fusers, g5_decode, g6_loss, g7_g10_loop, g9_ibr,
            g11_released_like, g12_latent_code, g13_metrics, g14_initial_pose, g15_losses, g16_bop_reader, g17_api_helpers, g18_training_prep,
            g19_metropolis, g20_released_width, g21_bop_scene,
            g22_photographer_skip, g23_ibr_generator, g24_tile_projection, g25_released_arch, g27_training_gradients, g28_occlusion16]
    only = sys.argv[1:]                                  # e.g. `python oracle/make_golden.py g13` regenerates one group
    for fn in gens:
        if not only or any(fn.__name__.startswith(o + '_') or fn.__name__ == o for o in only):
            fn(lf)


if __name__ == '__main__':
    main()
