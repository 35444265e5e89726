"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the
host-side mirror of the reference interface (Camera, Observation, block grammar, estimators'
control logic) is pinned to the reference's golden vectors.  No kernel is launched here."""
import copy
import ctypes
import os
import re

import pytest
import torch

import lf_oracle as O
from lf_oracle import nets, pose as opose

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, atol=1e-5, rtol=1e-4):
    torch.testing.assert_close(a, b, atol=atol, rtol=rtol)


def prod_camera(d):
    from latentfusion_amd.modules.geometry import Camera
    return Camera(d['K'].clone(), None, d['z_span'], d['viewport'].clone(), width=d['width'], height=d['height'],
                  log_quaternion=d['log_q'].clone(), translation=d['t'].clone())


# ---------------------------------------------------------------------------------------------
def test_cabi_exports_every_declared_symbol():
    from latentfusion_amd import _lib
    L = _lib.lib()                               # loads (and resolves) without a GPU
    # the product ABI (what a maintainer binds) and the experimental header (A/B switches, superseded kernels): each header
    # declares exactly the entry points of its binding table, every one of them is exported, and the two do not overlap
    tables = (('lf_hip.h', _lib.SIGNATURES), ('lf_hip_experimental.h', _lib.EXPERIMENTAL_SIGNATURES))
    for fname, table in tables:
        header = open(os.path.join(ROOT, 'include', fname)).read()
        header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)      # (comments mention functions of the other header)
        declared = set(re.findall(r'\b(lf_[a-z0-9_]+)\s*\(', header))
        assert declared, 'no declarations parsed'
        assert declared == set(table), (fname, declared ^ set(table))
        for name in declared:
            assert hasattr(L, name), name
    assert not set(_lib.SIGNATURES) & set(_lib.EXPERIMENTAL_SIGNATURES)
    assert L.lf_abi_version() == 1
    assert L.lf_conv3x3_cout_padded(7) == 16 and L.lf_conv3x3_cout_padded(48) == 64
    assert L.lf_conv1x1_cout_padded(200) == 256
    assert L.lf_resample3d_bwd_coef_scratch_bytes(8, 128, 128, 128) == 8 * 512 * 18 * 4


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'latentfusion_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'lf_oracle' not in src and 'import oracle' not in src, os.path.join(dirpath, f)


def test_hip_ops_fail_loudly_without_a_device():
    from latentfusion_amd import ops, _lib
    with pytest.raises(_lib.LFHipError):
        ops.conv3x3(torch.zeros(1, 16, 4, 4, 4), torch.zeros(16, 16, 3, 3, 3), None)
    with pytest.raises(_lib.LFHipError):
        ops.resample_o2c(torch.zeros(1, 4, 4, 4, 4), torch.zeros(1, 18))


# ---------------------------------------------------------------------------------------------
def test_camera_algebra_golden(golden):
    g = golden('g1_camera')
    cam = prod_camera(g['cam'])
    close(cam.quaternion, g['quaternion'])
    close(cam.rotation_matrix, g['R'])
    close(cam.obj_to_cam, g['obj_to_cam'])
    close(cam.cam_to_obj, g['cam_to_obj'])
    close(cam.znear, g['znear'])
    close(cam.position, g['position'])
    close(cam.zoom(None, 32, 2.5).viewport, g['zoom_viewport'], atol=1e-3)
    close(cam.normalize_depth(g['depth']), g['depth_norm'])
    close(cam.denormalize_depth(g['depth'] * 2 - 1), g['depth_denorm'])
    from latentfusion_amd.three import quaternion as Q
    close(Q.qmul(g['qa'], g['qb']), g['qmul'])
    close(Q.quat_to_mat(g['qa']), g['qa_mat'])
    close(Q.mat_to_quat(g['qa_mat']), g['mat_to_quat'])
    close(Q.qlog(g['qa']), g['qlog'])
    close(Q.angular_distance(g['qa'], g['qb']), g['angdist'], atol=1e-4)
    from latentfusion_amd import three
    torch.manual_seed(g['edq_seed'])
    close(three.orientation.evenly_distributed_quats(8), g['edq8'])
    torch.manual_seed(g['edq_seed'])
    close(three.orientation.evenly_distributed_quats(6, hemisphere=True, upright=True), g['edq6_hemi_upright'])


def test_camera_container_semantics(golden):
    from latentfusion_amd.modules.geometry import Camera
    cam = prod_camera(golden('g1_camera')['cam'])
    assert len(cam) == 5 and len(cam[1:3]) == 2 and len(cam[2]) == 1
    parts = cam.split(2)
    assert [len(p) for p in parts] == [2, 2, 1]
    close(Camera.cat(parts).translation, cam.translation)
    close(cam.repeat(2).viewport[5:], cam.viewport)
    rebuilt = Camera(cam.intrinsic[:, :, :3], cam.extrinsic, viewport=cam.viewport)
    close(rebuilt.rotation_matrix, cam.rotation_matrix, atol=1e-5)
    close(rebuilt.translation, cam.translation)
    with pytest.raises(ValueError):
        Camera(cam.intrinsic, None)


def test_coefficient_blocks_reproduce_reference_grids(golden):
    """The (N,18)/(N,16) coefficient blocks handed to the HIP resampler generate the same sampling
    grids as the reference's transforms."""
    from latentfusion_amd.modules.geometry import o2c_coefficients, c2o_coefficients
    d = golden('g2_resample')['cam']
    cam, ocam, S = prod_camera(d), O.cam_from_dict(d), 8
    lin = torch.linspace(0, 1, S, dtype=torch.float64)
    k, b, a = torch.meshgrid(lin, lin, lin, indexing='ij')
    c = o2c_coefficients(cam, 1.0).double().view(-1, 6, 3)
    basis = torch.stack((torch.ones_like(a), a, b, k, a * k, b * k), dim=-1)
    grid = torch.einsum('zyxj,njc->nzyxc', basis, c)
    close(grid.float(), nets.o2c_grid(ocam, S), atol=2e-6)
    A = c2o_coefficients(cam, 1.0).double().view(-1, 4, 4)
    l = torch.stack((2 * a - 1, 2 * b - 1, 2 * k - 1, torch.ones_like(a)), -1)
    num = torch.einsum('nrc,zyxc->nzyxr', A, l)
    g2 = torch.stack((num[..., 0] / num[..., 3], num[..., 1] / num[..., 3], num[..., 2]), -1)
    close(g2.float(), nets.c2o_grid(ocam, S), atol=2e-5)
    # skewed intrinsics (K[0,1] != 0, K[1,0] != 0): the C2O map is the FULL product K @ p_cam of the reference
    # (modules/geometry.py:634), not just (fu, fv, u0, v0)
    d2 = dict(d)
    d2['K'] = d['K'].clone()
    d2['K'][:, 0, 1] = 7.5
    d2['K'][:, 1, 0] = -3.25
    cam2, ocam2 = prod_camera(d2), O.cam_from_dict(d2)
    A = c2o_coefficients(cam2, 1.0).double().view(-1, 4, 4)
    num = torch.einsum('nrc,zyxc->nzyxr', A, l)
    g3 = torch.stack((num[..., 0] / num[..., 3], num[..., 1] / num[..., 3], num[..., 2]), -1)
    close(g3.float(), nets.c2o_grid(ocam2, S), atol=2e-5)
    assert (g3 - g2).abs().max() > 1e-3
    # zoom(zs=, centroid_uvs=) (reference :287-339): explicit depth / centre override the camera's own
    z0 = cam.zoom(None, 16, 2.0)
    z1 = cam.zoom(None, 16, 2.0, zs=cam.translation[:, 2] * 2.0)
    close((z1.viewport[:, 2] - z1.viewport[:, 0]) * 2.0, z0.viewport[:, 2] - z0.viewport[:, 0], atol=1e-3)
    z2 = cam.zoom(None, 16, 2.0, centroid_uvs=torch.tensor([[100.0, 50.0]]).expand(len(cam), -1))
    close((z2.viewport[:, 0] + z2.viewport[:, 2]) / 2, torch.full((len(cam),), 100.0), atol=1e-3)


def test_observation_preprocessing_golden(golden):
    from latentfusion_amd.observation import Observation
    g = golden('g0_preprocess')
    o = g['obs']
    obs = Observation(o['color'], o['depth'], o['mask'], prod_camera(o['cam']))
    z = obs.zoom(g['target_dist'], g['target_size'])
    assert z.meta['is_zoomed'] and not z.meta['is_prepared']
    close(z.color, g['zoom']['color'])
    close(z.depth, g['zoom']['depth'])
    close(z.mask, g['zoom']['mask'])
    n = z.prepare().normalize()
    assert n.meta['is_prepared'] and n.meta['is_normalized']
    close(n.color, g['normalize']['color'])
    close(n.depth, g['normalize']['depth'])
    assert len(obs) == 2 and len(obs[0]) == 1 and len(Observation.collate(obs.to_list())) == 2


def test_block_grammar_and_config_parser():
    from latentfusion_amd.utils import parse_block_config, ExponentialScheduler
    from latentfusion_amd.modules.blocks import create_blocks
    from latentfusion_amd.modules import EqualizedConv2d
    assert parse_block_config('64,D,128:128,U,64') == [[64, 'D', 128], [128, 'U', 64]]
    assert parse_block_config('none') == [] and parse_block_config('8,8') == [8, 8]
    cfg = [64, 'D', 128, 196, 'U', 32, 'I', 16]
    blocks = create_blocks(cfg, EqualizedConv2d, 0.5)
    plan = nets.plan_blocks(cfg, 0.5)
    assert len(blocks) == len(plan) == 4
    for blk, (cin, cout, scale) in zip(blocks, plan):
        assert blk.conv1.module.weight.shape[:2] == (cout, cin)
        assert (blk.interpolate.scale_factor if blk.interpolate else 1.0) == scale
    s = ExponentialScheduler(128, 48, 10)
    assert int(s.get(0)) == 128 and abs(s.get(9) - 48) < 1e-9 and s.get(50) == 48
    with pytest.raises(ValueError):
        create_blocks([8, 'X', 8], EqualizedConv2d, 0.5)


def test_checkpoint_roundtrip_keeps_reference_keys(golden):
    from latentfusion_amd.recon.models import Photographer, Sculptor
    r = golden('g5_decode')['factor']
    ph = Photographer.from_checkpoint(r['ck'])
    assert set(ph.state_dict()) == set(r['ck']['state_dict'])
    ck2 = ph.create_checkpoint()
    for k, v in r['ck']['state_dict'].items():
        assert torch.equal(ck2['state_dict'][k], v)
    g = golden('g10_encode_gru')
    sc = Sculptor.from_checkpoint(g['sculptor'])
    assert set(sc.state_dict()) == set(g['sculptor']['state_dict'])
    assert sc.out_size == 16 and sc.out_channels == 8
    with pytest.raises(ValueError):
        ph(torch.zeros(2, 8, 16, 16, 16), prod_camera(r['cam']))       # 2 volumes vs 3 cameras


def test_presets_and_loader():
    from latentfusion_amd.pose import estimation
    est = estimation.load_from_config(os.path.join(ROOT, 'configs', 'adam_quick.toml'), model=None, num_iters=5)
    assert isinstance(est, estimation.GradientPoseEstimator) and est.num_iters == 5 and est.optimizer == 'adam'
    assert est.loss_weights['ov_depth'] == 0.3 and est.loss_weights['missing'] == 0.0
    est = estimation.load_from_config(os.path.join(ROOT, 'configs', 'cross_entropy_linemod.toml'), model=None)
    assert isinstance(est, estimation.CrossEntropyPoseEstimator) and est.sample_flipped and est.num_samples == 128
    with pytest.raises(ValueError):
        estimation.load_from_config({'type': 'nope', 'args': {}, 'loss_weights': {}}, model=None)


@pytest.mark.parametrize('name', ['adam', 'adamw', 'sgd', 'adagrad'])
def test_batched_optimizer_matches_torch_optim(name):
    """N independent torch optimisers + ReduceLROnPlateau == one batched update + lr vector."""
    from latentfusion_amd.pose.estimation import BatchedOptimizer, _PlateauLR
    g = torch.Generator().manual_seed(0)
    n = 5
    init = [torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g)]
    ref = [[p[i:i + 1].clone().requires_grad_(True) for p in init] for i in range(n)]
    cls = {'adam': torch.optim.Adam, 'adamw': torch.optim.AdamW, 'sgd': torch.optim.SGD, 'adagrad': torch.optim.Adagrad}[name]
    opts = [cls(ps, lr=0.01) for ps in ref]
    scheds = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, patience=2, threshold=1e-4, factor=0.5) for o in opts]
    mine = [p.clone().requires_grad_(True) for p in init]
    bo = BatchedOptimizer(name, mine)
    pl = _PlateauLR(n, 0.01, 2, 1e-4, 0.5)
    for step in range(25):
        grads = [torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g)]
        metric = (torch.rand(n, generator=g) + 1.0 / (step + 1)).tolist()
        for i in range(n):
            for p, gr in zip(ref[i], grads):
                p.grad = gr[i:i + 1].clone()
            opts[i].step()
            scheds[i].step(metric[i])
        for p, gr in zip(mine, grads):
            p.grad = gr.clone()
        bo.step(pl.lr)
        pl.step(metric)
        assert pl.lr == [o.param_groups[0]['lr'] for o in opts]
    for j, p in enumerate(mine):
        close(p.detach(), torch.cat([ref[i][j] for i in range(n)]).detach(), atol=1e-6, rtol=1e-5)


# ---------------------------------------------------------------------------------------------
class OracleBackedModel:
    """LatentFusionModel stand-in whose renderer is the CPU oracle: lets the estimator CONTROL
    LOGIC (optimiser, schedulers, ranking, convergence, loss) run without a GPU."""

    def __init__(self, g):
        self.device = 'cpu'
        self.pck = g['photographer']
        self.camera_dist = g['camera_dist']
        self.input_size = g['sculptor']['args']['in_size']

    def render_latent_object(self, z_obj, camera, return_latent=True, apply_mask=True):
        ocam = O.Cam(camera.intrinsic, camera.log_quaternion, camera.translation, viewport=camera.viewport,
                     z_span=camera.z_span, width=camera.width, height=camera.height)
        y, lat, _ = nets.decode(self.pck, z_obj, ocam, apply_mask=apply_mask)
        return y, lat.squeeze(0)


def _target(g):
    from latentfusion_amd.observation import Observation
    tg = g['target']
    return Observation(None, tg['depth'], tg['mask'].float(), prod_camera(tg['cam']))


def test_gradient_estimator_follows_reference_trace(golden):
    """G7: the product's pose loop (batched optimiser, plateau vector, ranking) driven by an
    oracle-backed renderer reproduces the reference's per-iteration losses, argmin indices and
    camera trajectory."""
    from latentfusion_amd.pose import estimation
    g = golden('g7_adam_trace')
    est = estimation.load_from_config(copy.deepcopy(g['cfg']), OracleBackedModel(g), track_stats=True,
                                      return_camera_history=True)
    best, stats, hist = est.estimate(g['z_obj'], _target(g), camera=prod_camera(g['init']))
    close(stats['rank_loss'], g['rank_loss'], atol=5e-5, rtol=1e-3)
    for k in ('depth_loss', 'ov_depth_loss', 'iou_loss', 'mask_loss'):
        close(stats[k], g[k], atol=1e-4, rtol=1e-3)
    assert torch.argmin(stats['rank_loss'], dim=1).tolist() == g['argmin'].tolist()
    close(torch.stack([c.log_quaternion for _, c in hist]), g['hist_log_q'], atol=2e-3, rtol=1e-2)
    close(torch.stack([c.translation for _, c in hist]), g['hist_t'], atol=2e-3, rtol=1e-2)
    close(best.log_quaternion, g['best']['log_q'], atol=2e-3, rtol=1e-2)
    close(best.translation, g['best']['t'], atol=2e-3, rtol=1e-2)
    assert len(best) == g['cfg']['args']['ranking_size']


def test_cross_entropy_step_matches_reference(golden):
    from latentfusion_amd.pose import estimation
    g, t7 = golden('g8_ce_step'), golden('g7_adam_trace')
    est = estimation.CrossEntropyPoseEstimator(model=OracleBackedModel(t7), num_samples=24, num_elites=5, num_iters=3,
                                               num_gmm_components=2, learning_rate=0.9, sample_flipped=True,
                                               ranking_size=4, loss_weights=g['weights'])
    cams, loss = est.evaluate_samples(t7['z_obj'], _target(t7), prod_camera(g['cams']))
    close(cams.log_quaternion, g['all_cams']['log_q'], atol=1e-5)
    close(loss, g['loss'], atol=5e-5, rtol=1e-3)
    assert torch.argsort(loss).tolist() == g['order'].tolist()


def test_pose_loss_golden(golden):
    from latentfusion_amd.pose.loss import default_pose_loss, weigh_losses
    from latentfusion_amd.observation import Observation
    g = golden('g6_loss')
    tg = g['target']
    target = Observation(None, tg['depth'], tg['mask'].float(), prod_camera(tg['cam']))
    cam = prod_camera(g['cam'])
    cam.viewport.requires_grad_(True)
    cam.translation.requires_grad_(True)
    dn = g['depth_n'].clone().requires_grad_(True)
    lg = g['logits'].clone().requires_grad_(True)
    ld = default_pose_loss(target, cam.denormalize_depth(dn), lg, cam)
    for k, v in g['loss'].items():
        close(ld[k], v)
    sum(weigh_losses(ld, g['weights']).values()).mean().backward()
    close(dn.grad, g['g_depth_n'], atol=1e-7, rtol=1e-3)
    close(cam.viewport.grad, g['g_viewport'], atol=1e-6, rtol=1e-3)
    close(cam.translation.grad, g['g_t'], atol=1e-6, rtol=1e-3)


def test_synthetic_workload_is_consumable_by_the_oracle():
    """bench.py feeds the same generated checkpoints to the HIP path and to the CPU oracle."""
    from latentfusion_amd import synth
    sck, fck, pck, dist = synth.make_syn_checkpoints(8, 4, 'pool:mean', seed=1, bias_std=0.1)
    from latentfusion_amd.recon.models import Photographer, Sculptor
    assert set(Sculptor.from_checkpoint(sck).state_dict()) == set(sck['state_dict'])
    assert set(Photographer.from_checkpoint(pck).state_dict()) == set(pck['state_dict'])
    d = synth.make_observation_data(2, seed=3, height=60, width=80)
    cam = O.Cam.from_extrinsic(d['intrinsic'], d['extrinsic'], width=80, height=60)
    obs = opose.Obs(d['color'], d['depth'], d['mask'], cam)
    model = opose.Model(sck, fck, pck, dist)
    z = model.build_latent_object(obs)
    assert z.shape == (1, 1, 4, 8, 8, 8) and torch.isfinite(z).all()


def test_camera_vcat_and_translate(golden):
    from latentfusion_amd.modules.geometry import Camera
    cam = prod_camera(golden('g1_camera')['cam'])
    a, b = cam[0:2], cam[2:4]                      # two "objects" x one view each, twice
    v = Camera.vcat([a, b], batch_size=2)          # -> (object0: a0,b0), (object1: a1,b1)
    close(v.translation, torch.stack((cam.translation[0], cam.translation[2], cam.translation[1], cam.translation[3])))
    moved = cam.clone().translate(torch.tensor([0.01, -0.02, 0.03]))
    close(moved.position, cam.position + torch.tensor([0.01, -0.02, 0.03]), atol=1e-6)
    close(moved.rotation_matrix, cam.rotation_matrix)


def test_from_checkpoint_training_format(golden, tmp_path):
    """LatentFusionModel.from_checkpoint on the trainer's nested format (trainutils.py:274-285):
    {'args','epoch','name','modules':{'sculptor','fuser','photographer',...}} incl. legacy keys."""
    from latentfusion_amd.recon.inference import LatentFusionModel
    g = golden('g7_adam_trace')
    sck = copy.deepcopy(g['sculptor'])
    pck = copy.deepcopy(g['photographer'])
    for k in ('input_color', 'input_depth', 'input_mask'):
        sck['args'].pop(k)                          # legacy checkpoint: flags live in the global args
    for k in ('predict_color', 'predict_depth', 'predict_mask'):
        pck['args'].pop(k)
    ck = {'args': {'camera_dist': g['camera_dist'], 'generator_input_depth': False, 'generator_input_mask': True,
                   'predict_color': False, 'predict_depth': True, 'predict_mask': True, 'no_discriminator': True},
          'epoch': 12, 'name': 'unit-test', 'meter_hists': {},
          'modules': {'sculptor': sck, 'fuser': copy.deepcopy(g['fuser']), 'photographer': pck}}
    path = tmp_path / 'ck.pth'
    torch.save(ck, path)
    model = LatentFusionModel.from_checkpoint(str(path), device='cpu')
    assert model.input_size == 16 and abs(model.camera_dist - g['camera_dist']) < 1e-12
    assert model.sculptor.input_mask and not model.sculptor.input_depth and model.photographer.predict_mask
    for k, v in g['photographer']['state_dict'].items():
        assert torch.equal(model.photographer.state_dict()[k], v)
    assert type(model.fuser).__name__ == 'GRUFuser'
    # train() / eval() switch the modules' mode only (reference recon/inference.py:39-49); freezing is explicit and the
    # frozen() context gives every parameter its own flag back
    params = list(model.parameters())
    assert params and all(p.requires_grad for p in params) and not model.photographer.training
    params[0].requires_grad_(False)                                   # a parameter the user froze deliberately
    model.train(True)
    assert model.photographer.training and not params[0].requires_grad and all(p.requires_grad for p in params[1:])
    model.eval()
    assert not model.sculptor.training and all(p.requires_grad for p in params[1:])
    with model.frozen():
        assert not any(p.requires_grad for p in params)
    assert not params[0].requires_grad and all(p.requires_grad for p in params[1:])
    assert not any(p.requires_grad for p in model.freeze().parameters())
    assert all(p.requires_grad for p in model.unfreeze().parameters())


def test_pose_metrics_golden(golden):
    """pose/metrics.py (rotation/translation distance, ADD, ADD-S, ADD-sym, Proj2D) against values computed
    by the reference's own metrics module (oracle/make_golden.py:g13_metrics)."""
    from latentfusion_amd.pose import metrics
    g = golden('g13_metrics')
    gt, ev = prod_camera(g['gt']), prod_camera(g['ev'])
    got = metrics.camera_metrics(gt, ev, g['points'], g['scale'])
    assert isinstance(got, list) and len(got) == 3
    for m, want in zip(got, g['metrics']):
        assert set(m) == set(want)
        for k, v in want.items():
            assert abs(float(m[k]) - v) <= 1e-5 * max(1.0, abs(v)), (k, float(m[k]), v)
    one = metrics.camera_metrics(gt[1], ev[1], g['points'], g['scale'], use_add_s=False)
    assert 'add_s' not in one and abs(float(one['add']) - g['metrics'][1]['add']) < 1e-6
    cat = metrics.concat_camera_metrics(got)
    assert len(cat['add']) == 3
    # ADD-S <= ADD always; identical cameras score zero
    assert all(float(m['add_s']) <= float(m['add']) + 1e-7 for m in got)
    same = metrics.camera_metrics(gt[0], gt[0], g['points'], 1.0)
    assert same['rotation_dist'] < 1e-3 and float(same['add']) < 1e-6


def test_initial_pose_golden(golden):
    """pose/initialization.py against the reference (golden g14): mask viewports, MAD outlier rejection,
    translation estimate, identity rotation; PoseEstimator.initial_pose goes through it."""
    from latentfusion_amd.modules.geometry import Camera
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import estimation, initialization
    g = golden('g14_initial_pose')
    mask = g['mask'].float()
    close(initialization._masks_to_viewports(mask, 10.0), g['viewports'], atol=0, rtol=0)
    cam = initialization.estimate_initial_pose(g['depth'], mask, g['K'], g['width'], g['height'])
    close(cam.translation, g['translation'], atol=1e-6, rtol=1e-6)
    close(cam.log_quaternion, g['log_q'], atol=1e-7, rtol=0)
    close(cam.extrinsic, g['extrinsic'], atol=1e-6, rtol=1e-6)
    obs = Observation(None, g['depth'][:1], mask[:1], Camera(g['K'][:1], torch.eye(4).unsqueeze(0), width=g['width'], height=g['height']))
    cam1 = estimation.PoseEstimator.initial_pose(obs)
    close(cam1.translation, g['translation'][:1], atol=1e-6, rtol=1e-6)
    # the erosion itself (discarded by the reference's guard, see the module docstring) is a real erosion
    m = torch.zeros(1, 9, 9, dtype=torch.bool)
    m[0, 2:7, 2:7] = True
    assert initialization._erode_mask(m, size=1) is m


def test_training_losses_golden(golden):
    """losses.py (hard-pixel mining, reductions, Beta prior, criterion factory) against the reference's
    own modules (golden g15); optimiser factory keeps the reference's betas."""
    from latentfusion_amd import losses
    g = golden('g15_losses')
    x, y, m = g['x'], g['y'], g['m']
    for name in ('l1', 'smooth_l1', 'hard_l1', 'hard_smooth_l1', 'binary_cross_entropy'):
        crit = losses.get_recon_criterion(name, g['k'])
        got = losses.reduce_loss(crit(x, (y > 0).float() if name == 'binary_cross_entropy' else y))
        close(got, g['out'][name], atol=1e-6, rtol=1e-6)
    close(losses.beta_prior_loss(m, 0.01, 0.01), g['out']['beta_0.01'], atol=1e-6, rtol=1e-6)
    close(losses.beta_prior_loss(m, 2.0, 3.0, reduction='sum'), g['out']['beta_2_3_sum'], atol=1e-4, rtol=1e-6)
    with pytest.raises(ValueError):
        losses.get_recon_criterion('nope')
    opt = losses.get_optimizer([torch.nn.Parameter(torch.zeros(3))], 'adam', 1e-3)
    assert opt.defaults['betas'] == (0.0, 0.99)


def test_bop_reader_golden(golden):
    """datasets/bop.BOPDataset on the committed BOP-layout fixture (tests/golden/bop_fixture) against what the
    reference's reader produced from the same files (golden g16): items, scale, quaternions, view sampling,
    centring; plus Observation.from_dataset and the intrinsics JSON helper."""
    from pathlib import Path
    from latentfusion_amd.datasets.bop import BOPDataset, read_ply_vertices
    from latentfusion_amd.observation import Observation
    from latentfusion_amd.pose import bop as pbop
    g = golden('g16_bop_reader')
    root = Path(__file__).parent / 'golden' / 'bop_fixture' / 'lm'
    for center in (False, True):
        ds = BOPDataset(root, root / 'test' / '000002', object_id=2, center_object=center)
        r = g[center]
        assert len(ds) == r['len'] and ds.get_ids() == r['ids']
        assert abs(ds.object_scale - r['object_scale']) < 1e-12
        close(ds.centroid, r['centroid'], atol=0, rtol=0)
        close(ds.quaternions, r['quaternions'], atol=1e-6, rtol=1e-6)
        for i in range(len(ds)):
            it = ds[i]
            for k, v in r['items'][i].items():
                assert it[k].dtype == v.dtype and it[k].shape == v.shape, k
                close(it[k].float(), v.float(), atol=1e-6, rtol=1e-6)
        assert torch.equal(ds.sample_evenly(3), r['sample_evenly_3'])
        close(ds.denormalize_extrinsic(ds[1]['extrinsic']), r['denorm_E'], atol=1e-4, rtol=1e-5)
    obs = Observation.from_dataset(ds, [0, 3])
    assert obs.color.shape == (2, 3, 36, 48) and obs.depth.shape == (2, 1, 36, 48) and obs.mask.shape == (2, 1, 36, 48)
    assert len(obs.camera) == 2
    pts = ds.load_pointcloud()
    assert pts.shape == (12, 3)
    close(pts, torch.tensor(read_ply_vertices(ds.pointcloud_path)) * ds.object_scale, atol=0, rtol=0)
    with pytest.raises(ValueError):
        BOPDataset(root.parent, root / 'test' / '000002', object_id=2)
    K = pbop.parse_camera_intrinsics({'fx': 572.4, 'fy': 573.6, 'cx': 325.3, 'cy': 242.0})
    assert K.shape == (3, 4) and float(K[0, 2]) == pytest.approx(325.3)


def test_api_helpers_golden(golden):
    """The small host-side helpers that complete the reference's API surface (Camera lattices and
    back-projection, object lattice, point statistics, spherical helpers, rigid edits, batch/view concat,
    distances, functional) against the reference's own output (golden g17)."""
    from latentfusion_amd import distances, functional as LF, three
    from latentfusion_amd.modules.geometry import CameraToObjectTransform
    g = golden('g17_api_helpers')
    cam = prod_camera(g['cam'])
    close(cam.fov_u, g['fov_u']); close(cam.fov_v, g['fov_v'])
    for got, want in zip(cam.pixel_coords_uvz((3, 4, 5)), g['uvz']):
        close(got, want, atol=1e-6)
    for got, want in zip(cam.pixel_coords_uv((4, 5)), g['uv']):
        close(got, want, atol=1e-6)
    for got, want in zip(cam.camera_coords(4), g['cc']):
        close(got, want, atol=1e-6)
    for got, want in zip(cam.depth_camera_coords(g['depth']), g['dcc']):
        close(got, want, atol=1e-6)
    for got, want in zip(cam.depth_object_coords(g['depth']), g['doc']):
        close(got, want, atol=1e-5)
    with pytest.raises(AttributeError):
        cam.direction                                            # the reference property has a typo and raises
    close(CameraToObjectTransform(1.0).get_obj_coords(4), g['obj_coords'], atol=0, rtol=0)
    pts, th, ph, E = g['pts'], g['th'], g['ph'], g['E']
    close(three.points_bound(pts), g['bound'], atol=0, rtol=0)
    close(three.points_radius(pts), g['radius']); close(three.points_diameter(pts), g['diameter'])
    close(three.points_centroid(pts), g['centroid']); close(three.points_bounding_size(pts), g['bsize'])
    close(three.spherical_to_cartesian(th, ph, 2.0), g['s2c']); close(three.quaternion.from_spherical(th, ph, 2.0), g['qsph'])
    close(three.scale_matrix(E, 1.7), g['scale_m'], atol=1e-6)
    close(three.translate_matrix(E, torch.tensor([0.1, -0.2, 0.3])), g['translate_m'], atol=1e-6)
    close(three.extrinsic_to_quat(E), g['e2q'], atol=1e-6)
    close(three.vcat((torch.arange(12.).view(6, 2), torch.arange(100., 106.).view(3, 2)), batch_size=3), g['vcat'], atol=0, rtol=0)
    for got, want in zip(three.vsplit(torch.arange(18.).view(9, 2), [1, 2]), g['vsplit']):
        close(got, want, atol=0, rtol=0)
    a, b, t4 = g['a'], g['b'], g['t4']
    close(distances.cosine_distance(a, b), g['cosd']); close(distances.pairwise_distance(a, b), g['pair_c'])
    close(distances.pairwise_distance(a, b, metric='euclidean'), g['pair_e'])
    close(distances.distance(a, b, dim=1), g['dist_c']); close(distances.distance(a, b, metric='euclidean', dim=1), g['dist_e'])
    for m, want in g['outer'].items():
        close(distances.outer_distance(a, b, metric=m), want, atol=1e-6)
    close(LF.absolute_max_pool(t4, 0), g['amp'], atol=0, rtol=0)


def test_training_prep_golden(golden):
    """recon.utils.process_batch and the seeded orientation samplers against the reference (golden g18): same
    tensors AND the same consumption of torch's global RNG."""
    from latentfusion_amd import three
    from latentfusion_amd.recon import utils as RU
    g = golden('g18_training_prep')
    torch.manual_seed(7)
    out = RU.process_batch({'in': g['batch_views'], 'out_gt': g['batch_views']}, 1.0, 1.5, 24, 'cpu')
    for key, want in g['out'].items():
        for k in ('image', 'mask', 'depth'):
            close(out[key][k], want[k], atol=1e-6, rtol=1e-6)
        close(out[key]['camera'].extrinsic, prod_camera(want['cam']).extrinsic, atol=2e-6, rtol=1e-5)
        close(out[key]['camera'].viewport, want['cam']['viewport'], atol=1e-4, rtol=1e-6)
    for name, args in (('sample_hemisphere_rays', (9, (0., 0., 1.))), ('sample_segment_rays', (9, (0., 0., 1.), 0.2, 1.0)),
                       ('sample_segment_quats', (9, (0., 1., 0.), 0.2, 1.0))):
        torch.manual_seed(5)
        close(getattr(three.orientation, name)(*args), g['samplers'][name], atol=1e-6, rtol=1e-6)
    close(three.orientation.spiral_orbit(7), g['samplers']['spiral_orbit'], atol=0, rtol=0)
    t = torch.randn(3, 4, 5, 6)
    assert RU.repeat_tensor_as(t, torch.zeros(2, 7, 3, 4, 5, 6)).shape == (2, 7, 3, 4, 5, 6)


def test_observation_save_load_roundtrip(tmp_path):
    """Observation.save / load in the reference's disk format (cameras.json + per-view PNGs): round trip within
    the 8-bit colour / millimetre depth quantisation; estimate_camera / zoom_estimate run."""
    from latentfusion_amd import synth
    from latentfusion_amd.observation import Observation
    obs = synth.make_observation(3, seed=4, device='cpu')
    obs.save(tmp_path / 'obs')
    assert (tmp_path / 'obs' / 'cameras.json').exists() and (tmp_path / 'obs' / '0002.depth.png').exists()
    back = Observation.load(tmp_path / 'obs')
    assert back.color.shape == obs.color.shape and len(back.camera) == 3
    assert (back.color - obs.color).abs().max().item() <= 1.0 / 255.0 + 1e-6
    assert (back.depth - obs.depth).abs().max().item() <= 1e-3 + 1e-6
    assert torch.equal(back.mask, obs.mask)
    close(back.camera.extrinsic, obs.camera.extrinsic, atol=1e-5, rtol=1e-5)
    one = Observation.load(tmp_path / 'obs', frames=1)
    assert len(one) == 1 and torch.equal(one.mask, obs.mask[1:2])
    cam = obs.estimate_camera()
    assert cam.translation.shape == (3, 3) and float(cam.translation[:, 2].min()) > 0.5
    z = obs.zoom_estimate(1.5, 32)
    assert z.color.shape[-2:] == (32, 32)
    # meta values the reference's encoder knows (paths, tensors) are written; anything else is an error, not a string
    import pathlib
    obs.meta['source'] = pathlib.Path('/data/scene')
    obs.meta['extent'] = torch.tensor([1.0, 2.0])
    obs.save(tmp_path / 'obs2')
    import json
    meta = json.load(open(tmp_path / 'obs2' / 'cameras.json'))['meta']
    assert meta['source'] == '/data/scene' and meta['extent'] == [1.0, 2.0]
    obs.meta['bad'] = object()
    with pytest.raises(TypeError):
        obs.save(tmp_path / 'obs3')


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r01_bench_1gpu.json is the last bench.py line measured on an MI355X: every field the bench
    contract names must be there, with the roofline / cpu_baseline objects of the hot-path tier."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r01_bench_1gpu.json')
    d = json.load(open(path))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['n_gpus'] == 1
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['unit'] == d['unit']
    assert abs(d['value'] - d['n_gpus'] * d['steps'] / (d['ms_per_step'] * 1e-3 * d['steps'])) < 1e-6 * d['value']


def test_photographer_skip_connections_allocation(golden):
    """Photographer(skip_connections=True): same parameter names and shapes as the reference allocates
    (recon/models.py:296-313).  Its forward cannot run in the reference (quirk Q20, pinned by g22 and reproduced on
    the GPU in tests/test_decode_gpu.py::test_skip_connections_quirk)."""
    from latentfusion_amd.recon.models import Photographer
    g = golden('g22_photographer_skip')
    for name, case in g['cases'].items():
        ph = Photographer(in_size=g['in_size'], image_config=g['image_config'], camera_config=case['camera_config'],
                          object_config=case['object_config'], projection_type='factor', skip_connections=True,
                          scale_mode='nearest')
        assert {k: tuple(v.shape) for k, v in ph.state_dict().items()} == case['shapes'], name
        assert case['error'][0] == 'RuntimeError' and 'channels' in case['error'][1]
        assert ph.create_checkpoint()['args']['skip_connections'] is True
    with pytest.raises(ValueError):
        ph(torch.zeros(1, 4, 8, 8, 8), type('C', (), {'__len__': lambda self: 1})())


def test_tap_pair_packs_match_the_kernel_table():
    """The weight packs of the ring kernels (lf_conv3d_c16_split, lf_conv3d_c16_ring_bf16) are built with one gather; the
    layout contract is the C side's tap-pair table (lf_conv3d_c16_split_pairs, a host function): pair p holds taps
    table[2p], table[2p+1] in K slots [0,16) and [16,32), a missing second tap is zeros, every tap appears exactly once."""
    import ctypes
    from latentfusion_amd import _lib, ops
    table = (ctypes.c_int * 28)()
    _lib.lib().lf_conv3d_c16_split_pairs(table)
    taps = [t for t in table if t >= 0]
    assert sorted(taps) == list(range(27)) and list(table).count(-1) == 1
    w = torch.randn(16, 16, 3, 3, 3, generator=torch.Generator().manual_seed(0))
    for transpose in (False, True):
        wt = w.transpose(0, 1).flip(dims=(2, 3, 4)) if transpose else w
        flat = wt.reshape(16, 16, 27)
        want = torch.zeros(14, 16, 32)
        for p in range(14):
            for sel in range(2):
                if table[2 * p + sel] >= 0:
                    want[p, :, sel * 16:(sel + 1) * 16] = flat[:, :, table[2 * p + sel]]
        bf = ops.pack_conv3d_c16_ring_bf16(w, transpose=transpose)
        assert bf.dtype == torch.bfloat16 and torch.equal(bf.float(), want.to(torch.bfloat16).float())
        sp = ops.pack_conv3d_c16_split(w, transpose=transpose)                        # [14][hi, lo][16][32] f16
        assert torch.equal(sp[:, 0].float(), want.half().float())
        assert torch.equal(sp[:, 1].float(), (want - want.half().float()).half().float())
        assert bf.numel() == _lib.lib().lf_conv3d_c16_ring_bf16_wpack_elems()


def test_torch_library_registration():
    """the C-ABI replacements are registered with the dispatcher as ordinary operators (namespace `lf`).  Every
    operator exists with the schema the module documents, and none has a CPU kernel behind it (LFHipError on host tensors)."""
    import torch
    import latentfusion_amd.torch_ops as T
    from latentfusion_amd._lib import LFHipError
    assert len(T.SCHEMAS) >= 19
    for name, schema in T.SCHEMAS.items():
        op = getattr(torch.ops.lf, name)
        got = str(op.default._schema)
        assert got.replace(' ', '') == ('lf::' + schema).replace(' ', ''), (got, schema)
    import pytest
    with pytest.raises(LFHipError):
        torch.ops.lf.pixelnorm(torch.randn(2, 4, 3, 3))
    with pytest.raises(LFHipError):
        torch.ops.lf.conv_block(torch.randn(1, 4, 5, 5), torch.randn(4, 4, 3, 3), None, True, True)


def test_diag_gmm_is_sklearns_algorithm():
    """pose/gmm.DiagGMM (the cross-entropy search's mixture fit without the estimator framework) against
    sklearn.mixture.GaussianMixture(covariance_type='diag', reg_covar=1e-5): from the same start the EM reaches the same
    mixture, and `sample` draws the same numbers from numpy's global generator."""
    import warnings
    import numpy as np
    import sklearn.mixture
    from latentfusion_amd.pose.gmm import DiagGMM
    rng = np.random.RandomState(3)
    for k, n in ((6, 48), (2, 9), (3, 17)):
        X = np.concatenate([rng.randn(n // k + 1, 6) * 0.07 + rng.randn(1, 6) for _ in range(k)])[:n].astype(np.float32)
        labels = rng.randint(0, k, size=n)
        labels[:k] = np.arange(k)                                   # every component owns a point
        resp = np.zeros((n, k))
        resp[np.arange(n), labels] = 1.0
        mine = DiagGMM(k, reg_covar=1e-5).fit(X, resp=resp)
        # the same start for scikit-learn: the parameters of the first M step
        start = DiagGMM(k, reg_covar=1e-5)
        start._m_step(X.astype(np.float64), resp)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref = sklearn.mixture.GaussianMixture(covariance_type='diag', n_components=k, reg_covar=1e-5, weights_init=start.weights_,
                                                  means_init=start.means_, precisions_init=1.0 / start.covariances_).fit(X.astype(np.float64))
        assert mine.n_iter_ == ref.n_iter_ and mine.converged_ == ref.converged_
        np.testing.assert_allclose(mine.weights_, ref.weights_, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(mine.means_, ref.means_, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(mine.covariances_, ref.covariances_, rtol=1e-5, atol=1e-10)
        np.testing.assert_allclose(mine.precisions_cholesky_, ref.precisions_cholesky_, rtol=1e-5)
        np.random.seed(11)
        a, ya = mine.sample(32)
        ref.weights_, ref.means_, ref.covariances_ = mine.weights_, mine.means_, mine.covariances_
        np.random.seed(11)
        b, yb = ref.sample(32)
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)
        assert (ya == yb).all()
    # the default start (k-means labels) gives a valid mixture; fewer samples than components is an error as in scikit-learn
    np.random.seed(0)
    g = DiagGMM(6).fit(X)
    assert g.converged_ and abs(g.weights_.sum() - 1.0) < 1e-12 and (g.covariances_ > 0).all()
    import pytest
    with pytest.raises(ValueError):
        DiagGMM(6).fit(X[:3])


def test_wino_fused_xcd_order_is_a_permutation():
    """csrc/wino_fused.hip re-numbers its workgroups so that the output-channel blocks of one tile block run side by side on one
    XCD (dispatch L lands on XCD L mod 8): the same arithmetic here must visit every (tile block, channel block) exactly once, keep
    a tile block's channel blocks on ONE XCD and put them next to each other in that XCD's dispatch order."""
    for gx, gy in ((8, 2), (512, 2), (64, 4), (24, 3), (16, 16)):
        seen, xcd_of, order = set(), {}, {}
        for by in range(gy):
            for bx in range(gx):
                L = bx + gx * by                                     # hardware dispatch order: x fastest
                slot = L >> 3
                byi, bxi = slot % gy, (slot // gy) * 8 + (L & 7)
                assert 0 <= bxi < gx and 0 <= byi < gy
                seen.add((bxi, byi))
                xcd_of.setdefault(bxi, set()).add(L & 7)
                order.setdefault(bxi, []).append(slot)
        assert len(seen) == gx * gy
        assert all(len(v) == 1 for v in xcd_of.values())
        assert all(max(v) - min(v) == gy - 1 for v in order.values())   # consecutive slots of that XCD
