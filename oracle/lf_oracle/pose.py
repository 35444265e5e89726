"""ORACLE (test infrastructure, not product): pose loss and the pose-optimisation loops on CPU.

  default_pose_loss            pose/estimation.py:70-118, pose/utils.py:81-117
  GradientPoseEstimator loop   pose/estimation.py:500-713
  CrossEntropy refine step     pose/estimation.py:376-410 (GMM sampling is injected, not restated)
  Metropolis refine / loop     pose/estimation.py:219-295 (random draws are injected, not restated)
  camera sampling              pose/utils.py:28-45
"""
import copy
import math

import torch
import torch.nn.functional as F

from . import camera as camlib
from . import nets, quat
from .camera import Cam


class Obs:
    """Minimal Observation record (observation.py:71-163,225-290)."""

    def __init__(self, color, depth, mask, cam, is_zoomed=False, is_prepared=False, is_normalized=False):
        self.color, self.depth, self.mask, self.cam = color, depth, mask, cam
        self.is_zoomed, self.is_prepared, self.is_normalized = is_zoomed, is_prepared, is_normalized

    def __len__(self):
        return len(self.cam)

    def zoom(self, target_dist, target_size):
        """observation.py:225-236 (note the (dist,size) order matches Camera.zoom(size,dist) by Q5)."""
        color, cam = self.cam.zoom(self.color, target_size, target_dist, 'bilinear')
        depth, _ = self.cam.zoom(self.depth, target_size, target_dist, 'nearest')
        mask, _ = self.cam.zoom(self.mask, target_size, target_dist, 'nearest')
        return Obs(color, depth, mask, cam, True, self.is_prepared, self.is_normalized)

    def prepare(self):
        """observation.py:251-264."""
        color = (((self.color * 2.0 - 1.0) * self.mask + 1.0) / 2.0).clamp(0, 1)
        return Obs(color, self.depth * self.mask, self.mask.clone(), self.cam.clone(),
                   self.is_zoomed, True, self.is_normalized)

    def normalize(self):
        """observation.py:266-273."""
        return Obs(self.color * 2.0 - 1.0, self.cam.normalize_depth(self.depth), self.mask.clone(),
                   self.cam.clone(), self.is_zoomed, self.is_prepared, True)


class Model:
    """LatentFusionModel facade over oracle nets (recon/inference.py:14-128)."""

    def __init__(self, sculptor_ck, fuser_ck, photographer_ck, camera_dist):
        self.sck, self.fck, self.pck = sculptor_ck, fuser_ck, photographer_ck
        self.camera_dist = camera_dist
        self.input_size = sculptor_ck['args']['in_size']

    def preprocess(self, obs):
        if not obs.is_zoomed:
            obs = obs.zoom(self.camera_dist, self.input_size)
        if not obs.is_prepared:
            obs = obs.prepare()
        if not obs.is_normalized:
            obs = obs.normalize()
        return obs

    def build_latent_object(self, obs):
        obs = self.preprocess(obs)
        with torch.no_grad():
            return nets.encode(self.sck, self.fck, obs.cam, obs.color, obs.depth, obs.mask)

    def compute_latent_code(self, obs, cam):
        """Encode the (single) target crop under every candidate camera, decode, return the 2-D
        latent (inference.py:86-99 -> autoencode models.py:73-81)."""
        obs = self.preprocess(obs)
        n = len(cam)
        outs = []
        for i in range(n):                                  # one (object = 1 view) per candidate camera
            z_obj = nets.encode(self.sck, self.fck, cam[i], obs.color, obs.depth, obs.mask)
            _, lat, _ = nets.decode(self.pck, z_obj, cam[i])
            outs.append(lat[:, 0])
        return torch.cat(outs, dim=0)

    def render_latent_object(self, z_obj, cam, apply_mask=True):
        y, lat, _ = nets.decode(self.pck, z_obj, cam, apply_mask=apply_mask)
        return y, lat.squeeze(0)


def pose_loss(target, pred_depth_crop, pred_mask_logits_crop, cam):
    """default_pose_loss without the latent term (estimation.py:70-118, Appendix A8)."""
    pred_depth, _ = cam.uncrop(pred_depth_crop, 'nearest')
    pred_logits, _ = cam.uncrop(pred_mask_logits_crop, 'bilinear')
    pred_mask = torch.sigmoid(pred_logits)
    pred_depth = pred_depth * pred_mask
    invalid = (target.depth == 0) & (target.mask > 0.1)
    tmask = target.mask
    tdepth = target.depth * target.mask                         # target.prepare()
    valid = (~invalid).float()
    l1 = F.l1_loss(pred_depth, tdepth.expand_as(pred_depth), reduction='none') * valid
    ov = pred_mask * tmask
    out = {}
    out['ov_depth'] = (l1.squeeze(1) * ov.squeeze(1)).sum(dim=(-2, -1)).clamp(min=1e-5) \
        / ov.squeeze(1).sum(dim=(-2, -1)).clamp(min=1e-4)
    out['depth'] = l1.mean(dim=(1, 2, 3))
    tm_valid = tmask * valid
    inter = torch.sum(pred_mask * tm_valid, dim=(1, 2, 3))
    union = torch.sum(pred_mask, dim=(1, 2, 3)) + torch.sum(tm_valid, dim=(1, 2, 3)) - inter
    out['iou'] = torch.log(union.clamp(min=1e-4)) - torch.log(inter.clamp(min=1e-4))
    out['mask'] = F.binary_cross_entropy_with_logits(pred_logits, tmask.expand_as(pred_mask),
                                                     reduction='none').mean(dim=(1, 2, 3))
    return out


def latent_loss(z_pred, z_target):
    """1 - cosine similarity of the flattened 2-D latents (estimation.py:111-116, distances.py:5-9)."""
    a = z_pred.reshape(z_pred.shape[0], -1)
    b = z_target.reshape(z_target.shape[0], -1).expand_as(a)
    return 1.0 - torch.cosine_similarity(a, b, 1, 1e-8)


def weigh(loss_dict, weights):
    return sum(weights.get(k, 0.0) * v for k, v in loss_dict.items())


def sample_cameras_with_estimate(n, cam_est, hemisphere=False, upright=False):
    """pose/utils.py:28-45 (translation_std=0).  Consumes the global RNG like the reference:
    randn_like(translation) first, then the roll vectors of evenly_distributed_quats."""
    t = cam_est.t.expand(n, -1)
    t = t + torch.randn_like(t) * 0.0
    q = quat.evenly_distributed_quats(n, hemisphere=hemisphere, upright=upright)
    E = quat.extrinsic(t, q)
    return Cam.from_extrinsic(cam_est.K.expand(n, -1, -1), E, viewport=cam_est.viewport.expand(n, -1),
                              z_span=cam_est.z_span, width=cam_est.width, height=cam_est.height)


def gradient_estimate(model, z_obj, target, init_cams, cfg):
    """GradientPoseEstimator._estimate/_optimize_camera (estimation.py:534-679).

    cfg: the TOML dict ({'args':..., 'loss_weights':...}).  Returns a trace dict with the
    per-iteration rank losses, argmin indices, parameters and the final ranking."""
    a = cfg['args']
    weights = dict(cfg['loss_weights'])
    n_iters = a['num_iters']
    cams = init_cams.zoom(None, model.input_size, model.camera_dist)     # Q8: optimise the zoomed camera
    params, optims, scheds = [], [], []
    opt_cls = {'adam': torch.optim.Adam, 'adamw': torch.optim.AdamW, 'sgd': torch.optim.SGD,
               'adagrad': torch.optim.Adagrad}[a.get('optimizer', 'adamw')]
    for i in range(len(cams)):
        c = cams[i].clone()
        p = [c.log_q.requires_grad_(True), c.t.requires_grad_(True), c.viewport.requires_grad_(True)]
        params.append(c)
        o = opt_cls(p, lr=a['learning_rate'])
        optims.append(o)
        scheds.append(torch.optim.lr_scheduler.ReduceLROnPlateau(
            o, patience=a.get('lr_reduce_patience', 25), threshold=a.get('lr_reduce_threshold', 1e-5),
            factor=a.get('lr_reduce_factor', 0.5)))
    ranking, trace = [], {'rank_loss': [], 'argmin': [], 'log_q': [], 't': [], 'viewport': [],
                          'grad_log_q': [], 'grad_t': [], 'grad_viewport': []}
    converge = 0
    for step in range(n_iters):
        for o in optims:
            o.zero_grad()
        cam = camlib.cat(params)
        y, _ = model.render_latent_object(z_obj, cam, apply_mask=True)
        z_depth = cam.denormalize_depth(y['depth'].squeeze(0))
        ld = pose_loss(target, z_depth, y['mask_logits'].squeeze(0), cam)
        optim_loss = weigh(ld, weights)
        optim_loss.mean().backward()
        rank_loss = weigh(ld, weights).detach()
        trace['rank_loss'].append(rank_loss.clone())
        trace['argmin'].append(int(torch.argmin(rank_loss)))
        trace['log_q'].append(cam.log_q.detach().clone())
        trace['t'].append(cam.t.detach().clone())
        trace['viewport'].append(cam.viewport.detach().clone())
        trace['grad_log_q'].append(torch.cat([p.log_q.grad for p in params]).clone())
        trace['grad_t'].append(torch.cat([p.t.grad for p in params]).clone())
        trace['grad_viewport'].append(torch.cat([p.viewport.grad for p in params]).clone())
        det = cam.uncrop().detach().clone()
        prev_best = ranking[0][1] if ranking else float('inf')
        ranking.extend((det[i], rank_loss[i].item(), step) for i in range(len(det)))
        ranking.sort(key=lambda r: r[1])
        del ranking[a['ranking_size']:]
        delta = prev_best - ranking[0][1] if ranking[0][1] < prev_best else 0.0
        for i, (o, s) in enumerate(zip(optims, scheds)):
            o.step()
            s.step(rank_loss[i])
        if delta < a['converge_threshold']:
            converge += 1
        elif delta > a['converge_threshold']:
            converge = 0
        if converge >= a['converge_patience']:
            break
    trace = {k: (torch.stack(v) if torch.is_tensor(v[0]) else torch.tensor(v)) for k, v in trace.items()}
    trace['final_loss'] = torch.tensor([r[1] for r in ranking])
    trace['final_step'] = torch.tensor([r[2] for r in ranking])
    trace['final_log_q'] = torch.cat([r[0].log_q for r in ranking])
    trace['final_t'] = torch.cat([r[0].t for r in ranking])
    return camlib.cat([r[0] for r in ranking]), trace


def ce_refine(model, z_obj, target, cams, loss_weights, num_elites, sample_flipped):
    """One CrossEntropyPoseEstimator._refine_pose step for injected sample cameras
    (estimation.py:376-410 + base _render_observation :207-216).  No grad."""
    if sample_flipped:
        cams = camlib.cat([cams, camlib.flip(cams, (0.0, 0.0, 1.0)), camlib.flip(cams, (0.0, 1.0, 0.0)),
                           camlib.flip(cams, (1.0, 0.0, 0.0))])
    with torch.no_grad():
        zc = cams.zoom(None, model.input_size, model.camera_dist)
        y, _ = model.render_latent_object(z_obj, zc, apply_mask=True)
        z_mask = y['mask'].squeeze(0)
        z_depth = cams.denormalize_depth(y['depth'].squeeze(0)) * z_mask
        ld = pose_loss(target, z_depth, y['mask_logits'].squeeze(0), zc)
        loss = weigh(ld, loss_weights)
    order = torch.argsort(loss)
    return loss, order, order[:num_elites], cams


def metropolis_refine(model, z_obj, target, prev_cam, prev_error, noise_t, noise_q, thresholds, temperature,
                      loss_weights, translation_std, quaternion_std):
    """MetropolisPoseEstimator._refine_pose (estimation.py:277-295) with its three random draws injected:
    noise_t / noise_q are the randn_like draws of pu.perturb_camera (pose/utils.py:13-17, translation first),
    thresholds the rand_like draw of the acceptance test.  The latent term is always evaluated: the target code is
    computed under the PERTURBED, un-zoomed cameras (estimation.py:279), the prediction under their zoom (base
    _render_observation :207-216).  Returns (cameras, errors, accept mask, raw losses before accept/reject)."""
    cam = prev_cam.like(t=prev_cam.t + noise_t * translation_std, log_q=prev_cam.log_q + noise_q * quaternion_std)
    with torch.no_grad():
        z_target_latent = model.compute_latent_code(target, cam)
        zc = cam.zoom(None, model.input_size, model.camera_dist)
        y, lat = model.render_latent_object(z_obj, zc, apply_mask=True)
        z_depth = cam.denormalize_depth(y['depth'].squeeze(0)) * y['mask'].squeeze(0)
        ld = pose_loss(target, z_depth, y['mask_logits'].squeeze(0), zc)
        ld['latent'] = latent_loss(lat, z_target_latent)
        raw = weigh(ld, loss_weights)
    accept = torch.exp((prev_error - raw) / temperature) > thresholds
    keep = ~accept
    loss = torch.where(keep, prev_error, raw)
    out = cam.like(t=torch.where(keep[:, None], prev_cam.t, cam.t), log_q=torch.where(keep[:, None], prev_cam.log_q, cam.log_q),
                   viewport=torch.where(keep[:, None], prev_cam.viewport, cam.viewport))
    return out, loss, accept, raw


def metropolis_estimate(model, z_obj, target, init_cams, draws, num_iters, ranking_size, loss_weights,
                        translation_std, quaternion_std):
    """MetropolisPoseEstimator._estimate (estimation.py:237-275) from given initial sample cameras; `draws[k]` =
    (noise_t, noise_q, thresholds) of step k.  Temperature: ExponentialScheduler(0.1 w, 0.005 w), w = 1 / mean t_z."""
    cam = init_cams.clone()
    error = torch.full((len(cam),), 100.0)
    w = 1.0 / init_cams.t[:, -1].mean().item()
    sched = ExponentialScheduler(w * 0.1, w * 0.005, num_steps=num_iters)
    ranking, trace = [], []
    for step in range(num_iters):
        T = sched.get(step)
        nt, nq, th = draws[step]
        cam, error, accept, raw = metropolis_refine(model, z_obj, target, cam, error, nt, nq, th, T, loss_weights,
                                                    translation_std, quaternion_std)
        ranking.extend((cam[i], error[i].item(), step) for i in range(len(cam)))
        ranking.sort(key=lambda r: r[1])
        del ranking[ranking_size:]
        trace.append({'temperature': T, 'error': error.clone(), 'accept': accept.clone(), 'raw': raw.clone(),
                      't': cam.t.clone(), 'log_q': cam.log_q.clone()})
    return camlib.cat([r[0] for r in ranking]), trace, ranking


class ExponentialScheduler:
    """utils.py:151-162."""

    def __init__(self, initial_value, final_value, num_steps):
        self.initial_value, self.final_value, self.num_steps = initial_value, final_value, num_steps
        self.mean_lifetime = -(num_steps - 1) / math.log(final_value / initial_value)

    def get(self, step):
        if step >= self.num_steps:
            return self.final_value
        return self.initial_value * math.exp(-step / self.mean_lifetime)


def optimal_camera_dist(focal_length, size, radius, slack=1.5):
    """recon/utils.py:13-22."""
    theta = math.atan2(size / 2.0, focal_length)
    x = radius * math.cos(theta) / math.sin(theta)
    return math.sqrt(x ** 2 + radius ** 2 - 2 * x * radius * math.cos(math.pi / 2.0 - theta)) + slack
