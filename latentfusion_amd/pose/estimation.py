"""Pose estimators driving the render loop (API mirror of latentfusion/pose/estimation.py):
`load_from_config(toml|dict, model, **overrides)` -> estimator with `.estimate(z_obj, target_obs,
camera=|cameras=)`, the TOML formats of configs/*.toml, `default_pose_loss`, track_stats /
return_camera_history outputs.

MI355X design notes (vs. the reference loop, pose/estimation.py:579-679):
  * the N pose samples live in THREE batched leaf tensors (N,3)/(N,3)/(N,4) instead of 3N tiny
    nn.Parameters; the optimiser is one batched update that reproduces torch.optim.{Adam,AdamW,
    SGD,Adagrad} element for element (a per-sample learning-rate vector carries the
    ReduceLROnPlateau state), so N independent optimisers cost one launch chain, not N;
  * exactly one device->host read-back per iteration (the N rank losses), used for the
    plateau schedulers, the ranking and the convergence test -- all host-side scalar logic;
  * weight gradients of the renderer are never formed.
"""
import contextlib
import copy
import math
from collections import defaultdict
from pathlib import Path

import torch

from .. import three
from ..modules.geometry import Camera
from ..utils import ExponentialScheduler, LinearScheduler
from . import utils as pu
from .loss import default_pose_loss, weigh_losses  # noqa: F401  (re-exported like the reference)

DEFAULT_TRANSLATION_STD = 0.01
DEFAULT_QUATERION_STD = 10.0 / 180.0 * math.pi


def _load_toml(path):
    import tomli
    with open(path, 'rb') as f:
        return tomli.load(f)


def load_from_config(config, model, **kwargs):
    if isinstance(config, (Path, str)):
        config = _load_toml(config)
    params = dict(config['args'])
    params.update(kwargs)
    kind = config['type']
    if kind == 'metropolis':
        return MetropolisPoseEstimator(model=model, **params, loss_weights=config['loss_weights'])
    if kind == 'cross_entropy':
        return CrossEntropyPoseEstimator(model=model, **params, loss_weights=config['loss_weights'])
    if kind == 'gradient':
        schedules = {k: load_schedules_from_config(v) for k, v in config.get('loss_schedules', {}).items()}
        return GradientPoseEstimator(model=model, **params, loss_weights=config['loss_weights'],
                                     loss_schedules=schedules)
    raise ValueError(f"Unknown estimator type {kind}")


def load_schedules_from_config(config):
    config = dict(config)
    kind = config.pop('type')
    if kind == 'exponential':
        return ExponentialScheduler(**config)
    if kind == 'linear':
        return LinearScheduler(**config)
    raise ValueError(f'Unknown schedule type {kind}')


class PoseEstimator:
    def __init__(self, *, model, ranking_size, loss_weights, loss_func=None, return_camera_history=False,
                 verbose=False, shard_hypotheses=False, use_engine=True, conv_mode='auto', fuse_projection=None):
        """shard_hypotheses (an addition; the reference is single-process): under torch.distributed every rank
        renders and scores only its contiguous slice of the pose hypotheses; the per-hypothesis rows (loss
        [+ camera parameters]) are all-gathered once per iteration (parallel.gather_rows) and every rank ranks the
        full set, so the returned ranking is identical on all ranks and to a one-rank run.  Host-random draws
        (initial hypotheses, GMM samples) are taken from rank 0."""
        self.model = model
        # use_engine / conv_mode / fuse_projection (additions): the fused render-and-score engine (engine.py) evaluates the
        # hypotheses when the renderer and the loss are of the kind it sequences; False selects the generic module path
        self.use_engine, self.conv_mode, self.fuse_projection = use_engine, conv_mode, fuse_projection
        self._engine_cache = None
        self.last_scored_on_engine = False
        self.shard_hypotheses = bool(shard_hypotheses)
        self.ranking_size = ranking_size
        self.loss_func = default_pose_loss if loss_func is None else loss_func
        self.loss_weights = defaultdict(float)
        self.loss_weights.update(loss_weights)
        self.return_camera_history = return_camera_history
        self.verbose = verbose

    @property
    def device(self):
        return self.model.device

    @classmethod
    def initial_pose(cls, target_obs):
        """Translation from the target's depth and mask, identity rotation (reference :148-164)."""
        from . import initialization
        return initialization.estimate_initial_pose(target_obs.depth, target_obs.mask, target_obs.camera.intrinsic,
                                                    target_obs.camera.width, target_obs.camera.height)

    def estimate(self, z_obj, target_obs, **kwargs):
        if len(target_obs) > 1:
            raise ValueError('The pose can only be estiamted for one observation at a time.')
        try:
            with self._frozen_model():
                return self._estimate(z_obj, target_obs, **kwargs)
        finally:
            # the ranking engine of this call holds the resident volume copy, the target buffers and weight packs: it does not
            # outlive the estimate (evaluate_samples() called directly builds one per (object, target) and keeps it)
            self._engine_cache = None

    def _frozen_model(self):
        """The estimators differentiate w.r.t. the cameras only: run with the network's parameters frozen (no
        weight-gradient kernels; the reference accumulates gradients nobody reads) and give every parameter
        its own flag back afterwards.  Models without the facade's frozen() (stubs in tests) run as they are."""
        frozen = getattr(self.model, 'frozen', None)
        return frozen() if callable(frozen) else contextlib.nullcontext()

    def _sharding(self):
        """(rank, size) when hypothesis sharding is active, else (0, 1)."""
        if not self.shard_hypotheses:
            return 0, 1
        from .. import parallel
        return parallel.world()

    def _sync_cameras_from_rank0(self, camera):
        """Every rank must rank the SAME hypotheses: take rank 0's (they come from host RNGs)."""
        if self._sharding()[1] == 1:
            return camera
        from .. import parallel
        blk = torch.cat((camera.log_quaternion, camera.translation, camera.viewport), dim=1).detach().clone().contiguous()
        parallel.broadcast_(blk)
        return camera._like(log_quaternion=blk[:, 0:3].clone(), translation=blk[:, 3:6].clone(), viewport=blk[:, 6:10].clone())

    def _track_best_items(self, ranking, step, items, loss):
        """Merge this step's (camera, loss) pairs into the top-`ranking_size` list; returns the
        improvement of the best loss (reference :187-205).  `loss` is a host sequence."""
        prev_best = ranking[0][1] if ranking else float('inf')
        ranking.extend((c, float(e), step) for c, e in zip(items, loss))
        ranking.sort(key=lambda r: r[1])
        del ranking[self.ranking_size:]
        best = ranking[0][1]
        return prev_best - best if best < prev_best else 0.0

    def _engine_for(self, z_obj, target_obs, schedules=()):
        """The fused HIP engine when the configuration allows it (default loss, factor-projection renderer); None selects
        the generic autograd-module path."""
        if not self.use_engine or self.loss_func is not default_pose_loss or not z_obj.is_cuda:
            return None
        from ..engine import RenderLoopEngine
        ph = getattr(self.model, 'photographer', None)
        if ph is None or not RenderLoopEngine.supports(ph, self.loss_weights):
            return None
        # a scheduled term the fused loss does not evaluate (e.g. [loss_schedules.latent] with loss_weights.latent = 0)
        # must not be dropped silently: the reference applies every scheduled weight (estimation.py:612-617)
        if any(k not in RenderLoopEngine.LOSS_KEYS + ('latent',) for k in schedules):
            return None
        fp = self.fuse_projection
        wants_x = (self.conv_mode == 'winograd_f16x3' or fp is True or (isinstance(fp, (tuple, list, set)) and 'bwd' in fp)
                   or getattr(self, 'engine_streams', 1) > 1 or getattr(self, 'engine_graph', False))
        if wants_x:                                                # measured-and-rejected variants: experimental.py
            from ..experimental import RenderLoopEngineX
            return RenderLoopEngineX(ph, z_obj, target_obs, self.loss_weights, conv_mode=self.conv_mode, fuse_projection=fp)
        return RenderLoopEngine(ph, z_obj, target_obs, self.loss_weights, conv_mode=self.conv_mode, fuse_projection=fp)

    def _ranking_engine(self, z_obj, target_obs):
        """One engine per (object, target) for the ranking-only estimators; the cache keeps both alive, so an address
        cannot be reused by another volume while the entry exists."""
        c = self._engine_cache
        if c is None or c[0] is not z_obj or c[1] is not target_obs:
            self._engine_cache = c = (z_obj, target_obs, self._engine_for(z_obj, target_obs))
            self.last_scored_on_engine = c[2] is not None         # (outlives the cache entry: bench.py reports it)
        return c[2]

    def _score_samples(self, z_obj, target_obs, cameras, z_target_latent=None):
        """Weighted pose loss of every hypothesis, no gradient: zoom, render, denormalise, multiply by the mask, loss
        (reference :207-216 + :383-401).  On the fused engine when it supports the renderer (lf_pose_loss_fwd_masked)."""
        eng = self._ranking_engine(z_obj, target_obs)
        with torch.no_grad():
            if eng is not None:
                z_camera = cameras.zoom(None, self.model.input_size, self.model.camera_dist).to(self.device)
                losses, _ = eng.forward_backward(z_camera, need_grad=False, z_target_latent=z_target_latent, masked_depth=True)
                return losses[:, 4].clone()
            zd, zl, z_lat, z_camera = self._render_observation(z_obj, cameras.to(self.device))
            ld = self.loss_func(target_obs, zd, zl, z_camera, z_pred_latent=z_lat, z_target_latent=z_target_latent)
            return sum(weigh_losses(ld, self.loss_weights).values())

    def _render_observation(self, z_obj, camera, **kwargs):
        """Zoom, render without grad, denormalise, multiply by the mask (reference :207-216)."""
        z_camera = camera.zoom(None, self.model.input_size, self.model.camera_dist)
        with torch.set_grad_enabled(kwargs.get('grad_enabled', False)):
            pred, z_latent = self.model.render_latent_object(z_obj, z_camera.to(self.device), return_latent=True)
            z_mask = pred['mask'].squeeze(0)
            z_depth = camera.denormalize_depth(pred['depth'].squeeze(0)) * z_mask
        return z_depth, pred['mask_logits'].squeeze(0), z_latent, z_camera


# ---------------------------------------------------------------------------------------------
class MetropolisPoseEstimator(PoseEstimator):
    """Metropolis-Hastings with simulated annealing (reference :219-295)."""

    def __init__(self, *, num_samples, num_iters, translation_std=DEFAULT_TRANSLATION_STD,
                 quaternion_std=DEFAULT_QUATERION_STD, **kwargs):
        super().__init__(**kwargs)
        self.num_samples, self.num_iters = num_samples, num_iters
        self.translation_std, self.quaternion_std = translation_std, quaternion_std
        self.replay_draws = None        # tests: list of recorded draws consumed in order (randn t, randn q, rand)

    def _draw(self, kind, like):
        """The three random draws of one step, in the reference's order: randn_like(translation),
        randn_like(log_quaternion) (pu.perturb_camera, pose/utils.py:13-17), rand_like(loss) (:288)."""
        if self.replay_draws is not None:
            return self.replay_draws.pop(0).to(like.device)
        return torch.randn_like(like) if kind == 'randn' else torch.rand_like(like)

    def _estimate(self, z_obj, target_obs, **kwargs):
        if kwargs.get('cameras', None) is not None:             # given sample cameras (the reference always samples)
            camera_init = camera = kwargs['cameras']
            camera = camera.to(self.device)
        else:
            camera_init = kwargs['camera'] if 'camera' in kwargs else self.initial_pose(target_obs)
            camera = pu.sample_cameras_with_estimate(self.num_samples, camera_init).to(self.device)
        error = torch.full((len(camera),), 100.0, device=self.device)
        temp_weight = 1.0 / camera_init.translation[:, -1].mean().item()
        sched = ExponentialScheduler(temp_weight * 0.1, temp_weight * 0.005, num_steps=self.num_iters)
        target_obs = target_obs.to(self.device)
        ranking, history = [], []
        self.accept_history = []
        for step in range(self.num_iters):
            camera, error, n_acc = self._refine_pose(z_obj, camera.clone(), error.clone(), target_obs, sched.get(step))
            self.accept_history.append(n_acc)
            if self._track_best_items(ranking, step, list(camera), error.tolist()) > 0:
                history.append((error, camera.clone().to('cpu')))
        cameras = Camera.cat([c for c, _, _ in ranking])
        return (cameras, history) if self.return_camera_history else cameras

    def _refine_pose(self, z_obj, prev_camera, prev_error, target_obs, temperature=1.0):
        camera = prev_camera.clone()
        camera.translation = camera.translation + self._draw('randn', camera.translation) * self.translation_std
        camera.log_quaternion = camera.log_quaternion + self._draw('randn', camera.log_quaternion) * self.quaternion_std
        with torch.no_grad():
            # the reference evaluates the latent term unconditionally, the target code under the perturbed,
            # un-zoomed cameras (estimation.py:279)
            z_target_latent = self.model.compute_latent_code(target_obs, camera)
            zd, zl, z_lat, z_camera = self._render_observation(z_obj, camera)
            ld = self.loss_func(target_obs, zd, zl, z_camera, z_pred_latent=z_lat, z_target_latent=z_target_latent)
            loss = sum(weigh_losses(ld, self.loss_weights).values())
        accept = torch.exp((prev_error - loss) / temperature) > self._draw('rand', loss)
        camera[~accept] = prev_camera[~accept]
        loss[~accept] = prev_error[~accept]
        return camera, loss, int(accept.sum().item())


# ---------------------------------------------------------------------------------------------
class CrossEntropyPoseEstimator(PoseEstimator):
    """Cross-entropy method over a diagonal GMM on (t, log_q) (reference :298-497).  Rendering and
    loss evaluation are the device work; GMM fit/sampling is scikit-learn on the host as in the
    reference."""

    def __init__(self, *, num_samples, num_elites, num_iters, num_gmm_components, learning_rate,
                 sample_flipped=False, init_hemisphere=False, init_upright=False,
                 translation_std=DEFAULT_TRANSLATION_STD, quaternion_std=DEFAULT_QUATERION_STD, **kwargs):
        super().__init__(**kwargs)
        self.num_samples, self.num_elites, self.num_iters = num_samples, num_elites, num_iters
        self.num_gmm_components, self.learning_rate = num_gmm_components, learning_rate
        self.sample_flipped, self.init_upright, self.init_hemisphere = sample_flipped, init_upright, init_hemisphere
        self.translation_std, self.quaternion_std = translation_std, quaternion_std
        self.elite_sched = ExponentialScheduler(num_samples, num_elites, num_iters)

    def _estimate(self, z_obj, target_obs, **kwargs):
        if kwargs.get('cameras', None):
            cameras, camera_init = kwargs['cameras'], kwargs['cameras'][0]
        else:
            camera_init = kwargs['camera'] if 'camera' in kwargs else self.initial_pose(target_obs)
            cameras = pu.sample_cameras_with_estimate(n=self.num_gmm_components * self.num_samples,
                                                      camera_est=camera_init, upright=self.init_upright,
                                                      hemisphere=self.init_hemisphere)
        cameras = self._sync_cameras_from_rank0(cameras.to(self.device)) if self._sharding()[1] > 1 else cameras
        gmm = self._create_gmm(self._camera_to_params(cameras).cpu())
        target_obs = target_obs.to(self.device)
        prev_gmm, ranking, history = None, [], []
        for step in range(self.num_iters):
            n_elite = int(self.elite_sched.get(step))
            cameras, losses = self._refine_pose(z_obj, target_obs, prev_gmm, gmm, n_elite, camera_init)
            prev_gmm = gmm
            gmm = self._create_gmm(self._camera_to_params(cameras).cpu())
            if self._track_best_items(ranking, step, list(cameras), losses.tolist()) > 0:
                history.append((losses, Camera.cat([c for c, _, _ in ranking])))
        # the ranked cameras go back to the estimator's device, as the reference builds them (:381): the same return type on
        # one rank (where the sampler keeps its cameras on the host, _refine_pose) and when sharded
        out = Camera.cat([c for c, _, _ in ranking]).to(self.device)
        return (out, history) if self.return_camera_history else out

    def evaluate_samples(self, z_obj, target_obs, cameras):
        """Flip-augment, render without grad, weighted loss per sample (reference :383-401)."""
        if self.sample_flipped:
            cameras = Camera.cat([cameras, pu.flip_camera(cameras, axis=(0.0, 0.0, 1.0)),
                                  pu.flip_camera(cameras, axis=(0.0, 1.0, 0.0)),
                                  pu.flip_camera(cameras, axis=(1.0, 0.0, 0.0))])
        z_target_latent = None
        if self.loss_weights.get('latent', 0.0) > 0.0:
            with torch.no_grad():
                z_target_latent = self.model.compute_latent_code(target_obs, cameras[0].to(self.device))
        rank, size = self._sharding()
        local = cameras
        if size > 1:                                               # this rank's contiguous slice of the hypotheses
            from .. import parallel
            b, e = parallel.shard_range(len(cameras), rank, size)
            local = cameras[b:e]
        with torch.no_grad():
            if len(local):
                loss = self._score_samples(z_obj, target_obs, local, z_target_latent)
            else:
                loss = torch.zeros(0, device=cameras.device)
            if size > 1:
                loss = parallel.gather_rows(loss.contiguous(), len(cameras))     # N scalars per iteration
        return cameras, loss

    def _refine_pose(self, z_obj, target_obs, prev_gmm, gmm, num_elites, camera_init):
        sample_gmm = self._combined_gmm(prev_gmm, gmm, self.learning_rate) if prev_gmm is not None else gmm
        n = self.num_samples // 4 if self.sample_flipped else self.num_samples
        # Single rank: the sampled cameras stay HOST tensors through the jitter, the three flips and the zoom (a few hundred
        # scalars; on the device this was ~100 one-microsecond launches per iteration: quaternion products, sin / cos, norms,
        # concatenations) -- the zoomed cameras reach the device in one hop inside _score_samples (round 5; the GMM is fitted
        # and sampled on the host anyway, as in the reference :449-473)
        host = self._sharding()[1] == 1
        cameras = self._params_to_camera(self._sample_poses(sample_gmm, n, 'cpu' if host else self.device), camera_init,
                                         device='cpu' if host else self.device)
        cameras, loss = self.evaluate_samples(z_obj, target_obs, cameras)
        elite = torch.argsort(loss)[:num_elites]
        return cameras[elite.to(cameras.device)], loss[elite]

    def _sample_poses(self, gmm, n, device=None):
        device = self.device if device is None else device
        params, _ = gmm.sample(n)
        params = torch.tensor(params, dtype=torch.float32, device=device)
        params[:, :3] += torch.randn_like(params[:, :3]) * self.translation_std
        params[:, 3:] += torch.randn_like(params[:, 3:]) * self.quaternion_std
        if self._sharding()[1] > 1:                                # numpy / torch host RNGs differ per rank
            from .. import parallel
            parallel.broadcast_(params)
        return params

    # 'fast': pose/gmm.DiagGMM (scikit-learn's diag-covariance EM and sampling order without the estimator framework: 0.3 instead
    # of ~3 ms per fit with the GPU idle); 'sklearn': sklearn.mixture.GaussianMixture as the reference calls it (:412-420)
    gmm_backend = 'fast'

    def _create_gmm(self, params=None):
        if self.gmm_backend == 'sklearn':
            import sklearn.mixture
            gmm = sklearn.mixture.GaussianMixture(covariance_type='diag', n_components=self.num_gmm_components,
                                                  reg_covar=1e-5)
        else:
            from .gmm import DiagGMM
            gmm = DiagGMM(self.num_gmm_components, reg_covar=1e-5)
        if params is not None:
            gmm.fit(params.numpy() if torch.is_tensor(params) else params)
        return gmm

    def _combined_gmm(self, old_gmm, new_gmm, alpha):
        import numpy as np
        if alpha > 1.0 or alpha < 0.0:
            raise ValueError('alpha must be between 0.0 and 1.0')
        out = self._create_gmm()
        out.weights_ = np.concatenate([(1.0 - alpha) * old_gmm.weights_, alpha * new_gmm.weights_], axis=0)
        out.means_ = np.concatenate([old_gmm.means_, new_gmm.means_], axis=0)
        out.covariances_ = np.concatenate([old_gmm.covariances_, new_gmm.covariances_], axis=0)
        out.precisions_cholesky_ = np.concatenate([old_gmm.precisions_cholesky_, new_gmm.precisions_cholesky_], axis=0)
        return out

    @classmethod
    def _camera_to_params(cls, camera):
        return torch.cat([camera.translation, camera.log_quaternion], dim=-1).detach()

    @classmethod
    def _params_to_camera(cls, params, camera_init, device='cpu'):
        if params.dim() == 1:
            params = params.unsqueeze(0)
        return Camera(intrinsic=camera_init.intrinsic.expand(params.shape[0], -1, -1).to(device), extrinsic=None,
                      translation=params[:, :3].to(device), log_quaternion=params[:, 3:].to(device),
                      width=camera_init.width, height=camera_init.height, z_span=camera_init.z_span)


# ---------------------------------------------------------------------------------------------
class _PlateauLR:
    """N independent torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', threshold_mode='rel')
    instances as host-side vectors."""

    def __init__(self, n, lr, patience, threshold, factor, min_lr=0.0, eps=1e-8):
        self.lr = [lr] * n
        self.best = [float('inf')] * n
        self.bad = [0] * n
        self.patience, self.threshold, self.factor, self.min_lr, self.eps = patience, threshold, factor, min_lr, eps

    def step(self, metrics):
        for i, cur in enumerate(metrics):
            if cur < self.best[i] * (1.0 - self.threshold):
                self.best[i], self.bad[i] = cur, 0
            else:
                self.bad[i] += 1
            if self.bad[i] > self.patience:
                new_lr = max(self.lr[i] * self.factor, self.min_lr)
                if self.lr[i] - new_lr > self.eps:
                    self.lr[i] = new_lr
                self.bad[i] = 0


class BatchedOptimizer:
    """One update for N independent per-sample optimisers over rows of batched parameters.
    Reproduces the update rules of torch.optim.{Adam, AdamW, SGD, Adagrad} (defaults) exactly;
    `lr` is a per-row vector."""

    def __init__(self, name, params):
        if name not in ('adam', 'adamw', 'sgd', 'adagrad'):
            raise ValueError(f'Unknow optimizer {name!r}')
        self.name, self.params, self.t = name, params, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def _step_hip_adam(self, lr_rows):
        """lf_adam_step: one launch per parameter tensor (same arithmetic as the torch branch below)."""
        from .. import _lib
        L = _lib.lib()
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        dev = self.params[0].device
        # pinned staging (two alternating buffers: the copy of step t-2 has long completed), so that the upload
        # never makes the host wait for the stream
        if not hasattr(self, '_pin') or self._pin[0].shape[1] != len(lr_rows):
            self._pin = [torch.empty(2, len(lr_rows), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        host = self._pin[self.t & 1]
        host.copy_(torch.tensor([[l / bc1 for l in lr_rows], list(lr_rows)], dtype=torch.float32))
        both = host.to(dev, non_blocking=True)
        wd = 1e-2 if self.name == 'adamw' else 0.0
        stream = torch.cuda.current_stream().cuda_stream
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad.contiguous()
            _lib.check(L.lf_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), both[0].data_ptr(),
                                      both[1].data_ptr(), math.sqrt(bc2), b1, b2, eps, wd, p.shape[0], p.shape[1], stream),
                       'lf_adam_step')

    def step(self, lr_rows):
        self.t += 1
        if self.name in ('adam', 'adamw') and all(p.is_cuda and p.is_contiguous() and p.dim() == 2 for p in self.params):
            return self._step_hip_adam(lr_rows)
        dev = self.params[0].device
        lr = torch.tensor(lr_rows, dtype=torch.float32, device=dev).unsqueeze(1)
        with torch.no_grad():
            for p, m, v in zip(self.params, self.m, self.v):
                g = p.grad
                if g is None:
                    continue
                if self.name in ('adam', 'adamw'):
                    b1, b2, eps = 0.9, 0.999, 1e-8
                    if self.name == 'adamw':
                        p.mul_(1 - lr * 1e-2)
                    m.lerp_(g, 1 - b1)
                    v.mul_(b2).addcmul_(g, g, value=1 - b2)
                    bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
                    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
                    # step size formed in double on the host like torch.optim (lr / bias_correction1)
                    step_size = torch.tensor([l / bc1 for l in lr_rows], dtype=torch.float32, device=dev).unsqueeze(1)
                    p.sub_(step_size * (m / denom))
                elif self.name == 'sgd':
                    p.sub_(lr * g)
                else:                                            # adagrad: lr_decay 0, eps 1e-10
                    v.addcmul_(g, g, value=1.0)
                    p.sub_(lr * (g / v.sqrt().add_(1e-10)))


class GradientPoseEstimator(PoseEstimator):
    """Gradient descent on (log_quaternion, translation, viewport) of the ZOOMED camera of every
    pose sample (reference :500-713)."""

    def __init__(self, *, learning_rate, num_samples, num_iters, converge_threshold, converge_patience,
                 lr_reduce_patience=25, lr_reduce_threshold=1e-5, lr_reduce_factor=0.5, track_stats=False,
                 loss_schedules=None, optimizer='adamw', use_engine=True, conv_mode='auto', engine_streams=1,
                 fuse_projection=None, engine_graph=False, **kwargs):
        super().__init__(use_engine=use_engine, conv_mode=conv_mode, fuse_projection=fuse_projection, **kwargs)
        self.engine_graph = engine_graph         # replay the engine's evaluation from a captured hipGraph (engine.forward_backward_graph)
        self.engine_streams = engine_streams     # hypothesis groups evaluated concurrently on separate HIP streams (engine.py)
        self.learning_rate, self.num_samples, self.num_iters = learning_rate, num_samples, num_iters
        self.optimizer = optimizer
        self.lr_reduce_patience, self.lr_reduce_threshold = lr_reduce_patience, lr_reduce_threshold
        self.lr_reduce_factor = lr_reduce_factor
        self.converge_threshold, self.converge_patience = converge_threshold, converge_patience
        self.loss_schedules = dict(loss_schedules) if loss_schedules else {}
        self.track_stats = track_stats

    def _estimate(self, z_obj, target_obs, **kwargs):
        if 'camera' in kwargs:
            camera = kwargs['camera']
        else:
            camera = pu.sample_cameras_with_estimate(n=self.num_samples, camera_est=self.initial_pose(target_obs))
        target_obs = target_obs.to(self.device)
        camera = camera.zoom(None, self.model.input_size, self.model.camera_dist).to(self.device)
        camera = self._sync_cameras_from_rank0(camera)
        ranking = []
        stats, history = self._optimize_camera(z_obj, target_obs, camera, iters=self.num_iters, ranking=ranking)
        best = Camera.cat([c for c, _, _ in ranking])
        if self.track_stats and self.return_camera_history:
            return best, stats, history
        if self.track_stats:
            return best, stats
        if self.return_camera_history:
            return best, history
        return best

    # one iteration = render N samples + loss + backward to the camera parameters
    def loss_and_grad(self, z_obj, target_obs, cameras, step=0, z_target_latent=None):
        z_depth, z_mask_logits, z_pred_latent = self._render_observation(z_obj, cameras)
        optim_weights = copy.copy(self.loss_weights)
        optim_weights.update({k: v.get(step) for k, v in self.loss_schedules.items()})
        loss_dict = self.loss_func(target_obs, z_depth, z_mask_logits, cameras, z_pred_latent=z_pred_latent,
                                   z_target_latent=z_target_latent)
        optim_loss = sum(weigh_losses(loss_dict, optim_weights).values())
        optim_loss.mean().backward()
        rank_loss = sum(weigh_losses(loss_dict, self.loss_weights).values()).detach()
        return loss_dict, optim_loss.detach(), rank_loss, optim_weights

    def _engine_for(self, z_obj, target_obs, schedules=()):
        eng = super()._engine_for(z_obj, target_obs, schedules=self.loss_schedules)
        if eng is not None and self.engine_streams > 1:
            eng.set_streams(self.engine_streams)
        return eng

    @classmethod
    def get_optimizer(cls, name, *args, **kwargs):
        """torch.optim factory of the reference (:566-577).  The loop itself uses BatchedOptimizer, which applies
        the same update rules to all N samples in one launch."""
        from torch import optim
        table = {'adamw': optim.AdamW, 'adam': optim.Adam, 'sgd': optim.SGD, 'adagrad': optim.Adagrad}
        if name not in table:
            raise ValueError(f'Unknow optimizer {name!r}')
        return table[name](*args, **kwargs)

    def start(self, z_obj, target_obs, cameras, ranking=None):
        """Creates the per-run loop state (parameters, optimiser, schedulers); `cameras` must
        already be zoomed and on the device.  Exposed so that bench.py can time `iterate`."""
        rank, size = self._sharding()
        shard, full_cpu = None, None
        if size > 1:
            # this rank optimises its contiguous slice of the hypotheses (own optimiser / plateau state); the ranking
            # sees all of them through one all-gather of (loss, parameters) rows per iteration
            from .. import parallel
            b, e = parallel.shard_range(len(cameras), rank, size)
            if e <= b:
                raise ValueError(f'shard_hypotheses: {len(cameras)} hypotheses cannot be split over {size} ranks')
            shard, full_cpu = (b, e, len(cameras)), cameras.to('cpu')
            cameras = cameras[b:e]
        st = self._start_local(z_obj, target_obs, cameras, ranking)
        st['shard'], st['template_full_cpu'] = shard, full_cpu
        return st

    def _start_local(self, z_obj, target_obs, cameras, ranking):
        engine = self._engine_for(z_obj, target_obs)
        if engine is not None:
            from ..engine import camera_params
            P = camera_params(cameras).detach().clone().contiguous()          # (N,10) master copy on the device
            P.requires_grad_(True)
            cam = cameras._like(log_quaternion=P[:, 0:3], translation=P[:, 3:6], viewport=P[:, 6:10])   # views of P
            return {
                'engine': engine, 'P': P, 'z_obj': z_obj, 'target': target_obs, 'cam': cam, 'params': [P],
                'opt': BatchedOptimizer(self.optimizer, [P]),
                'sched': _PlateauLR(len(cameras), self.learning_rate, self.lr_reduce_patience,
                                    self.lr_reduce_threshold, self.lr_reduce_factor),
                'ranking': [] if ranking is None else ranking, 'step': 0, 'converge_count': 0,
                'stat_history': {}, 'camera_history': [], 'target_q': target_obs.camera.quaternion,
                'template_cpu': cameras.to('cpu'),
            }
        cam = pu.parameterize_camera(cameras, optimize_viewport=True)     # batched leaves (N,3),(N,3),(N,4)
        params = [cam.log_quaternion, cam.translation, cam.viewport]
        return {
            'z_obj': z_obj, 'target': target_obs, 'cam': cam, 'params': params,
            'opt': BatchedOptimizer(self.optimizer, params),
            'sched': _PlateauLR(len(cameras), self.learning_rate, self.lr_reduce_patience, self.lr_reduce_threshold,
                                self.lr_reduce_factor),
            'ranking': [] if ranking is None else ranking, 'step': 0, 'converge_count': 0,
            'stat_history': {}, 'camera_history': [], 'target_q': target_obs.camera.quaternion,
        }

    def iterate(self, st):
        """One pose-optimisation iteration: render the N samples, loss, backward to the camera
        parameters, rank, optimiser + scheduler step.  Returns True when converged."""
        with self._frozen_model():
            return self._iterate_engine(st) if 'engine' in st else self._iterate_modules(st)

    def _iterate_modules(self, st):
        cam, params, target_obs, step = st['cam'], st['params'], st['target'], st['step']
        for p in params:
            p.grad = None
        z_target_latent = None
        if self.loss_weights.get('latent', 0.0) > 0.0:
            with torch.no_grad():
                z_target_latent = self.model.compute_latent_code(target_obs, cam)
        loss_dict, optim_loss, rank_loss, optim_weights = self.loss_and_grad(st['z_obj'], target_obs, cam, step,
                                                                           z_target_latent)
        shard = st.get('shard')
        if shard is not None:
            from .. import parallel
            rows = torch.cat((rank_loss.unsqueeze(1), cam.log_quaternion.detach(), cam.translation.detach()), dim=1)
            rows = parallel.gather_rows(rows.contiguous(), shard[2]).cpu()
            rank_all = rows[:, 0].tolist()                            # the one D2H sync per iteration
            rank_host = rank_all[shard[0]:shard[1]]
            detached_all = st['template_full_cpu']._like(log_quaternion=rows[:, 1:4].clone(), translation=rows[:, 4:7].clone(),
                                                         viewport=None)
            if self.return_camera_history:
                st['camera_history'].append((rows[:, 0].clone(), detached_all))
            delta = self._track_best_items(st['ranking'], step, list(detached_all), rank_all)
            detached = cam.uncrop().detach().clone()
        else:
            rank_host = rank_loss.tolist()                            # the one D2H sync per iteration
            detached = cam.uncrop().detach().clone()
            if self.return_camera_history:
                st['camera_history'].append((rank_loss.cpu(), detached.to('cpu')))
            delta = self._track_best_items(st['ranking'], step, list(detached.to('cpu')), rank_host)
        if self.track_stats:
            angle = three.quaternion.angular_distance(detached.quaternion, st['target_q']).squeeze()
            trans = torch.norm(detached.translation - target_obs.camera.translation, dim=1).squeeze()
            self._record_stat_dict(st['stat_history'], {
                **{f'{k}_loss': v.detach().cpu() for k, v in loss_dict.items()},
                **{f'{k}_weight': v for k, v in optim_weights.items()},
                'delta': delta, 'converge_count': st['converge_count'], 'angle_dist': angle.cpu(),
                'trans_dist': trans.cpu(), 'optim_loss': optim_loss.cpu(), 'rank_loss': rank_loss.cpu()})
        st['opt'].step(st['sched'].lr)
        st['sched'].step(rank_host)
        if delta < self.converge_threshold:
            st['converge_count'] += 1
        elif delta > self.converge_threshold:
            st['converge_count'] = 0
        st['step'] += 1
        return st['converge_count'] >= self.converge_patience

    def _engine_weights(self, st, step):
        optim_weights = copy.copy(self.loss_weights)
        if self.loss_schedules:
            optim_weights.update({k: v.get(step) for k, v in self.loss_schedules.items()})
            st['engine'].set_weights(optim_weights)
        return optim_weights

    def _engine_launch(self, st, step):
        """Enqueues forward + backward of iteration `step` and the asynchronous read-back of its losses and
        parameters (pinned buffer + event); nothing here waits for the GPU."""
        eng, P = st['engine'], st['P']
        optim_weights = self._engine_weights(st, step)
        with torch.no_grad():
            z_target_latent = None
            if optim_weights.get('latent', 0.0) != 0.0 or self.loss_weights.get('latent', 0.0) != 0.0:
                # the target's latent code under every hypothesis (reference :606-608): encoder + renderer, no gradient
                z_target_latent = self.model.compute_latent_code(st['target'], st['cam'])
            if self.engine_graph and z_target_latent is None and not self.loss_schedules and hasattr(eng, 'forward_backward_graph'):
                losses, gparams = eng.forward_backward_graph(st['cam'], P)
            else:
                losses, gparams = eng.forward_backward(st['cam'], need_grad=True, z_target_latent=z_target_latent, params=P)
            dev = torch.cat((losses[:, :6], P.detach()), dim=1)
            if st.get('shard') is not None:                           # (N_local,15) -> (N,15) over the ranks
                from .. import parallel
                dev = parallel.gather_rows(dev.contiguous(), st['shard'][2])
        slot = st.setdefault('pinned', {})
        key = step & 1
        if key not in slot or slot[key].shape != dev.shape:
            slot[key] = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        slot[key].copy_(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return {'step': step, 'gparams': gparams, 'host': slot[key], 'event': ev, 'weights': optim_weights}

    def _iterate_engine(self, st):
        """Same iteration on the fused engine, software-pipelined against the host: the optimiser step of
        iteration t needs only the learning rates decided after t-1, so it and the whole forward/backward of
        t+1 are enqueued BEFORE the host waits for iteration t's losses (ranking, plateau scheduler,
        convergence test).  Every quantity is computed exactly as in the sequential order; if the loop stops
        at t the already-enqueued forward/backward of t+1 is simply never read.  ONE device->host transfer
        (losses + parameters) per iteration."""
        eng, P, step, target_obs = st['engine'], st['P'], st['step'], st['target']
        cur = st.pop('inflight', None)
        if cur is None or cur['step'] != step:
            cur = self._engine_launch(st, step)
        P.grad = cur['gparams']
        st['opt'].step(st['sched'].lr)                                # lr after scheduler.step(loss[t-1])
        if st.get('max_steps') is None or step + 1 < st['max_steps']:
            st['inflight'] = self._engine_launch(st, step + 1)        # keeps the GPU busy while the host ranks
        cur['event'].synchronize()                                    # the one D2H sync per iteration
        host = cur['host'].clone()
        optim_weights = cur['weights']
        comp = {k: host[:, i] for i, k in enumerate(eng.LOSS_KEYS)}
        if self.loss_weights.get('latent', 0.0) != 0.0 or 'latent' in self.loss_schedules:
            comp['latent'] = host[:, 5]
        rank = sum(self.loss_weights.get(k, 0.0) * v for k, v in comp.items())
        rank_host = rank.tolist()
        shard = st.get('shard')
        tpl = st['template_cpu'] if shard is None else st['template_full_cpu']
        detached = tpl._like(log_quaternion=host[:, 6:9].clone(), translation=host[:, 9:12].clone(), viewport=None)
        if self.return_camera_history:
            st['camera_history'].append((rank.clone(), detached))
        delta = self._track_best_items(st['ranking'], step, list(detached), rank_host)
        if self.track_stats:
            angle = three.quaternion.angular_distance(detached.quaternion, st['target_q'].cpu()).squeeze()
            trans = torch.norm(detached.translation - target_obs.camera.translation.cpu(), dim=1).squeeze()
            self._record_stat_dict(st['stat_history'], {
                **{f'{k}_loss': v for k, v in comp.items()}, **{f'{k}_weight': v for k, v in optim_weights.items()},
                'delta': delta, 'converge_count': st['converge_count'], 'angle_dist': angle, 'trans_dist': trans,
                'optim_loss': host[:, 4].clone(), 'rank_loss': rank.clone()})
        st['sched'].step(rank_host if shard is None else rank_host[shard[0]:shard[1]])
        if delta < self.converge_threshold:
            st['converge_count'] += 1
        elif delta > self.converge_threshold:
            st['converge_count'] = 0
        st['step'] += 1
        return st['converge_count'] >= self.converge_patience

    def _optimize_camera(self, z_obj, target_obs, cameras, iters, ranking):
        st = self.start(z_obj, target_obs, cameras, ranking)
        st['max_steps'] = iters                                       # no speculative launch past the last iteration
        for _ in range(iters):
            if self.iterate(st):
                break
        return st['stat_history'], st['camera_history']

    @classmethod
    def _record_stat(cls, history, key, value):
        value = value.detach().cpu() if torch.is_tensor(value) else torch.tensor(value)
        value = value.squeeze().unsqueeze(0)
        if value.dim() > 2:
            for i in range(value.shape[-1]):
                cls._record_stat(history, f'{key}[{i}]', value[..., i])
        else:
            history[key] = torch.cat((history[key], value), dim=0) if key in history else value

    @classmethod
    def _record_stat_dict(cls, history, d):
        for k, v in d.items():
            cls._record_stat(history, k, v)

    def _render_observation(self, z_obj, camera, **kwargs):
        """The optimised camera IS the zoomed one; the un-thresholded depth goes to the loss
        (reference :703-713)."""
        pred, z_latent = self.model.render_latent_object(z_obj, camera.to(self.model.device), return_latent=True)
        z_depth = camera.denormalize_depth(pred['depth'].squeeze(0))
        return z_depth, pred['mask_logits'].squeeze(0), z_latent
