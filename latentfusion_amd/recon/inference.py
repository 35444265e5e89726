"""LatentFusionModel: the inference facade the pose estimators drive (API mirror of
latentfusion/recon/inference.py:14-149)."""
from pathlib import Path

import torch

from ..observation import Observation
from . import models


class _Frozen:
    def __init__(self, model):
        self.params = None
        self.model = model

    def __enter__(self):
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        for p in self.params:
            p.requires_grad_(False)
        return self.model

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)
        return False


class LatentFusionModel:
    @classmethod
    def from_checkpoint(cls, checkpoint, device='cpu'):
        if isinstance(checkpoint, (Path, str)):
            checkpoint = torch.load(checkpoint, map_location='cpu', weights_only=False)
        sculptor, fuser, photographer, _, generator = models.load_models(checkpoint, device=device, return_generator=True)
        return cls(sculptor, fuser, photographer, checkpoint['args']['camera_dist'], device, generator=generator)

    def __init__(self, sculptor, fuser, photographer, camera_dist, device, generator=None):
        self.device = device
        self.sculptor, self.fuser, self.photographer = sculptor.to(device), fuser.to(device), photographer.to(device)
        self.generator = generator.to(device) if generator is not None else None
        self.camera_dist = camera_dist
        self.input_size = sculptor.in_size
        self.eval()

    def eval(self):
        return self.train(False)

    def train(self, train=True):
        """Switches the modules' train / eval mode, like the reference (recon/inference.py:39-49).  Whether parameters
        take gradients is a separate matter: see freeze() / frozen()."""
        for m in self._modules():
            m.train(train)
        return self

    def _modules(self):
        return [m for m in (self.sculptor, self.photographer, self.fuser, self.generator) if m is not None]

    def parameters(self):
        for m in self._modules():
            yield from m.parameters()

    def freeze(self):
        """requires_grad False on every parameter: no weight-gradient kernel is launched by a backward through this
        model (the reference's pose loops accumulate weight gradients nobody reads)."""
        for p in self.parameters():
            p.requires_grad_(False)
        return self

    def unfreeze(self):
        for p in self.parameters():
            p.requires_grad_(True)
        return self

    def frozen(self):
        """Context manager: parameters frozen inside, every parameter's own flag restored on exit.  The pose
        estimators run under it, so a model that is also being trained keeps its flags."""
        return _Frozen(self)

    def zoom_observation(self, observation):
        if not observation.meta['is_zoomed']:
            return observation.zoom(self.camera_dist, self.input_size)
        return observation

    def preprocess_observation(self, observation):
        if not observation.meta['is_zoomed']:
            observation = observation.zoom(self.camera_dist, self.input_size)
        if not observation.meta['is_prepared']:
            observation = observation.prepare()
        if not observation.meta['is_normalized']:
            observation = observation.normalize()
        return observation

    def build_latent_object(self, observation: Observation):
        """Reference views -> fused latent volume (1,1,C,S,S,S)."""
        observation = self.preprocess_observation(observation.to(self.device))
        with torch.no_grad():
            z_obj, _ = self.sculptor.encode(self.fuser, camera=observation.camera, color=observation.color.unsqueeze(0),
                                            depth=observation.depth.unsqueeze(0), mask=observation.mask.unsqueeze(0))
        return z_obj

    def compute_latent_code(self, observation, camera):
        observation = self.preprocess_observation(observation)
        if len(observation) == 1:
            observation = observation.expand(len(camera))
        _, feats = models.autoencode(self.sculptor, self.fuser, self.photographer, camera=camera,
                                     color=observation.color.unsqueeze(1), depth=observation.depth.unsqueeze(1),
                                     mask=observation.mask.unsqueeze(1))
        return feats

    def render_latent_object(self, z_obj, camera, return_latent=True, apply_mask=True):
        y, z, _ = self.photographer.decode(z_obj, camera, return_latent=return_latent, apply_mask=apply_mask)
        if return_latent:
            z = z.squeeze(0)
        return y, z

    def render_ibr_basic(self, z_obj, input_obs, camera_out, return_latent=True, apply_mask=True, p=0.5):
        from .. import ibr
        from ..three.batchview import b2bv
        input_obs = self.preprocess_observation(input_obs)
        y, z = ibr.render_latent_ibr2(self.photographer, z_obj, input_obs.camera.clone().to(self.device),
                                      camera_out.clone().to(self.device),
                                      b2bv(input_obs.color, batch_size=1).to(self.device), p=p, weight_type='cam_dist',
                                      return_latent=return_latent, apply_mask=apply_mask)
        if return_latent:
            z = z.squeeze(0)
        return {k: v.squeeze(0) for k, v in y.items()}, z

    def render_ibr(self, z_obj, input_obs, camera_out, return_latent=True):
        """Colour by the learned IBR generator (reference recon/inference.py:151-191): the reprojected input views,
        their reprojected depths and a camera-similarity plane per view are stacked into the generator U-Net, whose
        logits blend the views per pixel after a residual flow of at most 5 pixels (ibr.warp_blend_logits)."""
        from .. import ibr
        if self.generator is None:
            raise ValueError('this checkpoint carries no IBR generator (modules.generator)')
        input_obs = self.preprocess_observation(input_obs)
        y_out, z_out, image_reproj, depth_reproj, _mask_out, depth_out, _dist_r, dist_t = self._render_reprojections(
            z_obj, input_obs.color, input_obs.camera, camera_out)
        if return_latent:
            z_out = z_out.squeeze(0)
        sims = 1.0 - dist_t * 2
        x = torch.cat((image_reproj, depth_reproj,
                       sims[:, :, None, None, None].expand(-1, -1, -1, *image_reproj.shape[-2:])), dim=2)
        x = x.reshape(-1, x.shape[1] * x.shape[2], x.shape[3], x.shape[4])              # views folded into channels
        x = torch.cat((depth_out, x), dim=1)
        logits = self.generator(x)
        y_out['color'], _, _, _ = ibr.warp_blend_logits(logits, image_reproj, 5)
        return {k: v.squeeze(0) for k, v in y_out.items()}, z_out

    def _render_reprojections(self, z_obj, color_in, camera_in, camera_out, return_latent=True):
        """reference recon/inference.py:193-217: depths of the input and output views from the latent renderer, the
        input views reprojected into every output view and masked by the predicted output mask."""
        from .. import ibr
        from ..three.batchview import bv2b
        y_in, _, _ = self.photographer.decode(z_obj, camera_in)
        y_out, z_out, _ = self.photographer.decode(z_obj, camera_out, return_latent=return_latent)
        mask_out, depth_out = y_out['mask'], y_out['depth']
        image_reproj, depth_reproj, dist_r, dist_t = ibr.reproject_views_batch(color_in.unsqueeze(0), y_in['depth'],
                                                                                y_out['depth'], camera_in, camera_out)
        image_reproj = image_reproj * mask_out.unsqueeze(2)
        depth_reproj = (depth_reproj + 1.0) * mask_out.unsqueeze(2) - 1.0
        return (y_out, z_out, bv2b(image_reproj), bv2b(depth_reproj), bv2b(mask_out), bv2b(depth_out), bv2b(dist_r),
                bv2b(dist_t))

    def render_full(self, z_obj, camera, input_obs=None, p=0.5):
        """Full-frame depth/mask(/colour).  Note SURVEY Q6: in the reference the input_obs=None branch
        raises (a 5-D tensor reaches uncrop); here that branch squeezes the object axis and works."""
        camera_zoom = camera.zoom(None, self.camera_dist, self.input_size).to(self.device)
        if input_obs is None:
            pred, _ = self.render_latent_object(z_obj, camera_zoom, apply_mask=True, return_latent=False)
            pred = {k: v.squeeze(0) for k, v in pred.items()}
        else:
            pred, _ = self.render_ibr_basic(z_obj, input_obs, camera_zoom, apply_mask=True, return_latent=False, p=p)
        out = {}
        mask = pred['mask']
        out['depth'], _ = camera_zoom.uncrop(camera_zoom.denormalize_depth(pred['depth']) * mask)
        out['mask'], _ = camera_zoom.uncrop(mask)
        if 'color' in pred:
            out['color'], _ = camera_zoom.uncrop(pred['color'] / 2 + 0.5)
        return out
