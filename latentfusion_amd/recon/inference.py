"""LatentFusionModel: the inference facade the pose estimators drive (API mirror of
latentfusion/recon/inference.py:14-149)."""
from pathlib import Path

import torch

from ..observation import Observation
from . import models


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

    def train(self, train):
        """eval() also freezes the parameters (requires_grad False): the pose estimators differentiate w.r.t. the
        camera only, and the reference's backward through this facade accumulates weight gradients nobody reads
         -- here those kernels are simply not launched.  train(True) re-enables them."""
        for m in (self.sculptor, self.photographer, self.fuser, self.generator):
            if m is not None:
                m.train(train)
                m.requires_grad_(bool(train))
        return self

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
