"""Observation container (API mirror of latentfusion/observation.py:71-163,225-290):
color (B,3,H,W) in [0,1], depth (B,1,H,W), mask (B,1,H,W) in {0,1}, a Camera, and the meta flags
is_zoomed / is_prepared / is_normalized that LatentFusionModel.preprocess_observation keys on.
Dataset / disk IO of the reference class is out of scope."""
import copy

import torch

from .modules.geometry import Camera


def gan_normalize(t):
    return t * 2.0 - 1.0


def gan_denormalize(t):
    return ((t + 1.0) / 2.0).clamp(0, 1)


class Observation:
    def __init__(self, color, depth, mask, camera, **kwargs):
        self.color = color.unsqueeze(0) if color is not None and color.dim() == 3 else color
        self.depth = depth.unsqueeze(0) if depth.dim() == 3 else depth
        self.mask = mask.unsqueeze(0) if mask.dim() == 3 else mask
        self.camera = camera
        self.meta = {'object_scale': kwargs.get('object_scale', 1.0), 'is_zoomed': kwargs.get('is_zoomed', False),
                     'is_normalized': kwargs.get('is_normalized', False), 'is_prepared': kwargs.get('is_prepared', False)}

    @classmethod
    def from_dataset(cls, dataset, inds=None):
        """Batch of the dataset items `inds` (all by default) as one Observation (reference :73-79)."""
        inds = range(len(dataset)) if inds is None else [int(i) for i in inds]
        items = [dataset[i] for i in inds]
        return cls.from_dict({k: torch.stack([it[k] for it in items], dim=0) for k in items[0]})

    @classmethod
    def from_dict(cls, d):
        h, w = d['color'].shape[-2:]
        return cls(d['color'], d['depth'].unsqueeze(-3), d['mask'].unsqueeze(-3).float(),
                   Camera(d['intrinsic'], d['extrinsic'], width=w, height=h))

    @property
    def device(self):
        return self.depth.device

    def __len__(self):
        return len(self.camera)

    def _new(self, color, depth, mask, camera, **over):
        meta = copy.deepcopy(self.meta)
        meta.update(over)
        return Observation(color, depth, mask, camera, **meta)

    def __getitem__(self, item):
        if isinstance(item, int):
            item = slice(item, item + 1)
        return self._new(self.color[item], self.depth[item], self.mask[item], self.camera[item])

    def clone(self):
        return self._new(self.color.clone(), self.depth.clone(), self.mask.clone(), self.camera.clone())

    @classmethod
    def collate(cls, observations):
        return cls(torch.cat([o.color for o in observations]), torch.cat([o.depth for o in observations]),
                   torch.cat([o.mask for o in observations]), Camera.cat([o.camera for o in observations]),
                   **observations[0].meta)

    def to_list(self):
        return [self[i] for i in range(len(self))]

    def to(self, device):
        return self._new(self.color.to(device) if self.color is not None else None, self.depth.to(device),
                         self.mask.to(device), self.camera.clone().to(device))

    def expand(self, n):
        if len(self) > 1:
            raise ValueError(f'Must be single but has batch size {len(self)}.')
        return self._new(self.color.expand(n, -1, -1, -1), self.depth.expand(n, -1, -1, -1),
                         self.mask.expand(n, -1, -1, -1), self.camera.repeat(n))

    def zoom(self, target_dist, target_size, camera: Camera = None):
        """Crop every map to the canonical zoomed viewport (reference :225-236; argument order of
        Camera.zoom is (size, dist) but the box is symmetric in the two)."""
        camera = self.camera if camera is None else camera
        color, new_camera = camera.zoom(self.color, target_size, target_dist, scale_mode='bilinear')
        depth, _ = camera.zoom(self.depth, target_size, target_dist, scale_mode='nearest')
        mask, _ = camera.zoom(self.mask, target_size, target_dist, scale_mode='nearest')
        return self._new(color, depth, mask, new_camera, is_zoomed=True)

    # ---- disk format of the reference (observation.py:164-223): cameras.json + %04d.{color,depth,mask}.png ----
    def save(self, path):
        """PNG colour (8-bit), depth (16-bit millimetres of the stored unit) and mask per view + cameras.json."""
        import json
        from pathlib import Path

        import numpy as np
        from PIL import Image

        path = Path(path)
        path.mkdir(exist_ok=True, parents=True)
        cam = self.camera.to('cpu')
        camera_json = cam.to_kwargs()
        camera_json['meta'] = self.meta
        with open(path / 'cameras.json', 'w') as f:
            def encode(o):
                # reference utils.MyEncoder: paths as strings, arrays as lists, anything else is an error (a silently
                # stringified object would not survive the round trip through Observation.load)
                from pathlib import PurePath
                if isinstance(o, PurePath):
                    return str(o)
                if torch.is_tensor(o) or isinstance(o, np.ndarray):
                    return o.tolist()
                raise TypeError(f'Object of type {type(o).__name__} is not JSON serializable')
            json.dump(camera_json, f, indent=2, default=encode)
        color, depth, mask = self.color.cpu(), self.depth.cpu(), self.mask.cpu()
        for i in range(len(self)):
            Image.fromarray((255.0 * color[i].permute(1, 2, 0).numpy()).astype(np.uint8)).save(path / f'{i:04d}.color.png')
            Image.fromarray((1000.0 * depth[i][0]).numpy().astype(np.uint16)).save(path / f'{i:04d}.depth.png')
            Image.fromarray(mask[i][0].numpy().astype(np.uint8) * 255).save(path / f'{i:04d}.mask.png')

    @classmethod
    def load(cls, path, frames=None) -> 'Observation':
        import json
        from pathlib import Path

        import numpy as np
        from PIL import Image
        path = Path(path)
        with open(path / 'cameras.json', 'r') as f:
            camera_json = json.load(f)
        meta = camera_json.pop('meta', {})
        cameras = Camera(**{k: torch.tensor(v, dtype=torch.float32) if isinstance(v, list) else v
                            for k, v in camera_json.items()})
        inds = list(range(len(cameras))) if frames is None else ([frames] if isinstance(frames, int) else list(frames))
        cameras = cameras[inds]
        color = torch.stack([torch.tensor(np.array(Image.open(path / f'{i:04d}.color.png')).astype(np.float32) / 255.0)
                             .permute(2, 0, 1) for i in inds], dim=0)
        depth = torch.stack([torch.tensor(np.array(Image.open(path / f'{i:04d}.depth.png')).astype(np.float32) / 1000.0)
                             .unsqueeze(0) for i in inds], dim=0)
        mask = torch.stack([torch.tensor(np.array(Image.open(path / f'{i:04d}.mask.png')).astype(bool)).float().unsqueeze(0)
                            for i in inds], dim=0)
        return cls(color, depth, mask, cameras, **meta)

    def estimate_camera(self) -> Camera:
        """Camera with the translation estimated from depth + mask (reference :284-287)."""
        from .pose.initialization import estimate_initial_pose
        return estimate_initial_pose(self.depth, self.mask, self.camera.intrinsic, self.camera.width,
                                     self.camera.height).to(self.device)

    def zoom_estimate(self, target_dist, target_size):
        return self.zoom(target_dist, target_size, camera=self.estimate_camera())

    def uncrop(self, camera=None):
        camera = self.camera if camera is None else camera
        color, new_camera = camera.uncrop(self.color, scale_mode='bilinear')
        depth, _ = camera.uncrop(self.depth, scale_mode='nearest')
        mask, _ = camera.uncrop(self.mask, scale_mode='nearest')
        return self._new(color, depth, mask, new_camera, is_zoomed=False)

    def prepare(self, crop_color=True, crop_depth=True):
        color = self.color
        if crop_color and color is not None:
            color = gan_denormalize(gan_normalize(color) * self.mask)
        depth = self.depth * self.mask if crop_depth else self.depth
        return self._new(color, depth, self.mask.clone(), self.camera.clone(), is_prepared=True)

    def normalize(self):
        return self._new(gan_normalize(self.color), self.camera.normalize_depth(self.depth), self.mask.clone(),
                         self.camera.clone(), is_normalized=True)

    def denormalize(self):
        return self._new(gan_denormalize(self.color), self.camera.denormalize_depth(self.depth), self.mask.clone(),
                         self.camera.clone(), is_normalized=False)
