"""Reader for scenes in the BOP layout (LINEMOD `lm` / `lmo`, T-LESS `tless`):
latentfusion/datasets/bop.py:49-236.  Pure host code: JSON camera/pose files, PNG colour / depth / visible-mask
images, object-scale normalisation (unit-diameter objects), evenly spread reference-view selection.

    dataset_path/models[_reconst]/obj_%06d.ply, dataset_path/models_eval/{obj_%06d.ply, models_info.json}
    scene_path/{rgb, depth, mask_visib}/%06d[_%06d].png, scene_path/{scene_camera.json, scene_gt.json}

Items are dictionaries {'color' (3,H,W) in [0,1], 'mask' (H,W) bool, 'depth' (H,W) in object units,
'extrinsic' (4,4), 'intrinsic' (3,4)} -- what `Observation.from_dict` consumes."""
import json
import struct
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import Dataset

from .. import three

_LINEMOD_NAMES = ('ape benchvise bowl camera can cat mug driller duck eggbox glue holepuncher iron lamp phone').split()
LINEMOD_ID_TO_NAME = {f'{i:06d}': name for i, name in enumerate(_LINEMOD_NAMES, start=1)}

# dataset directory name -> (scale that maps the model diameter to the working unit, folder of the meshes)
_LAYOUTS = {'lm': (1.0, 'models'), 'lmo': (1.0, 'models'), 'tless': (0.60, 'models_reconst')}


def read_ply_vertices(path):
    """(V,3) float32 vertex positions of an ASCII or binary-little-endian PLY (the reference goes through
    trimesh, meshutils.py; only x, y, z are used)."""
    with open(path, 'rb') as f:
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode('ascii', 'replace').strip()
            if line.startswith('format'):
                fmt = line.split()[1]
            elif line.startswith('element'):
                in_vertex = line.split()[1] == 'vertex'
                if in_vertex:
                    n_vert = int(line.split()[2])
            elif line.startswith('property') and in_vertex:
                props.append((line.split()[-1], line.split()[1]))
            elif line == 'end_header':
                break
        names = [p[0] for p in props]
        ix, iy, iz = names.index('x'), names.index('y'), names.index('z')
        if fmt == 'ascii':
            rows = [f.readline().split() for _ in range(n_vert)]
            return np.array([[float(r[ix]), float(r[iy]), float(r[iz])] for r in rows], dtype=np.float32)
        if fmt != 'binary_little_endian':
            raise ValueError(f'unsupported PLY format {fmt!r}')
        codes = {'float': 'f', 'float32': 'f', 'double': 'd', 'float64': 'd', 'uchar': 'B', 'uint8': 'B', 'char': 'b',
                 'int': 'i', 'int32': 'i', 'uint': 'I', 'uint32': 'I', 'short': 'h', 'ushort': 'H'}
        rec = struct.Struct('<' + ''.join(codes[t] for _, t in props))
        buf = f.read(rec.size * n_vert)
        out = np.empty((n_vert, 3), dtype=np.float32)
        for i in range(n_vert):
            r = rec.unpack_from(buf, i * rec.size)
            out[i] = (r[ix], r[iy], r[iz])
        return out


class BOPDataset(Dataset):
    def __init__(self, dataset_path, scene_path, object_id, center_object=False, object_scale=None):
        super().__init__()
        self.dataset_path, self.scene_path, self.object_id = Path(dataset_path), Path(scene_path), object_id
        if self.dataset_path.name not in _LAYOUTS:
            raise ValueError(f'Unknown dataset type {self.dataset_path.name}')
        base_obj_scale, mesh_dir = _LAYOUTS[self.dataset_path.name]
        self.models_path = self.dataset_path / mesh_dir
        self.model_path = self.models_path / f'obj_{self.object_id:06d}.ply'
        self.pointcloud_path = self.dataset_path / 'models_eval' / f'obj_{self.object_id:06d}.ply'
        with open(self.dataset_path / 'models_eval' / 'models_info.json', 'r') as f:
            self.model_info = json.load(f)[str(object_id)]
        self.center_object = center_object
        self.object_scale = base_obj_scale / self.model_info['diameter'] if object_scale is None else object_scale
        self.image_scale = 1.0
        self.bounds = torch.tensor([(self.model_info[f'min_{ax}'], self.model_info[f'min_{ax}'] + self.model_info[f'size_{ax}'])
                                    for ax in 'xyz'])
        self.centroid = self.bounds.mean(dim=1)
        self.depth_dir, self.mask_dir, self.color_dir = (self.scene_path / 'depth', self.scene_path / 'mask_visib',
                                                         self.scene_path / 'rgb')
        self.intrinsics, self.depth_scales = self.load_intrinsics(self.scene_path / 'scene_camera.json')
        self.extrinsics, self.scene_object_inds = self.load_extrinsics(self.scene_path / 'scene_gt.json')
        self.extrinsics = torch.stack(self.extrinsics, dim=0)
        rotation, _ = three.decompose(self.extrinsics)
        self.quaternions = three.quaternion.mat_to_quat(rotation[:, :3, :3])
        frames = self.scene_object_inds
        self.depth_paths = sorted(self.depth_dir / f'{fi:06d}.png' for fi in frames.keys())
        self.mask_paths = [self.mask_dir / f'{fi:06d}_{oi:06d}.png' for fi, oi in frames.items()]
        self.color_paths = sorted(self.color_dir / f'{fi:06d}.png' for fi in frames.keys())
        assert len(self.depth_paths) == len(self.mask_paths) == len(self.color_paths)

    def load_pointcloud(self):
        return torch.tensor(read_ply_vertices(self.pointcloud_path), dtype=torch.float32) * self.object_scale

    @classmethod
    def load_intrinsics(cls, path):
        intrinsics, depth_scales = [], []
        with open(path, 'r') as f:
            d = json.load(f)
        for key in sorted(int(k) for k in d.keys()):
            v = d[str(key)]
            intrinsics.append(three.intrinsic_to_3x4(torch.tensor(v['cam_K']).reshape(3, 3)).float())
            depth_scales.append(v['depth_scale'])
        return intrinsics, depth_scales

    def load_extrinsics(self, path):
        extrinsics, scene_object_inds = [], {}
        with open(path, 'r') as f:
            d = json.load(f)
        for frame_ind in sorted(int(k) for k in d.keys()):
            for obj_ind, cam_d in enumerate(d[str(frame_ind)]):
                if cam_d['obj_id'] == self.object_id:
                    rotation = torch.tensor(cam_d['cam_R_m2c'], dtype=torch.float32).reshape(3, 3)
                    translation = torch.tensor(cam_d['cam_t_m2c'], dtype=torch.float32)
                    extrinsics.append(three.to_extrinsic_matrix(translation, three.quaternion.mat_to_quat(rotation)))
                    scene_object_inds[frame_ind] = obj_ind
        return extrinsics, scene_object_inds

    def __len__(self):
        return len(self.color_paths)

    def get_ids(self):
        return [p.stem for p in self.color_paths]

    def _load(self, path, dtype):
        image = Image.open(path)
        image = image.resize((int(image.width * self.image_scale), int(image.height * self.image_scale)))
        return np.array(image, dtype=dtype) if dtype is not None else np.array(image)

    def _load_color(self, path):
        return self._load(path, None)

    def _load_mask(self, path):
        image = self._load(path, bool)
        return image[:, :, 0] if image.ndim > 2 else image

    def _load_depth(self, path):
        return self._load(path, np.float32)

    # dataset units <-> working units: optional re-centring on the model's bounding-box centre, then the
    # translation is scaled so that the object has unit diameter (x base scale); intrinsics follow image_scale
    def _recentre(self, extrinsic, sign):
        return three.translate_matrix(extrinsic, sign * self.centroid.to(extrinsic.device)) if self.center_object else extrinsic

    def normalize_extrinsic(self, extrinsic):
        out = self._recentre(extrinsic.clone(), -1.0)
        out[..., :3, 3] *= self.object_scale
        return out

    def denormalize_extrinsic(self, extrinsic):
        out = extrinsic.clone()
        out[..., :3, 3] /= self.object_scale
        return self._recentre(out, 1.0)

    def _scale_intrinsic(self, intrinsic, factor):
        out = intrinsic.clone()
        out[..., :2, :] *= factor
        return out

    def normalize_intrinsic(self, intrinsic):
        return self._scale_intrinsic(intrinsic, self.image_scale)

    def denormalize_intrinsic(self, intrinsic):
        return self._scale_intrinsic(intrinsic, 1.0 / self.image_scale)

    def sample_evenly(self, n):
        """Indices of `n` views whose camera positions are spread by farthest-point sampling."""
        positions = three.extrinsic_to_position(self.extrinsics)
        _, inds = three.utils.farthest_points(positions, n_clusters=n, dist_func=F.pairwise_distance,
                                              return_center_indexes=True)
        return inds

    def __getitem__(self, idx):
        color = (torch.tensor(self._load_color(self.color_paths[idx])).float() / 255.0).permute(2, 0, 1)
        mask = torch.tensor(self._load_mask(self.mask_paths[idx])).bool()
        depth = torch.tensor(self._load_depth(self.depth_paths[idx])) * self.object_scale * self.depth_scales[idx]
        return {'color': color, 'mask': mask, 'depth': depth, 'extrinsic': self.normalize_extrinsic(self.extrinsics[idx]),
                'intrinsic': self.normalize_intrinsic(self.intrinsics[idx])}
