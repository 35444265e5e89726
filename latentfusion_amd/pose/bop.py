"""Camera-intrinsics JSON of the BOP tool-chain ({"fx", "fy", "cx", "cy"} -> 3 x 4 matrix;
latentfusion/pose/bop.py:6-19)."""
import json

import torch


def parse_camera_intrinsics(d):
    K = torch.zeros(3, 4, dtype=torch.float32)
    K[0, 0], K[1, 1], K[2, 2] = float(d['fx']), float(d['fy']), 1.0
    K[0, 2], K[1, 2] = float(d['cx']), float(d['cy'])
    return K


def load_camera_intrinsics(path):
    with open(path) as fh:
        return parse_camera_intrinsics(json.load(fh))
