"""Synthetic SYN(S,C) models and observations: the workload of bench.py and
of the large-size tests.  Everything is generated on the CPU generator (device independent)
and returned as plain data -- reference-format checkpoints and tensors -- so that the same
inputs can be fed to the HIP path and to the CPU oracle."""
import math

import torch

from . import consts
from .recon.utils import optimal_camera_dist

IMAGE_CONFIG = [[16, 32], [32, 16]]


def _conv_entries(sd, prefix, cout, cin, k, dims, gen, bias_std):
    sd[prefix + '.module.weight'] = torch.randn(cout, cin, *([k] * dims), generator=gen)
    sd[prefix + '.bias'] = torch.randn(cout, generator=gen) * bias_std if bias_std else torch.zeros(cout)


def _block_entries(sd, prefix, cin, cout, dims, gen, bias_std):
    _conv_entries(sd, prefix + '.conv1', cout, cin, 3, dims, gen, bias_std)
    _conv_entries(sd, prefix + '.conv2', cout, cout, 3, dims, gen, bias_std)


def _unet2d_entries(sd, prefix, in_channels, gen, bias_std):
    (d0, d1), (u0, u1) = IMAGE_CONFIG
    if in_channels is not None:
        _conv_entries(sd, prefix + '.input_block.conv', d0, in_channels, 1, 2, gen, bias_std)
    _block_entries(sd, prefix + '.down_blocks.0', d0, d1, 2, gen, bias_std)
    _block_entries(sd, prefix + '.up_blocks.0', u0, u1, 2, gen, bias_std)


def make_syn_checkpoints(S, C, fuser='gru', seed=0, bias_std=0.0):
    """Reference-format checkpoints of the SYN(S,C) family: He-equalised N(0,1) weights, zero bias
    (bias_std > 0 perturbs the biases for tests).  Returns (sculptor, fuser, photographer, camera_dist)."""
    gen = torch.Generator().manual_seed(seed)
    ssd = {}
    _unet2d_entries(ssd, 'image_encoder', 4, gen, bias_std)
    _conv_entries(ssd, 'projection_block.conv', C * S, IMAGE_CONFIG[1][1], 1, 2, gen, bias_std)
    _block_entries(ssd, 'camera_blocks.0', C, C, 3, gen, bias_std)
    _block_entries(ssd, 'object_blocks.0', C, C, 3, gen, bias_std)
    _conv_entries(ssd, 'output_block.conv', C, C, 1, 3, gen, bias_std)
    sculptor = {'args': dict(in_size=S, image_config=IMAGE_CONFIG, camera_config=[C, C], object_config=[C, C],
                             projection_type='factor', input_color=True, input_depth=False, input_mask=True,
                             scale_mode='nearest', cube_size=1.0, relu_slope=0.2, cube_activation_type=None),
                'state_dict': ssd}
    psd = {}
    _block_entries(psd, 'camera_blocks.0', C, C, 3, gen, bias_std)
    _conv_entries(psd, 'projection_block.conv', IMAGE_CONFIG[0][0], C * S, 1, 2, gen, bias_std)
    _unet2d_entries(psd, 'image_decoder', None, gen, bias_std)
    _conv_entries(psd, 'output_blocks.0.conv', 1, IMAGE_CONFIG[1][1], 1, 2, gen, bias_std)
    _conv_entries(psd, 'output_blocks.1.conv', 1, IMAGE_CONFIG[1][1], 1, 2, gen, bias_std)
    photographer = {'args': dict(in_size=S, image_config=IMAGE_CONFIG, camera_config=[C, C], object_config=[],
                                 projection_type='factor', predict_color=False, predict_depth=True, predict_mask=True,
                                 scale_mode='nearest', cube_size=1.0, occlusion_config=False, in_views=1,
                                 skip_connections=False, relu_slope=0.2),
                    'state_dict': psd}
    if fuser == 'gru':
        fsd = {}
        for gate in ('update_gate', 'reset_gate', 'out_gate'):
            _conv_entries(fsd, 'gru.' + gate, C, 2 * C + 3, 3, 3, gen, bias_std)
        fck = {'type': 'GRUFuser', 'args': {'in_channels': C, 'cube_size': 1.0}, 'state_dict': fsd}
    elif fuser.startswith('pool:'):
        fck = {'type': 'PoolFuser', 'pool_type': fuser.split(':')[1]}
    else:
        raise ValueError(fuser)
    dist = optimal_camera_dist(consts.INTRINSIC[1][1], S, 0.5, slack=128 / S)
    return sculptor, fck, photographer, dist


def make_observation_data(V, seed, height=480, width=640):
    """Raw tensors of a synthetic observation: evenly distributed orientations (global RNG under
    `seed`), t = (0,0,1), uniform colour, disc mask of radius 150 px at (315,251), depth 1..1.1."""
    from . import three
    torch.manual_seed(seed)
    q = three.orientation.evenly_distributed_quats(V)
    t = torch.tensor([[0.0, 0.0, 1.0]]).expand(V, -1)
    extrinsic = three.to_extrinsic_matrix(t, q)
    K = torch.tensor(consts.INTRINSIC).unsqueeze(0).expand(V, -1, -1).clone()
    color = torch.rand(V, 3, height, width)
    yy, xx = torch.meshgrid(torch.arange(float(height)), torch.arange(float(width)), indexing='ij')
    disc = (((xx - 315) ** 2 + (yy - 251) ** 2) <= 150 ** 2).float()
    mask = disc.view(1, 1, height, width).expand(V, -1, -1, -1).clone()
    depth = (1.0 + 0.1 * torch.rand(V, 1, height, width)) * mask
    return {'color': color, 'depth': depth, 'mask': mask, 'intrinsic': K, 'extrinsic': extrinsic,
            'width': width, 'height': height}


def make_observation(V, seed, device='cpu'):
    from .modules.geometry import Camera
    from .observation import Observation
    d = make_observation_data(V, seed)
    cam = Camera(d['intrinsic'], d['extrinsic'], width=d['width'], height=d['height'])
    return Observation(d['color'], d['depth'], d['mask'], cam).to(device)


def build_model(S, C, fuser='gru', seed=0, device='cuda', bias_std=0.0):
    """LatentFusionModel over the SYN(S,C) checkpoints (HIP path)."""
    from .recon import fusion
    from .recon.inference import LatentFusionModel
    from .recon.models import Photographer, Sculptor
    sck, fck, pck, dist = make_syn_checkpoints(S, C, fuser, seed, bias_std)
    model = LatentFusionModel(Sculptor.from_checkpoint(sck), fusion.from_checkpoint(fck),
                              Photographer.from_checkpoint(pck), dist, device)
    return model, (sck, fck, pck, dist)
