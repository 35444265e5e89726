"""Synthetic SYN(S,C) models and observations: the workload of bench.py and
of the large-size tests.  Everything is generated on the CPU generator (device independent)
and returned as plain data -- reference-format checkpoints and tensors -- so that the same
inputs can be fed to the HIP path and to the CPU oracle."""

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


# the released recipe's architecture (reference tools/train/train.sh:28-66, decoded in SURVEY appendix A10): 256^2 inputs,
# 16^3 x 256-channel latent volume, 512-channel U-Net levels, GRU fuser -- 68 M parameters
RELEASED_SCULPTOR = dict(in_size=256, image_config=[[64, 'D', 128, 'D', 196, 'D', 256, 'D', 512, 'D', 512, 'D', 512],
                                                    [512, 'U', 512, 'U', 256]],
                         camera_config=[64, 128, 256], object_config=[256, 256], projection_type='factor',
                         input_color=True, input_depth=False, input_mask=True, scale_mode='nearest')
RELEASED_PHOTOGRAPHER = dict(in_size=16, image_config=[[256, 'D', 512, 'D', 512],
                                                       [512, 'U', 512, 'U', 512, 'U', 256, 'U', 196, 'U', 128, 'U', 64]],
                             camera_config=[256, 256], object_config=[], projection_type='factor',
                             predict_depth=True, predict_mask=True, scale_mode='nearest')


def seeded_state_dict(template, seed, bias_std=0.0):
    """A state_dict with the keys / shapes of `template`, filled in SORTED KEY ORDER from one CPU generator: weights
    N(0,1) (He-equalised layers), biases N(0, bias_std).  Independent of the order in which a module's constructor
    happens to create its parameters, so any implementation with the same checkpoint keys -- this package, the CPU
    oracle, the reference itself (oracle/make_golden.py g25) -- gets the SAME network from a seed: 68 M parameters
    pinned by one integer."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(template):
        shape = tuple(template[k].shape)
        out[k] = torch.randn(shape, generator=gen) * (bias_std if k.endswith('bias') else 1.0)
    return out


def build_released_model(device='cuda', seed=0, bias_std=0.0):
    """The released ARCHITECTURE with random He-equalised weights (the public checkpoint is not obtainable offline):
    (LatentFusionModel on `device`, (sculptor, fuser, photographer checkpoints, camera_dist)).  The checkpoints are
    reference-format dicts, so the CPU oracle evaluates the same network (tests/test_fullshape_gpu.py, tools/rel_probe.py);
    the weights come from seeded_state_dict (seed, seed + 1, seed + 2 for sculptor, fuser, photographer)."""
    from .recon import fusion
    from .recon.inference import LatentFusionModel
    from .recon.models import Photographer, Sculptor
    sc = Sculptor(**RELEASED_SCULPTOR)
    ph = Photographer(**RELEASED_PHOTOGRAPHER)
    fu = fusion.get_fuser('gru', 256, 1.0)
    for i, m in enumerate((sc, fu, ph)):
        m.load_state_dict(seeded_state_dict(m.state_dict(), seed + i, bias_std))
    dist = optimal_camera_dist(consts.INTRINSIC[1][1], 256, 0.5, slack=0.5)
    cks = (sc.create_checkpoint(), fu.create_checkpoint(), ph.create_checkpoint(), dist)
    return LatentFusionModel(sc.eval(), fu.eval(), ph.eval(), dist, device), cks
