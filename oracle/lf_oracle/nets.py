"""ORACLE (test infrastructure, not product): the reconstruct / render networks on CPU.

Functional restatement of the reference's L1/L2 modules driven directly by reference-format
checkpoints ({'args': ..., 'state_dict': ...} with the reference's key names), so the same
weights can be fed to the reference, to this oracle and to the HIP product path.

  Block / Equalized / PixelNorm / Interpolate   modules/blocks.py:136-164, equalized.py:35-74,
                                                modules/__init__.py:8-36
  create_blocks grammar                         modules/blocks.py:10-75
  U-Net wiring                                  modules/unet.py:8-127
  lift / projection                             modules/geometry.py:693-749
  camera<->object resampling                    modules/geometry.py:593-690
  Sculptor / Photographer                       recon/models.py:84-505
  fusers, ConvGRU / ConvLSTM                    recon/fusion.py:17-246, modules/gru.py, lstm.py
"""
import math

import torch
import torch.nn.functional as F

SLOPE = 0.2


# ---------------------------------------------------------------------------------------------
# primitive layers
# ---------------------------------------------------------------------------------------------
def eq_conv(x, sd, prefix, padding=0):
    """He-equalised conv: conv(x, W) * sqrt(2/fan_in) + b  (equalized.py:57-74)."""
    w = sd[prefix + '.module.weight']
    he = math.sqrt(2.0 / w[0].numel())
    y = (F.conv3d if w.dim() == 5 else F.conv2d)(x, w, None, 1, padding)
    return y * he + sd[prefix + '.bias'].view(1, -1, *([1] * (w.dim() - 2)))


def pixel_norm(x):
    """x / sqrt(mean_c(x^2) + 1e-8)  (modules/__init__.py:14-15)."""
    return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def act_norm(x):
    return pixel_norm(F.leaky_relu(x, SLOPE))


def rescale(x, factor, mode):
    """modules/__init__.py:18-36; 'bilinear' becomes 'trilinear' for volumes (blocks.py:34-35)."""
    if factor == 1.0 or factor is None:
        return x
    if mode == 'bilinear' and x.dim() == 5:
        mode = 'trilinear'
    ac = False if mode in ('bilinear', 'trilinear') else None
    return F.interpolate(x, scale_factor=factor, mode=mode, align_corners=ac)


def block(x, sd, prefix, scale, mode):
    """conv-lrelu-pixelnorm twice, then optional rescale (blocks.py:152-164)."""
    x = act_norm(eq_conv(x, sd, prefix + '.conv1', 1))
    x = act_norm(eq_conv(x, sd, prefix + '.conv2', 1))
    return rescale(x, scale, mode)


def plan_blocks(config, default_scale, skip=False, skip_start=1, skip_end=None, in_views=1,
                skip_views=None):
    """Decodes the block mini-language into [(c_in, c_out, scale)] (blocks.py:10-75).

    Tokens 'I'/'U'/'D' set the rescale applied at the END of the next block."""
    if skip_views is None:
        skip_views = in_views
    n_blocks = sum(1 for b in config if isinstance(b, int)) - 1
    skip_end = n_blocks if skip_end is None else min(n_blocks, skip_end)
    plan, idx, pending, c_in = [], 0, 1.0, config[0]
    for tok in config[1:]:
        if isinstance(tok, int) or (isinstance(tok, str) and tok.isdigit()):
            extra = c_in * skip_views if (skip and skip_start <= idx < skip_end) else 0
            if idx == 0:
                c_in *= in_views
            plan.append((c_in + extra, int(tok), pending))
            c_in, idx, pending = int(tok), idx + 1, 1.0
        elif tok == 'I':
            pending = default_scale
        elif tok == 'U':
            pending = 2.0
        elif tok == 'D':
            pending = 0.5
        else:
            raise ValueError(f'unknown block token {tok!r}')
    return plan


def run_blocks(x, sd, prefix, plan, mode):
    mids = []
    for i, (_, _, scale) in enumerate(plan):
        x = block(x, sd, f'{prefix}.{i}', scale, mode)
        mids.append(x)
    return x, mids


def unet(x, sd, prefix, block_config, in_channels=None, out_channels=None, mode='bilinear'):
    """Generic U-Net (unet.py:95-127).  Up block i (1 <= i < n_down) concatenates the i-th
    deepest-first down output."""
    down_cfg, up_cfg = block_config
    n_down = sum(1 for b in down_cfg if isinstance(b, int)) - 1
    n_up = sum(1 for b in up_cfg if isinstance(b, int)) - 1
    down = plan_blocks(down_cfg, 0.5)
    up = plan_blocks(up_cfg, 2.0, skip=True, skip_end=min(n_down, n_up))
    if in_channels is not None:
        x = F.leaky_relu(eq_conv(x, sd, prefix + '.input_block.conv', 0), SLOPE)
    stack = []
    for i, (_, _, scale) in enumerate(down):
        x = block(x, sd, f'{prefix}.down_blocks.{i}', scale, mode)
        stack.insert(0, x)
    for i, (_, _, scale) in enumerate(up):
        if 1 <= i < len(stack):
            x = torch.cat((x, stack[i]), dim=1)
        x = block(x, sd, f'{prefix}.up_blocks.{i}', scale, mode)
    if isinstance(out_channels, int):
        x = eq_conv(x, sd, prefix + '.output_block.conv', 0)
    elif out_channels is not None:
        x = torch.cat([eq_conv(x, sd, f'{prefix}.output_block.{j}.conv', 0)
                       for j in range(len(out_channels))], dim=1)
    return x


def unet_sizes(block_config, in_size):
    nd = block_config[0].count('I') + block_config[0].count('D')
    nu = block_config[1].count('I') + block_config[1].count('U')
    bott = in_size // (2 ** nd)
    return bott, bott * (2 ** nu)


# ---------------------------------------------------------------------------------------------
# camera <-> object resampling (grid_sample, border, align_corners=False: quirks Q1-Q4)
# ---------------------------------------------------------------------------------------------
def _sample3d(vol, grid):
    return F.grid_sample(vol.float(), grid.float(), mode='bilinear', padding_mode='border',
                         align_corners=False)


def o2c_grid(cam, size, cube_size=1.0):
    """Sampling grid of ObjectToCameraTransform (geometry.py:469-493,515-531,669-686; row a2)."""
    lin = torch.linspace(0.0, 1.0, size)
    zp, vp, up = torch.meshgrid(lin, lin, lin, indexing='ij')
    B = len(cam)

    def e(t):
        return t.view(B, 1, 1, 1)
    u = up.unsqueeze(0) * e(cam.vw) + e(cam.viewport[:, 0])
    v = vp.unsqueeze(0) * e(cam.vh) + e(cam.viewport[:, 1])
    z = zp.unsqueeze(0) * cam.z_span + e(cam.znear)
    yc = (v - e(cam.v0)) / e(cam.fv) * z
    xc = (u - e(cam.u0)) / e(cam.fu) * z
    pc = torch.stack((xc, yc, z, torch.ones_like(z)), dim=-1).view(B, -1, 4)
    po = (cam.cam_to_obj @ pc.transpose(2, 1))[:, :3, :].transpose(1, 2)
    return (po / (cube_size / 2)).view(B, size, size, size, 3)


def o2c(obj_volume, cam, cube_size=1.0):
    size = obj_volume.size(-1)
    vol = obj_volume.expand(len(cam), -1, -1, -1, -1)
    return _sample3d(vol, o2c_grid(cam, size, cube_size))


def c2o_grid(cam, size, cube_size=1.0):
    """Sampling grid of CameraToObjectTransform (geometry.py:599-611,625-654; row a3).
    Note Q2: the z coordinate is mapped to [0,1], not [-1,1]."""
    lin = torch.linspace(-cube_size / 2, cube_size / 2, size)
    zc, yc, xc = torch.meshgrid(lin, lin, lin, indexing='ij')
    po = torch.stack((xc, yc, zc, torch.ones_like(xc)), dim=-1).view(-1, 4)
    B = len(cam)
    pc = cam.obj_to_cam @ po.t().unsqueeze(0).expand(B, -1, -1)
    pix = cam.K @ pc
    px = pix[:, 0] / pix[:, 2]
    py = pix[:, 1] / pix[:, 2]
    zn, zf = cam.znear.view(-1, 1), cam.zfar.view(-1, 1)
    g = torch.stack(((px - cam.viewport[:, 0, None]) / cam.vw[:, None] * 2 - 1,
                     (py - cam.viewport[:, 1, None]) / cam.vh[:, None] * 2 - 1,
                     (pix[:, 2] - zn) / (zf - zn)), dim=-1)
    return g.view(-1, size, size, size, 3)


def c2o(cam_volume, cam, cube_size=1.0):
    return _sample3d(cam_volume, c2o_grid(cam, cam_volume.size(-1), cube_size))


# ---------------------------------------------------------------------------------------------
# misc coordinate channels (recon/utils.py:35-61)
# ---------------------------------------------------------------------------------------------
def voxel_coords_zyx(ref):
    D, H, W = ref.shape[-3:]
    z, y, x = torch.meshgrid(torch.linspace(-1, 1, D), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W),
                             indexing='ij')
    return torch.stack((z, y, x), dim=0).unsqueeze(0).expand(ref.shape[0], -1, -1, -1, -1)


def voxel_depth(ref):
    B, _, D, H, W = ref.shape
    return torch.linspace(-1.0, 1.0, D).view(1, 1, D, 1, 1).expand(B, 1, D, H, W)


# ---------------------------------------------------------------------------------------------
# Sculptor (encoder) and fusers
# ---------------------------------------------------------------------------------------------
def sculptor_forward(ck, x, cam):
    """Sculptor.forward (recon/models.py:198-224). x: (V, Cin, S, S)."""
    a, sd = ck['args'], ck['state_dict']
    mode = a.get('scale_mode', 'bilinear')
    cube = a.get('cube_size', 1.0)
    cin = (3 if a.get('input_color', True) else 0) + (1 if a.get('input_mask', True) else 0) \
        + (1 if a.get('input_depth', False) else 0)
    z = unet(x, sd, 'image_encoder', a['image_config'], in_channels=cin)
    _, img_out = unet_sizes(a['image_config'], a['in_size'])
    c0 = a['camera_config'][0]
    z = act_norm(eq_conv(z, sd, 'projection_block.conv', 0))
    if a.get('projection_type', 'tile') == 'factor':                    # geometry.py:711-728
        z = z.view(z.size(0), c0, -1, z.size(-2), z.size(-1))
    else:                                                               # geometry.py:693-708
        z = z.unsqueeze(2).expand(-1, -1, img_out, -1, -1)
    cam_mid, obj_mid = [], []
    for i, (_, _, scale) in enumerate(plan_blocks(a['camera_config'], 0.5)):
        z = block(z, sd, f'camera_blocks.{i}', scale, mode)
        cam_mid.append(c2o(z, cam, cube))
    z = c2o(z, cam, cube)
    if a['object_config']:
        for i, (_, _, scale) in enumerate(plan_blocks(a['object_config'], 0.5)):
            z = block(z, sd, f'object_blocks.{i}', scale, mode)
            obj_mid.append(z)
    z = eq_conv(z, sd, 'output_block.conv', 0)
    act = a.get('cube_activation_type')
    if act == 'lrelu':
        z = F.leaky_relu(z, SLOPE)
    elif act == 'relu':
        z = F.relu(z)
    elif act == 'tanh':
        z = torch.tanh(z)
    return z, cam_mid, obj_mid


def gru_step(sd, prefix, x, h):
    """ConvGRUCell.forward (modules/gru.py:36-43) -- no tanh on the candidate (Q13)."""
    xin = torch.cat([x, h], dim=1)
    upd = torch.sigmoid(eq_conv(xin, sd, prefix + '.update_gate', 1))
    rst = torch.sigmoid(eq_conv(xin, sd, prefix + '.reset_gate', 1))
    cand = eq_conv(torch.cat([x, h * rst], dim=1), sd, prefix + '.out_gate', 1)
    return h * (1 - upd) + cand * upd


def lstm_step(sd, prefix, x, h, c):
    """ConvLSTMCell.forward (modules/lstm.py:41-56)."""
    cc = eq_conv(torch.cat([x, h], dim=1), sd, prefix + '.conv', 1)
    i, f, o, g = torch.split(cc, h.shape[1], dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def fuse(fck, z_obj, cam_mid=None, cam=None):
    """Fuser.forward for every fuser type (recon/fusion.py:77-246). z_obj: (B,V,C,S,S,S)."""
    kind = fck['type']
    sd = fck.get('state_dict')
    if kind == 'PoolFuser':
        pool = fck.get('pool_type', 'mean')
        if pool == 'mean':
            return z_obj.mean(dim=1, keepdim=True)
        if pool == 'max':
            return z_obj.max(dim=1, keepdim=True)[0]
        if pool == 'median':
            return z_obj.median(dim=1, keepdim=True)[0]                 # lower median (Q14)
        if pool == 'abs_max':                                           # functional.py:47-49
            idx = z_obj.abs().max(dim=1, keepdim=True)[1]
            return torch.gather(z_obj, 1, idx)
        raise ValueError(pool)
    if kind == 'ConcatFuser':
        B, V, C, D, H, W = z_obj.shape
        return z_obj.reshape(B, 1, V * C, D, H, W)
    if kind == 'GRUFuser':
        h = z_obj[:, 0]
        coords = voxel_coords_zyx(h)
        for i in range(1, z_obj.shape[1]):
            h = gru_step(sd, 'gru', torch.cat((z_obj[:, i], coords), dim=1), h)
        return h.unsqueeze(1)
    if kind == 'LSTMFuser':
        h = z_obj[:, 0]
        c = torch.zeros_like(h)
        coords = voxel_coords_zyx(h)
        for i in range(1, z_obj.shape[1]):
            h, c = lstm_step(sd, 'lstm', torch.cat((z_obj[:, i], coords), dim=1), h, c)
        return h.unsqueeze(1)
    if kind == 'BlendFuser':                                            # fusion.py:125-149
        a = fck['args']
        zc = cam_mid[-1]
        V = zc.shape[1]
        zc = zc.reshape(-1, *zc.shape[2:])
        w = unet(torch.cat((zc, voxel_depth(zc)), dim=1), sd, 'unet', a['block_config'],
                 in_channels=a['in_channels'] + 1, out_channels=1)
        w = c2o(w, cam, a.get('cube_size', 1.0))
        w = torch.softmax(w.view(-1, V, *w.shape[1:]), dim=1)
        return torch.sum(z_obj * w, dim=1, keepdim=True)
    raise ValueError(kind)


def encode(sck, fck, cam, color, depth=None, mask=None):
    """Sculptor.encode (recon/models.py:226-258) for one object: color (V,3,H,W) etc.
    Returns z_obj (1,1,C,S,S,S)."""
    a = sck['args']
    parts = []
    if a.get('input_color', True):
        parts.append(color)
    if a.get('input_depth', False):
        parts.append(depth)
    if a.get('input_mask', True):
        parts.append(mask * 2.0 - 1.0)                                  # gan_normalize
    x = torch.cat(parts, dim=1)
    z, cam_mid, _ = sculptor_forward(sck, x, cam)
    V = x.shape[0]
    z = z.view(1, V, *z.shape[1:])
    # BlendFuser consumes the *camera-space* mid volumes already resampled to object space.
    cam_mid = [m.view(1, V, *m.shape[1:]) for m in cam_mid]
    return fuse(fck, z, cam_mid, cam)


# ---------------------------------------------------------------------------------------------
# Photographer (decoder / "renderer")
# ---------------------------------------------------------------------------------------------
def photographer_forward(ck, z_obj, cam):
    """Photographer.forward without skip connections (recon/models.py:397-453).
    z_obj: (N,C,S,S,S) already expanded to len(cam).  Returns logits, 2-D latent, z_depth."""
    a, sd = ck['args'], ck['state_dict']
    if a.get('skip_connections', False):
        raise NotImplementedError('skip_connections are unused by every shipped recipe')
    if z_obj.shape[0] != len(cam):
        raise ValueError('batch dimension of z_obj and camera must match')
    mode = a.get('scale_mode', 'bilinear')
    cube = a.get('cube_size', 1.0)
    z = z_obj
    if a['object_config']:
        for i, (_, _, scale) in enumerate(plan_blocks(a['object_config'], 2.0, in_views=a.get('in_views', 1))):
            z = block(z, sd, f'object_blocks.{i}', scale, mode)
    z = o2c(z, cam, cube)
    for i, (_, _, scale) in enumerate(plan_blocks(a['camera_config'], 2.0)):
        z = block(z, sd, f'camera_blocks.{i}', scale, mode)
    z_depth = None
    if a.get('occlusion_config'):                                       # models.py:378-395,427-430
        logits = unet(torch.cat((z, voxel_depth(z)), dim=1), sd, 'occlusion_module',
                      a['occlusion_config'], in_channels=a['object_config'][-1] + 1, out_channels=1)
        w = torch.softmax(logits, dim=2)
        w_resized = torch.softmax(F.interpolate(logits, z.size(-1)), dim=2)
        z_depth = (voxel_depth(w) * w).sum(dim=2)
        z = z * w_resized
    proj = a.get('projection_type', 'sum')
    if proj == 'sum':
        z = z.sum(dim=2)
    elif proj == 'factor':                                              # geometry.py:731-749
        z = z.view(z.size(0), z.size(1) * z.size(2), z.size(3), z.size(4))
        z = act_norm(eq_conv(z, sd, 'projection_block.conv', 0))
    y = unet(z, sd, 'image_decoder', a['image_config'])
    heads = int(a.get('predict_color', False)) + int(a.get('predict_depth', True)) \
        + int(a.get('predict_mask', True))
    y = torch.cat([eq_conv(y, sd, f'output_blocks.{j}.conv', 0) for j in range(heads)], dim=1)
    return y, z, z_depth


def interpret_logits(ck, logits, apply_mask=False):
    """Photographer.interpret_logits (recon/models.py:455-484)."""
    a = ck['args']
    y, base = {}, 0
    if a.get('predict_color', False):
        y['color_logits'] = logits[:, base:base + 3]
        y['color'] = torch.tanh(y['color_logits'])
        base += 3
    if a.get('predict_depth', True):
        y['depth_logits'] = logits[:, base:base + 1]
        y['depth'] = torch.tanh(y['depth_logits'])
        base += 1
    if a.get('predict_mask', True):
        y['mask_logits'] = logits[:, base:base + 1]
        y['mask'] = torch.sigmoid(y['mask_logits'])
    else:
        y['mask'] = (y['depth'].detach() > -1.0).float()
        y['mask_logits'] = 100 * y['mask'] + (-100) * (1.0 - y['mask'])
    if apply_mask and a.get('predict_mask', True):
        if a.get('predict_depth', True):
            y['depth'] = (y['depth'] + 1) * (y['mask'] > 0.5) - 1
        if a.get('predict_color', False):
            y['color'] = y['color'] * (y['mask'] > 0.5)
    return y


def decode(ck, z_obj, cam, apply_mask=False):
    """Photographer.decode for ONE object (recon/models.py:486-505).
    z_obj: (1,1,C,S,S,S).  Returns dict of (1,N,.,H,W) and latent (1,N,C2,h,w)."""
    n = len(cam)
    z = z_obj.expand(-1, n, -1, -1, -1, -1).reshape(-1, *z_obj.shape[2:])
    logits, lat, z_depth = photographer_forward(ck, z, cam)
    y = interpret_logits(ck, logits, apply_mask)
    return {k: v.unsqueeze(0) for k, v in y.items()}, lat.unsqueeze(0), z_depth
