"""Sculptor (2-D -> 3-D encoder) and Photographer (3-D -> 2-D "renderer") built on the HIP ops.

API mirror of latentfusion/recon/models.py: same constructor arguments, checkpoint format
({'args', 'state_dict'} with identical key names), forward/encode/decode signatures and output
dictionaries, so `LatentFusionModel.from_checkpoint` loads released weights unchanged.
"""
import torch
from torch import nn

from ..modules import EqualizedConv3d, unet
from ..modules.blocks import OutputBlock2d, OutputBlock3d, create_blocks
from ..modules.geometry import (Camera, CameraToObjectTransform, FactorProjection2d3d, FactorProjection3d2d,
                                ObjectToCameraTransform, TileProjection2d3d)
from ..three.batchview import b2bv, bv2b
from .. import ops
from . import fusion
from .utils import get_normalized_voxel_depth


def gan_normalize(t):
    return t * 2.0 - 1.0


def _get_activation(kind, relu_slope=0.2):
    if kind is None or kind == 'none':
        return None
    table = {'lrelu': lambda: nn.LeakyReLU(relu_slope), 'relu': nn.ReLU, 'tanh': nn.Tanh}
    if kind not in table:
        raise ValueError(f'Unknown activation type {kind}')
    return table[kind]()


def load_models(checkpoint, kwargs=None, device=None, return_generator=False):
    """Rebuilds (sculptor, fuser, photographer, discriminator[, generator]) from a training
    checkpoint (reference :32-70).  The discriminator is training-only (its module is out of scope) and is
    returned as None; the IBR generator (a UNet2d) is rebuilt when the checkpoint carries one."""
    if kwargs is None:
        kwargs = checkpoint['args']
    sck = checkpoint['modules']['sculptor']
    for key, flag in (('input_color', None), ('input_depth', 'generator_input_depth'), ('input_mask', 'generator_input_mask')):
        if key not in sck['args']:                      # legacy checkpoints
            sck['args'][key] = True if flag is None else kwargs[flag]
    pck = checkpoint['modules']['photographer']
    for key in ('predict_color', 'predict_depth', 'predict_mask'):
        if key not in pck['args']:
            pck['args'][key] = kwargs[key]
    sculptor = Sculptor.from_checkpoint(sck).to(device)
    photographer = Photographer.from_checkpoint(pck).to(device)
    fuser = fusion.from_checkpoint(checkpoint['modules']['fuser']).to(device)
    if return_generator:
        generator = None
        if 'generator' in checkpoint.get('modules', {}):                    # reference :63-66
            generator = unet.UNet2d.from_checkpoint(checkpoint['modules']['generator']).to(device)
        return sculptor, fuser, photographer, None, generator
    return sculptor, fuser, photographer, None


def autoencode(sculptor, fuser, photographer, camera, color, depth=None, mask=None):
    z_obj, _ = sculptor.encode(fuser, camera, color, depth, mask)
    y, z_pix, _ = photographer.decode(z_obj, camera, return_latent=True, interpret_logits=True)
    return {k: v.squeeze(1) for k, v in y.items()}, z_pix.squeeze(1)


class _Checkpointed(nn.Module):
    @classmethod
    def from_checkpoint(cls, checkpoint):
        model = cls(**checkpoint['args'])
        model.load_state_dict(checkpoint['state_dict'])
        return model

    def create_checkpoint(self):
        return {'args': dict(self._ctor_args), 'state_dict': {k: v.detach().cpu() for k, v in self.state_dict().items()}}


class Sculptor(_Checkpointed):
    def __init__(self, in_size, image_config, camera_config, object_config, relu_slope=0.2, cube_size=1.0,
                 cube_activation_type=None, projection_type='tile', input_color=True, input_depth=False,
                 input_mask=True, scale_mode='bilinear', **kwargs):
        super().__init__()
        self.image_config, self.camera_config, self.object_config = image_config, camera_config, object_config
        self.input_color, self.input_depth, self.input_mask = input_color, input_depth, input_mask
        self.relu_slope, self.cube_size, self.cube_activation_type = relu_slope, cube_size, cube_activation_type
        self.projection_type, self.scale_mode, self.in_size = projection_type, scale_mode, in_size
        self.in_channels = (3 if input_color else 0) + (1 if input_mask else 0) + (1 if input_depth else 0)
        self._ctor_args = dict(in_channels=self.in_channels, in_size=in_size, image_config=image_config,
                               camera_config=camera_config, object_config=object_config, relu_slope=relu_slope,
                               cube_size=cube_size, cube_activation_type=cube_activation_type,
                               projection_type=projection_type, input_color=input_color, input_depth=input_depth,
                               input_mask=input_mask, scale_mode=scale_mode)

        self.image_encoder = unet.UNet2d(self.in_channels, None, image_config)
        proj = {'tile': TileProjection2d3d, 'factor': FactorProjection2d3d}.get(projection_type)
        if proj is None:
            raise ValueError(f'Unknown projection type {projection_type!r}')
        self.projection_block = proj(in_channels=self.image_encoder.out_channels, out_channels=camera_config[0],
                                     out_size=self.image_out_size)
        self.camera_blocks = create_blocks(camera_config, EqualizedConv3d, 0.5, scale_mode=scale_mode)
        self.transform_block = CameraToObjectTransform(cube_size)
        self.object_blocks = (create_blocks(object_config, EqualizedConv3d, 0.5, scale_mode=scale_mode)
                              if object_config else nn.ModuleList())
        self.output_block = OutputBlock3d(self.out_channels, self.out_channels,
                                          activation=_get_activation(cube_activation_type))

    @property
    def image_out_size(self):
        return self.image_encoder.output_size(self.in_size)

    @property
    def image_bottleneck_size(self):
        return self.image_encoder.bottleneck_size(self.in_size)

    @property
    def camera_out_size(self):
        return self.image_out_size // (2 ** self.camera_config.count('D'))

    @property
    def out_size(self):
        return self.camera_out_size // (2 ** self.object_config.count('D')) if self.object_config else self.camera_out_size

    @property
    def out_channels(self):
        return self.object_config[-1] if self.object_config else self.camera_config[-1]

    def forward(self, x, camera: Camera):
        z = self.projection_block(self.image_encoder(x))
        z_cam_mid, z_obj_mid = [], []
        for block in self.camera_blocks:
            z = block(z)
            z_cam_mid.append(self.transform_block(z, camera))
        z = z_cam_mid[-1] if z_cam_mid else self.transform_block(z, camera)   # same resample: reuse it
        for block in self.object_blocks:
            z = block(z)
            z_obj_mid.append(z)
        return self.output_block(z), z_cam_mid, z_obj_mid

    def encode(self, fuser, camera, color, depth=None, mask=None, data_parallel=False):
        device = next(self.parameters()).device
        num_views = color.shape[1] if color.dim() == 5 else 1
        parts = []
        if self.input_color:
            parts.append(bv2b(color) if color.dim() == 5 else color)
        if self.input_depth:
            parts.append(bv2b(depth) if depth.dim() == 5 else depth)
        if self.input_mask:
            parts.append(gan_normalize(bv2b(mask) if mask.dim() == 5 else mask))
        x = torch.cat(parts, dim=1).to(device)
        z_obj, z_cam_mid, z_obj_mid = self(x, camera.to(device))
        z_obj = b2bv(z_obj, num_views)
        z_cam_mid = [b2bv(z, num_views) for z in z_cam_mid]
        z_obj_mid = [b2bv(z, num_views) for z in z_obj_mid]
        return fuser(z_obj, z_cam_mid, z_obj_mid, camera)


class Photographer(_Checkpointed):
    def __init__(self, in_size, image_config, camera_config, object_config, projection_type='sum',
                 occlusion_config=False, in_views=1, skip_connections=False, relu_slope=0.2, cube_size=1.0,
                 predict_color=False, predict_depth=True, predict_mask=True, scale_mode='bilinear', **kwargs):
        super().__init__()
        self.image_config, self.camera_config, self.object_config = image_config, camera_config, object_config
        self.occlusion_config, self.projection_type = occlusion_config, projection_type
        self.predict_color, self.predict_depth, self.predict_mask = predict_color, predict_depth, predict_mask
        self.in_views, self.relu_slope, self.skip_connections = in_views, relu_slope, skip_connections
        self.cube_size, self.scale_mode, self.in_size = cube_size, scale_mode, in_size
        self.out_channels = ([3] if predict_color else []) + ([1] if predict_depth else []) + ([1] if predict_mask else [])
        self._ctor_args = dict(image_config=image_config, camera_config=camera_config, occlusion_config=occlusion_config,
                               object_config=object_config, projection_type=projection_type, relu_slope=relu_slope,
                               out_channels=self.out_channels, in_views=in_views, in_size=in_size,
                               skip_connections=skip_connections, cube_size=cube_size, predict_color=predict_color,
                               predict_depth=predict_depth, predict_mask=predict_mask, scale_mode=scale_mode)

        # skip_connections (reference :296-313): object blocks from the second one and "all" camera blocks take the
        # matching encoder activations as extra input channels.  Quirk Q20: the camera blocks are created with
        # skip_connect_start=True (== 1), so camera block 0 gets NO extra channels allocated although forward
        # concatenates them for every camera block -- the reference cannot run forward() with skip_connections=True
        # (no shipped recipe sets it).  Construction / state_dict are reproduced exactly; forward concatenates like
        # the reference and fails on the same channel mismatch.
        self.object_blocks = (create_blocks(object_config, EqualizedConv3d, 2.0, in_views=in_views,
                                            skip_connections=skip_connections, scale_mode=scale_mode)
                              if object_config else nn.ModuleList())
        self.transform_block = ObjectToCameraTransform(cube_size)
        self.occlusion_module = (unet.UNet3d(object_config[-1] + 1, 1, occlusion_config) if occlusion_config else None)
        self.camera_blocks = create_blocks(camera_config, EqualizedConv3d, 2.0, skip_connections=skip_connections,
                                           skip_connect_start=True, skip_connection_views=in_views, scale_mode=scale_mode)
        self.projection_block = (FactorProjection3d2d(camera_config[-1], image_config[0][0], out_size=self.camera_out_size)
                                 if projection_type == 'factor' else None)
        self.image_decoder = unet.UNet2d(None, None, image_config)
        self.output_blocks = nn.ModuleList([OutputBlock2d(self.image_decoder.out_channels, c) for c in self.out_channels])

    @property
    def object_out_size(self):
        return self.in_size * (2 ** self.object_config.count('U'))

    @property
    def camera_out_size(self):
        return self.object_out_size * (2 ** self.camera_config.count('U'))

    @property
    def image_bottleneck_size(self):
        return self.image_decoder.bottleneck_size(self.camera_out_size)

    @property
    def out_size(self):
        return self.image_decoder.output_size(self.camera_out_size)

    def forward(self, z_obj, camera, z_cam_mid=None, z_obj_mid=None, return_latent=False, broadcast=False):
        """broadcast=True (used by decode): z_obj is ONE volume (1,C,S,S,S) rendered under every camera -- the resampler reads
        it with a zero batch stride, and its gradient comes back as one volume instead of through an expanded view."""
        if z_obj.shape[0] != len(camera) and not (broadcast and z_obj.shape[0] == 1):
            raise ValueError(f'batch dimension of z_obj and camera much match. ({z_obj.shape[0]} != {len(camera)})')
        if self.skip_connections and (z_cam_mid is None or z_obj_mid is None):
            raise ValueError('z_cam_intermediate / z_obj_intermediate required for skip connections.')
        if self.skip_connections:                                     # reference :407-409
            z_cam_mid = [self.transform_block(zc, camera) for zc in z_cam_mid]
        z = z_obj
        for block_id, block in enumerate(self.object_blocks):
            if self.skip_connections and block_id >= 1:
                z = torch.cat((z, z_obj_mid[-block_id - 1]), dim=1)
            z = self._checked(block, z)
        z = self.transform_block(z, camera)
        for block_id, block in enumerate(self.camera_blocks):
            if self.skip_connections:
                z = torch.cat((z, z_cam_mid[-block_id - 1]), dim=1)
            z = self._checked(block, z)
        z_depth = None
        if self.occlusion_module is not None:                 # reference :378-395,427-430
            logits = self.occlusion_module(torch.cat((z, get_normalized_voxel_depth(z)), dim=1))
            # softmax over the depth column + expected depth in one pass (lf_column_softmax_fwd)
            w, z_depth = ops.column_softmax(logits)
            if tuple(logits.shape[-3:]) != tuple(z.shape[-3:]):
                # occlusion U-Net at another resolution: nearest resize of the logits first (reference :385)
                w = ops.column_softmax(ops.resize_nearest_to(logits, z.size(-1)))[0]
            z = ops.column_scale(z, w)
        if self.projection_type == 'sum':
            z = ops.column_sum(z)
        elif self.projection_type == 'factor':
            z = self.projection_block(z)
        y, rescale = self.decode_features(z)
        y = torch.cat([ob(y) for ob in self.output_blocks], dim=1)
        if rescale is not None:
            y = rescale(y)
        return y, (z if return_latent else None), z_depth

    def decode_features(self, z):
        """The 2-D decoder on the projected latent: (features, rescale) where `rescale` is None or the resize the caller owes the
        LOGITS.  When the decoder ends in an up-sampling (the released architecture: ..., 'U', 64 at 128^2 -> 256^2,
        tools/train/train.sh:28-66; the reference's 2-D U-Nets resize bilinearly, modules/blocks.py:10-75) and the output blocks
        are pointwise convolutions without activation (reference recon/models.py:316-327, blocks.py:108-119), the resize commutes
        with them: it is a linear map over space whose weights sum to one, they are affine over channels, so
        heads(resize(x)) = resize(heads(x)) -- bit for bit with nearest neighbours, to fp32 rounding with bilinear weights.  The
        heads then run on a quarter of the pixels and the 64-channel 256^2 tensor (2.1 GB for the 128 renders of a cross-entropy
        iteration) is never written."""
        dec = self.image_decoder
        last = dec.up_blocks[-1] if len(dec.up_blocks) else None
        if (last is not None and dec.output_block is None and last.interpolate is not None and last.interpolate.mode in ('nearest', 'bilinear')
                and all(getattr(ob, 'activation', None) is None and getattr(ob.conv, 'kernel_size', 0) == 1 for ob in self.output_blocks)):
            return dec(z, defer_last_rescale=True), last.interpolate
        return dec(z), None

    @staticmethod
    def _checked(block, z):
        want = block.conv1.module.weight.shape[1]
        if z.shape[1] != want:          # the message F.conv3d gives in the reference (Q20)
            raise RuntimeError(f'Given groups=1, weight of size {list(block.conv1.module.weight.shape)}, expected input'
                               f'{list(z.shape)} to have {want} channels, but got {z.shape[1]} channels instead')
        return block(z)

    def interpret_logits(self, logits, apply_mask=False):
        """logits -> {'depth','mask',(+'color'), '*_logits'} (reference :455-484)."""
        y, base = {}, 0
        if self.predict_color:
            y['color_logits'] = logits[:, base:base + 3]
            y['color'] = torch.tanh(y['color_logits'])
            base += 3
        if self.predict_depth:
            y['depth_logits'] = logits[:, base:base + 1]
            y['depth'] = torch.tanh(y['depth_logits'])
            base += 1
        if self.predict_mask:
            y['mask_logits'] = logits[:, base:base + 1]
            y['mask'] = torch.sigmoid(y['mask_logits'])
        else:
            y['mask'] = (y['depth'].detach() > -1.0).float()
            y['mask_logits'] = 100 * y['mask'] + (-100) * (1.0 - y['mask'])
        if apply_mask and self.predict_mask:
            if self.predict_depth:
                y['depth'] = (y['depth'] + 1) * (y['mask'] > 0.5) - 1
            if self.predict_color:
                y['color'] = y['color'] * (y['mask'] > 0.5)
        return y

    def decode(self, z_obj, camera, interpret_logits=True, return_latent=False, data_parallel=False, apply_mask=False):
        """z_obj (B,1,C,S,S,S) is broadcast (stride 0, no copy) over the views of `camera`."""
        num_views = camera.length // z_obj.shape[0]
        if z_obj.shape[0] == 1 and not len(self.object_blocks) and not self.skip_connections:
            # one object, no object-frame blocks: the volume goes to the resampler un-expanded (its backward then returns one
            # volume; an expanded view costs a zero-filled (V,C,S,S,S) gradient plus a reduction over V)
            y, z, z_depth = self(z_obj[0], camera, return_latent=return_latent, broadcast=True)
        else:
            if z_obj.shape[0] == 1:
                z = z_obj[0].expand(num_views, -1, -1, -1, -1)
            else:
                z = z_obj.expand(-1, num_views, -1, -1, -1, -1).reshape(-1, *z_obj.shape[2:])
            y, z, z_depth = self(z, camera, return_latent=return_latent)
        if z is not None:
            z = b2bv(z, num_views)
        if interpret_logits:
            y = {k: b2bv(v, num_views) for k, v in self.interpret_logits(y, apply_mask=apply_mask).items()}
        return y, z, z_depth
