"""Multi-view fusers (API mirror of latentfusion/recon/fusion.py:17-246).

z_obj is (B, V, C, S, S, S).  Pool fusers are reductions over the view axis (shardable over
GPUs with one all-reduce of the fused volume, see latentfusion_amd.parallel); GRU/LSTM fusers
are order-dependent recurrences whose gate convolutions run on the fused conv kernel."""
import abc

import torch
from torch import nn

from ..functional import absolute_max_pool  # noqa: F401  (reference functional.py:47-49; re-exported here)
from ..modules import EqualizedConv2d, EqualizedConv3d, unet
from ..modules.geometry import CameraToObjectTransform
from ..three.batchview import b2bv, bv2b
from . import utils


def get_fuser(fuser_type, in_channels, cube_size, block_config=None, conv_module=EqualizedConv3d):
    if fuser_type.startswith('pool:'):
        return PoolFuser(fuser_type.split(':')[1])
    if fuser_type == 'concat':
        return ConcatFuser()
    if fuser_type == 'blend':
        return BlendFuser(block_config, in_channels=in_channels, cube_size=cube_size, conv_module=conv_module)
    if fuser_type == 'gru':
        return GRUFuser(in_channels=in_channels, cube_size=cube_size, conv_module=conv_module)
    if fuser_type == 'lstm':
        return LSTMFuser(in_channels=in_channels, cube_size=cube_size, conv_module=conv_module)
    raise ValueError(f'Unknown fuser type {fuser_type!r}')


def from_checkpoint(checkpoint):
    return globals()[checkpoint['type']].from_checkpoint(checkpoint)


def pool_tensor(tensor, pool_type, dim=0):
    """Device tensors pooled over the view axis of a (B,V,...) stack go through lf_fuse_views_fwd (no ATen on the
    product path).  Host tensors -- only the world-size-2 gloo tests of the sharding logic (tests/test_parallel.py)
    ever pass them -- and other axes use the expressions below."""
    if tensor.is_cuda and dim == 1 and tensor.dim() >= 3:
        from .. import ops
        return ops.fuse_views(tensor, pool_type)
    if pool_type == 'max':
        return tensor.max(dim=dim, keepdim=True)[0]
    if pool_type == 'abs_max':
        return absolute_max_pool(tensor, dim=dim)
    if pool_type == 'mean':
        return tensor.mean(dim=dim, keepdim=True)
    if pool_type == 'median':
        return tensor.median(dim=dim, keepdim=True)[0]        # lower median for even V
    raise ValueError(f'Unknown pool_type value {pool_type}')


class Fuser(nn.Module, abc.ABC):
    @classmethod
    def from_checkpoint(cls, checkpoint):
        if 'args' in checkpoint:
            model = cls(**checkpoint['args'])
            model.load_state_dict(checkpoint['state_dict'])
            return model
        return cls(**({'pool_type': checkpoint['pool_type']} if 'pool_type' in checkpoint else {}))

    def create_checkpoint(self):
        return {'type': self.__class__.__qualname__}


class PoolFuser(Fuser):
    def __init__(self, pool_type='mean'):
        super().__init__()
        self.pool_type = pool_type

    def create_checkpoint(self):
        # the reference drops pool_type on save (fusion.py:66-69); keep it as an extra key
        return {'type': 'PoolFuser', 'pool_type': self.pool_type}

    def forward(self, z_obj, z_cam_mid, z_obj_mid, camera):
        return pool_tensor(z_obj, self.pool_type, dim=1), {}


class ConcatFuser(Fuser):
    def forward(self, z_obj, z_cam_mid, z_obj_mid, camera):
        N, V, C, D, H, W = z_obj.shape
        return z_obj.reshape(N, 1, V * C, D, H, W), {}


class _ArgsFuser(Fuser):
    def create_checkpoint(self):
        return {'type': self.__class__.__qualname__, 'args': self._args(),
                'state_dict': {k: v.cpu() for k, v in self.state_dict().items()}}


class BlendFuser(_ArgsFuser):
    """Per-view blend logits from a 3-D U-Net, softmax over views (reference :95-149)."""

    def __init__(self, block_config, in_channels, cube_size=1.0, conv_module=EqualizedConv3d):
        super().__init__()
        self.block_config, self.in_channels, self.cube_size = block_config, in_channels, cube_size
        self.unet = unet.BaseUNet(in_channels + 1, 1, block_config, conv_module=conv_module)
        self.transform_block = CameraToObjectTransform(cube_size)

    def _args(self):
        return {'block_config': self.block_config, 'in_channels': self.in_channels, 'cube_size': self.cube_size}

    def compute_blend_logits(self, z_cam, camera):
        V = z_cam.shape[1]
        z_cam = bv2b(z_cam)
        w = self.unet(torch.cat((z_cam, utils.get_normalized_voxel_depth(z_cam)), dim=1))
        return b2bv(self.transform_block(w, camera), V)

    def compute_blend_weights(self, z_cam, camera):
        return torch.softmax(self.compute_blend_logits(z_cam, camera), dim=1)

    def forward(self, z_obj, z_cam_mid, z_obj_mid, camera):
        # softmax over the views and the weighted sum in one pass over the V volumes (lf_fuse_blend_fwd)
        from .. import ops
        out, w = ops.fuse_blend(z_obj, self.compute_blend_logits(z_cam_mid[-1], camera))
        return out, {'blend_weights': w.squeeze(2)}


class ConvGRUCell(nn.Module):
    """u = s(conv_u[x,h]); r = s(conv_r[x,h]); c = conv_o[x, h*r] (NO tanh);
    h' = h(1-u) + c u   (reference modules/gru.py:7-43)."""

    def __init__(self, in_channels, hidden_channels, kernel_size, bias=True, conv_module=EqualizedConv3d):
        super().__init__()
        self.input_dim, self.hidden_dim = in_channels, hidden_channels
        pad = kernel_size // 2
        self.update_gate = conv_module(in_channels + hidden_channels, hidden_channels, kernel_size, padding=pad, bias=bias)
        self.reset_gate = conv_module(in_channels + hidden_channels, hidden_channels, kernel_size, padding=pad, bias=bias)
        self.out_gate = conv_module(in_channels + hidden_channels, hidden_channels, kernel_size, padding=pad, bias=bias)

    def init_hidden(self, b, h, w):
        """reference gru.py:45-46 (2-D zero state on the current device)."""
        return torch.zeros(b, self.hidden_dim, h, w, device=self.update_gate.module.weight.device)

    def forward(self, x, h_cur):
        x_in = torch.cat([x, h_cur], dim=1)
        update = torch.sigmoid(self.update_gate(x_in))
        reset = torch.sigmoid(self.reset_gate(x_in))
        x_out = self.out_gate(torch.cat([x, h_cur * reset], dim=1))
        return h_cur * (1 - update) + x_out * update

    def parts_ok(self, z, h):
        """3-D, 16 state channels, 16 + 3 input channels, on the device: the gates can run as sums of 16 -> 16
        Winograd convolutions over (view, coords, state) without building the 35-channel concatenation."""
        w = self.update_gate.module.weight
        return (z.is_cuda and z.dim() == 5 and w.dim() == 5 and self.hidden_dim == 16 and self.input_dim == 19
                and z.shape[1] == 16 and h.shape[1] == 16 and z.shape[2] * z.shape[3] * z.shape[4] * 64 < 2 ** 31)

    def coords_base(self, c16):
        """conv(coords) * he + bias of the three gates: the part of every gate pre-activation that does not depend on the
        view or the state, evaluated once per forward (the inference path does the same); its weight columns and the bias
        get their gradients from the sum of the gate gradients over the views."""
        from .. import ops
        return tuple(ops.conv3x3_sum16(g.module.weight, g.bias, None, (c16,), cols=((16, 3),))
                     for g in (self.update_gate, self.reset_gate, self.out_gate))

    def forward_parts(self, z, c16, h_cur, base=None):
        """forward(cat(z, coords), h_cur) with the coordinate channels given zero-padded to 16 (`c16`), or with their share of
        the three gates precomputed (`base` = coords_base(c16))."""
        from .. import ops

        def gate(i, g, state):
            if base is not None:
                return ops.conv3x3_sum16(g.module.weight, None, None, (z, state), cols=((0, 16), (19, 16)), addend=base[i])
            return ops.conv3x3_sum16(g.module.weight, g.bias, (16, 3, 16), (z, c16, state))
        update, rh = ops.gru_gates(gate(0, self.update_gate, h_cur), gate(1, self.reset_gate, h_cur), h_cur)
        return ops.gru_blend(h_cur, update, gate(2, self.out_gate, rh))


class ConvLSTMCell(nn.Module):
    """reference modules/lstm.py:7-56."""

    def __init__(self, in_channels, hidden_channels, kernel_size, bias=True, conv_module=EqualizedConv3d):
        super().__init__()
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.conv = conv_module(in_channels + hidden_channels, 4 * hidden_channels, kernel_size,
                                padding=kernel_size // 2, bias=bias)

    def forward(self, input_tensor, cur_state):
        h_cur, c_cur = cur_state
        cc = self.conv(torch.cat([input_tensor, h_cur], dim=1))
        if cc.is_cuda:                                        # gate arithmetic in one kernel (lf_lstm_cell_fwd / _bwd)
            from .. import ops
            return ops.lstm_cell(cc, c_cur)
        i, f, o, g = torch.split(cc, self.hidden_channels, dim=1)          # (host tensors: CPU-side tests only)
        c_next = torch.sigmoid(f) * c_cur + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c_next), c_next


class GRUFuser(_ArgsFuser):
    """h0 = view 0; for each further view x = cat(z_i, voxel coords (z,y,x)) (reference :152-201)."""

    recurrence = 'gru'             # parallel.fuse_sharded pipelines the state over the ranks

    def __init__(self, in_channels, cube_size=1.0, conv_module=EqualizedConv3d):
        super().__init__()
        self.in_channels, self.cube_size, self.conv_module = in_channels, cube_size, conv_module
        n_coord = 2 if conv_module == EqualizedConv2d else 3
        self.gru = ConvGRUCell(in_channels + n_coord, in_channels, kernel_size=3, bias=True, conv_module=conv_module)
        self.split_gates = True        # False: always the concatenated 35-channel convolutions (tests compare the two)
        self.hoist_coords = True       # the coordinate channels' share of the gates once per forward, not once per view
        self.fused_recurrence = True   # the whole recurrence as one autograd node (ops.gru_fuse); False: per-gate functions

    def _args(self):
        return {'in_channels': self.in_channels, 'cube_size': self.cube_size}

    def forward(self, z_obj, z_cam_mid, z_obj_mid, camera):
        if (not torch.is_grad_enabled() and z_obj.is_cuda and z_obj.shape[0] == 1 and z_obj.dim() == 6
                and self.conv_module != EqualizedConv2d):
            from .. import ops as _o
            return self._forward_inference(_o._f32(z_obj)), {}
        # (not z_obj[:, i]: the backward of V selects is V zero-filled copies of the whole view stack plus V - 1 full-size
        # additions -- 1 GB each at 8 x 128^3 x 16; ops.split_views assembles the V gradients in one pass)
        from .. import ops as _ops
        if (self.fused_recurrence and self.hoist_coords and self.split_gates and z_obj.dim() == 6 and z_obj.shape[0] == 1
                and self.gru.parts_ok(z_obj[:, 0], z_obj[:, 0])):
            # the whole recurrence as one autograd node (views in fp32 or bf16 storage)
            h0 = z_obj[:, 0]
            c16 = _ops.empty_cl((1, 16) + tuple(h0.shape[2:]), h0.device).zero_()
            c16[:, :3] = utils.get_normalized_voxel_coords(h0)
            return _ops.gru_fuse(z_obj, c16, self.gru).unsqueeze(1), {}
        z_obj = _ops._f32(z_obj)                                   # (the per-gate functions below work on fp32 storage)
        views = _ops.split_views(z_obj)
        h = views[0]
        coords = (utils.get_normalized_pixel_coords(h) if self.conv_module == EqualizedConv2d
                  else utils.get_normalized_voxel_coords(h))
        if self.split_gates and self.gru.parts_ok(h, h):
            from .. import ops
            c16 = ops.empty_cl((h.shape[0], 16) + tuple(h.shape[2:]), h.device).zero_()
            c16[:, :3] = coords
            base = self.gru.coords_base(c16) if self.hoist_coords else None
            for v in views[1:]:
                h = self.gru.forward_parts(v, c16, h, base)
            return h.unsqueeze(1), {}
        for v in views[1:]:
            h = self.gru(torch.cat((v, coords), dim=1), h)
        return h.unsqueeze(1), {}

    def _forward_inference(self, z_obj):
        """Same recurrence without the three concatenations and five element-wise passes per view.

        16-channel volumes: every gate convolution over [z_i | coords | state] is evaluated as a sum of 16 -> 16
        Winograd convolutions (lf_conv3d_c16_wino, the addend form): coords part once per object, z_i part, then
        the state part added on top; the gate arithmetic (lf_gru_stage_a/b) works on plain 16-channel tensors.
        Other widths: one channels-last record [z_i | coords | state] per voxel feeds a merged update+reset
        convolution and the out convolution; the gate arithmetic writes the state slots in place."""
        from .. import _lib, ops
        L = _lib.lib()
        cell = self.gru
        V, C = z_obj.shape[1], z_obj.shape[2]
        D, H, W = z_obj.shape[3:]
        dev = z_obj.device
        nvox = D * H * W
        s = torch.cuda.current_stream().cuda_stream
        gates = (cell.update_gate, cell.reset_gate, cell.out_gate)
        coords = utils.get_normalized_voxel_coords(z_obj[:, 0])
        if C == 16 and nvox * 64 < 2 ** 31:
            he = ops.he_constant(cell.update_gate.module.weight)

            def packs(gate):
                w = gate.module.weight

                def make():
                    wd = w.detach()
                    wc = wd.new_zeros(16, 16, 3, 3, 3)
                    wc[:, :3] = wd[:, C:C + 3]
                    return tuple(ops.pack_conv3d_c16_wino(p) for p in (wd[:, :C].contiguous(), wc, wd[:, C + 3:].contiguous()))
                return ops._cached(w, 'gru_wino', make)
            c16 = ops.empty_cl((1, 16, D, H, W), dev).zero_()
            c16[:, :3] = coords
            add = (None, None, _lib.LF_EPI_ADD)
            # coords part + bias of every gate, once per object
            base = []
            for g in gates:
                b = g.bias.detach() if g.bias is not None else None
                base.append(ops.conv3d_c16_wino(c16, packs(g)[1], b, he, 0)[0])
            del c16
            h = ops.cl(z_obj[:, 0]).clone()
            u = torch.empty(nvox * C, device=dev, dtype=torch.float32)
            rh = ops.empty_cl((1, C, D, H, W), dev)
            for i in range(1, V):
                zi = ops.cl(z_obj[:, i])
                pre = []
                for k in (0, 1):
                    xk = ops.conv3d_c16_wino(zi, packs(gates[k])[0], None, he, 0, prev=(base[k],) + add[1:])[0]
                    pre.append(ops.conv3d_c16_wino(h, packs(gates[k])[2], None, he, 0, prev=(xk,) + add[1:])[0])
                _lib.check(L.lf_gru_stage_a(pre[0].data_ptr(), pre[1].data_ptr(), C, h.data_ptr(), u.data_ptr(), rh.data_ptr(),
                                            nvox, C, C, 0, s), 'lf_gru_stage_a')
                xo = ops.conv3d_c16_wino(zi, packs(gates[2])[0], None, he, 0, prev=(base[2],) + add[1:])[0]
                cand = ops.conv3d_c16_wino(rh, packs(gates[2])[2], None, he, 0, prev=(xo,) + add[1:])[0]
                h_new = torch.empty_like(h)
                _lib.check(L.lf_gru_stage_b(h.data_ptr(), u.data_ptr(), cand.data_ptr(), h_new.data_ptr(), None, nvox, C, C, 0, s),
                           'lf_gru_stage_b')
                h = h_new
            return h.unsqueeze(1)
        rec = ops.empty_cl((1, 2 * C + 3, D, H, W), dev)
        rec[:, C:C + 3] = coords
        h = ops.cl(z_obj[:, 0]).clone()
        w_ur = torch.cat((cell.update_gate.module.weight, cell.reset_gate.module.weight), dim=0).detach()
        b_ur = torch.cat((cell.update_gate.bias, cell.reset_gate.bias), dim=0).detach() \
            if cell.update_gate.bias is not None else None
        w_o, b_o = cell.out_gate.module.weight.detach(), cell.out_gate.bias
        u = torch.empty(nvox * C, device=dev, dtype=torch.float32)
        rec[:, C + 3:] = h
        for i in range(1, V):
            rec[:, :C] = z_obj[:, i]
            ur = ops.conv3x3(rec, w_ur, b_ur, lrelu=False, pixelnorm=False)
            _lib.check(L.lf_gru_stage_a(ur.data_ptr(), ur.data_ptr() + 4 * C, 2 * C, h.data_ptr(), u.data_ptr(), rec.data_ptr(),
                                        nvox, C, 2 * C + 3, C + 3, s), 'lf_gru_stage_a')
            cand = ops.conv3x3(rec, w_o, b_o.detach() if b_o is not None else None, lrelu=False, pixelnorm=False)
            h_new = torch.empty_like(h)
            _lib.check(L.lf_gru_stage_b(h.data_ptr(), u.data_ptr(), cand.data_ptr(), h_new.data_ptr(),
                                        rec.data_ptr() if i + 1 < V else None, nvox, C, 2 * C + 3, C + 3, s), 'lf_gru_stage_b')
            h = h_new
        return h.unsqueeze(1)


class LSTMFuser(_ArgsFuser):
    recurrence = 'lstm'            # parallel.fuse_sharded pipelines (h, c) over the ranks

    def __init__(self, in_channels, cube_size=1.0, conv_module=EqualizedConv3d):
        super().__init__()
        self.in_channels, self.cube_size = in_channels, cube_size
        self.lstm = ConvLSTMCell(in_channels + 3, in_channels, kernel_size=3, bias=True, conv_module=conv_module)

    def _args(self):
        return {'in_channels': self.in_channels, 'cube_size': self.cube_size}

    def forward(self, z_obj, z_cam_mid, z_obj_mid, camera, initial_state=None):
        """reference :226-246 (h0 = view 0, c0 = 0).  `initial_state` = (h, c) continues a recurrence begun elsewhere
        (every view of `z_obj` is then a step; view-sharded reconstruction, parallel.fuse_sharded); the final (h, c) is
        returned in the second result under 'state'."""
        from .. import ops as _ops
        z_obj = _ops._f32(z_obj)
        views = _ops.split_views(z_obj)                            # (one pass in the backward, see GRUFuser.forward)
        if initial_state is None:
            h, steps = views[0], views[1:]
            c = torch.zeros_like(h)
        else:
            (h, c), steps = initial_state, views
        coords = utils.get_normalized_voxel_coords(h)
        for v in steps:
            h, c = self.lstm(torch.cat((v, coords), dim=1), (h, c))
        return h.unsqueeze(1), {'state': (h, c)}
