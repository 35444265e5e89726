"""The HIP operators as ORDINARY torch operators (`torch.ops.lf.*`): SURVEY 8(b) recommends registering the C-ABI replacements
with the dispatcher so that the reference's modules can call them like any ATen op.  Importing this module defines the
namespace `lf`; each operator is implemented by the autograd function of ops.py / engine.py that wraps the C entry point
(registered as CompositeImplicitAutograd: the dispatcher runs that implementation and autograd differentiates through it, so
`torch.ops.lf.o2c(vol, coef).sum().backward()` launches lf_resample3d_bwd_coef / lf_resample3d_bwd_vol_det).  No CPU kernel is
registered: the operators raise LFHipError on host tensors, like the wrappers they call.

    import latentfusion_amd.torch_ops
    y = torch.ops.lf.conv_block(x, weight, bias, True, True)       # modules/blocks.py:152-158 in one launch

| operator | reference site it replaces | C entry points behind it |
|---|---|---|
| lf::o2c, lf::c2o | modules/geometry.py:669-690, 625-657 | lf_resample3d_fwd / _bwd_coef / _bwd_vol(_det) |
| lf::conv_block | modules/blocks.py:152-158 (+ equalized.py:57-64) | lf_conv3x3_*, lf_conv3d_c16_wino, lf_wino_fused_gemm |
| lf::conv1x1, lf::factor_project, lf::lift | blocks.py:78-133, geometry.py:711-749 | lf_conv1x1_*, lf_lift_unfold / lf_lift_permute |
| lf::column_sum / column_softmax / column_scale | recon/models.py:378-395,427-437 | lf_column_* |
| lf::fuse_views, lf::fuse_blend | recon/fusion.py:45-57,139-148 | lf_fuse_views_*, lf_fuse_blend_* |
| lf::gru_gates, lf::gru_blend, lf::lstm_cell | modules/gru.py:30-43, lstm.py:47-56 | lf_gru_stage_*, lf_lstm_cell_* |
| lf::pixelnorm, lf::interpolate, lf::grid_sample2d | modules/__init__.py:8-36, geometry.py:20-44 | lf_pixelnorm_fwd, lf_resize_*, lf_grid_sample2d_* |
| lf::camera_coefs, lf::pose_loss | geometry.py:106-153,469-531; pose/estimation.py:70-118 | lf_camera_coefs(_bwd), lf_pose_loss_fwd / _bwd |
"""
import torch

from . import engine, ops

_LIB = torch.library.Library('lf', 'DEF')
SCHEMAS = {}


def _reg(schema, fn):
    name = schema.split('(')[0]
    _LIB.define(schema)
    _LIB.impl(name, fn, 'CompositeImplicitAutograd')
    SCHEMAS[name] = schema


_reg('o2c(Tensor vol, Tensor coef) -> Tensor', ops.resample_o2c)
_reg('c2o(Tensor vol, Tensor coef) -> Tensor', ops.resample_c2o)
_reg('conv_block(Tensor x, Tensor weight, Tensor? bias, bool lrelu, bool pixelnorm) -> Tensor',
     lambda x, weight, bias, lrelu, pixelnorm: ops.conv3x3(x, weight, bias, lrelu, pixelnorm))
_reg('conv1x1(Tensor x, Tensor weight, Tensor? bias, bool lrelu, bool pixelnorm) -> Tensor',
     lambda x, weight, bias, lrelu, pixelnorm: ops.conv1x1(x, weight, bias, lrelu, pixelnorm))
_reg('factor_project(Tensor x, Tensor weight, Tensor? bias) -> Tensor', ops.factor_project)
_reg('lift(Tensor x, Tensor weight, Tensor? bias, int out_size) -> Tensor', ops.lift)
_reg('column_sum(Tensor x) -> Tensor', ops.column_sum)
_reg('column_softmax(Tensor logits) -> (Tensor, Tensor)', lambda logits: tuple(ops.column_softmax(logits)))
_reg('column_scale(Tensor z, Tensor w) -> Tensor', ops.column_scale)
_reg('fuse_views(Tensor z, str pool_type) -> Tensor', ops.fuse_views)
_reg('fuse_blend(Tensor z, Tensor logits) -> (Tensor, Tensor)', lambda z, logits: tuple(ops.fuse_blend(z, logits)))
_reg('gru_gates(Tensor upre, Tensor rpre, Tensor h) -> (Tensor, Tensor)', lambda upre, rpre, h: tuple(ops.gru_gates(upre, rpre, h)))
_reg('gru_blend(Tensor h, Tensor u, Tensor cand) -> Tensor', ops.gru_blend)
_reg('lstm_cell(Tensor cc, Tensor c_cur) -> (Tensor, Tensor)', lambda cc, c_cur: tuple(ops.lstm_cell(cc, c_cur)))
_reg('pixelnorm(Tensor x) -> Tensor', ops.pixelnorm)
_reg('interpolate(Tensor x, float scale_factor, str mode) -> Tensor', ops.interpolate)
_reg('grid_sample2d(Tensor img, Tensor grid, str mode, str padding_mode) -> Tensor', ops.grid_sample2d)
_reg('camera_coefs(Tensor params, Tensor intrinsics, float cube_size, float z_span, int crop_h, int crop_w) -> Tensor',
     lambda params, intrinsics, cube_size, z_span, crop_h, crop_w: engine._CameraCoefs.apply(params, intrinsics, cube_size, z_span,
                                                                                             crop_h, crop_w))
_reg('pose_loss(Tensor logits, Tensor coefs, Tensor target_depth, Tensor target_mask, Tensor weights, int H, int W) -> (Tensor, Tensor)',
     lambda logits, coefs, target_depth, target_mask, weights, H, W: tuple(engine.pose_loss(logits, coefs, target_depth, target_mask,
                                                                                           weights, H, W)))
