"""L1 operator modules of the hot path (API mirror of latentfusion/modules/__init__.py,
equalized.py): PixelNorm, Interpolate, Equalized convolutions.  Parameters keep the reference's
state_dict key names (`<conv>.module.weight`, `<conv>.bias`) so released checkpoints load as-is;
the arithmetic runs in the HIP kernels behind latentfusion_amd.ops."""
import torch
from torch import nn
from torch.nn import functional as F

from .. import ops


class PixelNorm(nn.Module):
    """x / sqrt(mean_c(x^2) + 1e-8)  (reference modules/__init__.py:8-15)."""

    def forward(self, x):
        return ops.pixelnorm(x)


class Interpolate(nn.Module):
    """Fixed-factor resize (reference modules/__init__.py:18-36)."""

    def __init__(self, scale_factor, mode='nearest'):
        super().__init__()
        self.scale_factor = scale_factor
        self.mode = mode
        self.align_corners = False if mode in ('bilinear', 'trilinear') else None

    def forward(self, x):
        return ops.interpolate(x, self.scale_factor, self.mode)

    def extra_repr(self):
        return f'scale_factor={self.scale_factor}'


class _Weight(nn.Module):
    """Parameter holder standing where the reference keeps an nn.ConvNd (`.module`)."""

    def __init__(self, out_channels, in_channels, kernel_size, dims):
        super().__init__()
        self.out_channels, self.in_channels = out_channels, in_channels
        self.kernel_size = kernel_size
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, *([kernel_size] * dims)))


class Equalized(nn.Module):
    """conv(x, W) * sqrt(2/fan_in) + bias with W ~ N(0,1)  (reference equalized.py:35-74).
    `fuse_act`/`fuse_norm` let the owning block fold LeakyReLU / PixelNorm into the kernel."""

    dims = 0

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, **kwargs):
        super().__init__()
        if isinstance(stride, (tuple, list)):
            stride = stride[0]
        if stride != 1:
            raise NotImplementedError('only stride 1 is used on the reconstruct-and-render path')
        if not ((kernel_size == 3 and padding == 1) or (kernel_size == 1 and padding == 0)):
            raise NotImplementedError('supported: kernel 3 / padding 1 and kernel 1 / padding 0')
        self.kernel_size, self.padding = kernel_size, padding
        self.module = _Weight(out_channels, in_channels, kernel_size, self.dims)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    @property
    def weight(self):                       # He constant, like the reference attribute
        return ops.he_constant(self.module.weight)

    def get_he_constant(self):
        """sqrt(2 / fan_in) (reference equalized.py:66-74)."""
        return ops.he_constant(self.module.weight)

    def forward(self, x, fuse_act=False, fuse_norm=False, chain=False):
        if self.kernel_size == 3:
            return ops.conv3x3(x, self.module.weight, self.bias, lrelu=fuse_act, pixelnorm=fuse_norm, chain=chain)
        return ops.conv1x1(x, self.module.weight, self.bias, lrelu=fuse_act, pixelnorm=fuse_norm)


class EqualizedConv2d(Equalized):
    dims = 2


class EqualizedConv3d(Equalized):
    dims = 3
