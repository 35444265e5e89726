"""Conv blocks and the block-config mini-language (API mirror of latentfusion/modules/blocks.py)."""
from torch import nn

from . import EqualizedConv2d, EqualizedConv3d, Interpolate


def count_blocks(config):
    return sum(1 for b in config if isinstance(b, int)) - 1


def create_blocks(config, conv_module, scale_factor, scale_mode='bilinear', kernel_size=3, skip_connections=False,
                  skip_connect_start=1, skip_connect_end=None, in_views=1, skip_connection_views=None):
    """Decodes e.g. [64,'D',128,128,'U',64] into Blocks (reference blocks.py:10-75): integers are
    channel counts, tokens I/U/D set the rescale applied at the END of the NEXT block."""
    if conv_module == EqualizedConv3d and scale_mode == 'bilinear':
        scale_mode = 'trilinear'
    if skip_connection_views is None:
        skip_connection_views = in_views
    n_blocks = count_blocks(config)
    skip_connect_end = n_blocks if skip_connect_end is None else min(n_blocks, skip_connect_end)
    blocks, idx, pending, c_in = [], 0, 1.0, config[0]
    for tok in config[1:]:
        if isinstance(tok, int) or (isinstance(tok, str) and tok.isdigit()):
            extra = c_in * skip_connection_views if (skip_connections and skip_connect_start <= idx < skip_connect_end) else 0
            if idx == 0:
                c_in *= in_views
            blocks.append(Block(c_in + extra, int(tok), kernel_size=kernel_size, conv_module=conv_module,
                                scale_mode=scale_mode, scale_factor=pending))
            c_in, idx, pending = int(tok), idx + 1, 1.0
        elif tok == 'I':
            pending = scale_factor
        elif tok == 'U':
            pending = 2.0
        elif tok == 'D':
            pending = 0.5
        else:
            raise ValueError(f'Unknown block type {tok!r}')
    return nn.ModuleList(blocks)


class InputBlock(nn.Module):
    """1x1 conv + LeakyReLU (reference blocks.py:78-91)."""

    def __init__(self, in_channels, out_channels, conv_module, kernel_size=1, relu_slope=0.2, padding=0):
        super().__init__()
        self.conv = conv_module(in_channels, out_channels, kernel_size, padding=padding)

    def forward(self, x):
        return self.conv(x, fuse_act=True)


class OutputBlock(nn.Module):
    """1x1 conv (+ optional activation module) (reference blocks.py:108-119)."""

    def __init__(self, in_channels, out_channels, conv_module, kernel_size=1, padding=0, activation=None):
        super().__init__()
        self.conv = conv_module(in_channels, out_channels, kernel_size, padding=padding)
        self.activation = activation

    def forward(self, x):
        x = self.conv(x)
        return self.activation(x) if self.activation else x


class InputBlock2d(InputBlock):
    def __init__(self, in_channels, out_channels, **kw):
        super().__init__(in_channels, out_channels, EqualizedConv2d, **kw)


class InputBlock3d(InputBlock):
    def __init__(self, in_channels, out_channels, **kw):
        super().__init__(in_channels, out_channels, EqualizedConv3d, **kw)


class OutputBlock2d(OutputBlock):
    def __init__(self, in_channels, out_channels, **kw):
        super().__init__(in_channels, out_channels, EqualizedConv2d, **kw)


class OutputBlock3d(OutputBlock):
    def __init__(self, in_channels, out_channels, **kw):
        super().__init__(in_channels, out_channels, EqualizedConv3d, **kw)


class Block(nn.Module):
    """(conv3 -> LeakyReLU(0.2) -> PixelNorm) x2 -> optional rescale (reference blocks.py:136-164);
    each conv+activation+norm triple is ONE fused kernel launch here."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, relu_slope=0.2,
                 conv_module=EqualizedConv3d, scale_factor=1.0, scale_mode='bilinear'):
        super().__init__()
        if relu_slope != 0.2:
            raise NotImplementedError('the fused epilogue is built for the reference slope 0.2')
        self.conv1 = conv_module(in_channels, out_channels, kernel_size, padding=padding)
        self.conv2 = conv_module(out_channels, out_channels, kernel_size, padding=padding)
        self.interpolate = None
        if scale_factor != 1.0 and scale_factor is not None:
            self.interpolate = Interpolate(scale_factor, mode=scale_mode)

    def forward(self, x, rescale=True):
        """rescale=False leaves the block's resize to the caller (a nearest up-sampling commutes with the pointwise layers behind
        it: Photographer.decode_features)."""
        x = self.conv1(x, fuse_act=True, fuse_norm=True)
        x = self.conv2(x, fuse_act=True, fuse_norm=True, chain=True)     # (conv1's output has no other consumer)
        return self.interpolate(x) if (self.interpolate and rescale) else x
