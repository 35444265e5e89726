"""Generic 2-D / 3-D U-Net (API mirror of latentfusion/modules/unet.py)."""
import torch
from torch import nn

from . import EqualizedConv2d, EqualizedConv3d
from .blocks import InputBlock, OutputBlock, count_blocks, create_blocks


class BaseUNet(nn.Module):
    def __init__(self, in_channels, out_channels, block_config, conv_module):
        super().__init__()
        self._in_channels, self._out_channels = in_channels, out_channels
        self.block_config = block_config
        down, up = block_config
        self.input_block = InputBlock(in_channels, down[0], conv_module=conv_module) if in_channels is not None else None
        self.down_blocks = create_blocks(down, conv_module, 0.5)
        self.up_blocks = create_blocks(up, conv_module, 2.0, skip_connections=True,
                                       skip_connect_end=min(count_blocks(down), count_blocks(up)))
        if out_channels is None:
            self.output_block = None
        elif isinstance(out_channels, int):
            self.output_block = OutputBlock(up[-1], out_channels, conv_module=conv_module)
        else:
            self.output_block = nn.ModuleList([OutputBlock(up[-1], c, conv_module=conv_module) for c in out_channels])

    @property
    def down_block_config(self):
        return self.block_config[0]

    @property
    def up_block_config(self):
        return self.block_config[1]

    @property
    def in_channels(self):
        return sum(self._in_channels) if self._in_channels is not None else self.down_block_config[0]

    @property
    def out_channels(self):
        return sum(self._out_channels) if self._out_channels is not None else self.up_block_config[-1]

    @classmethod
    def from_checkpoint(cls, checkpoint):
        """reference unet.py:42-46 (the stored `conv_module` name is dropped; the subclass fixes it)."""
        args = dict(checkpoint['args'])
        args.pop('conv_module', None)
        model = cls(**args)
        model.load_state_dict(checkpoint['state_dict'])
        return model

    def create_checkpoint(self):
        return {'args': {'in_channels': self._in_channels, 'out_channels': self._out_channels,
                         'block_config': self.block_config, 'conv_module': None},
                'state_dict': {k: v.cpu() for k, v in self.state_dict().items()}}

    def bottleneck_size(self, in_size):
        return in_size // (2 ** (self.block_config[0].count('I') + self.block_config[0].count('D')))

    def output_size(self, in_size):
        return self.bottleneck_size(in_size) * (2 ** (self.block_config[1].count('I') + self.block_config[1].count('U')))

    def forward(self, z, z_inject=None, return_intermediate=False, defer_last_rescale=False):
        """defer_last_rescale=True (no output block): the LAST up block runs without its resize, which the caller applies after
        the layers it commutes with."""
        if z_inject is not None:
            raise NotImplementedError('z_inject is unused on the reconstruct-and-render path')
        if defer_last_rescale and (self.output_block is not None or not len(self.up_blocks)):
            raise ValueError('defer_last_rescale needs a decoder that ends in an up block')
        if self.input_block is not None:
            z = self.input_block(z)
        mids = []                                   # deepest first
        for blk in self.down_blocks:
            z = blk(z)
            mids.insert(0, z)
        for i, blk in enumerate(self.up_blocks):
            if 1 <= i < len(mids):
                z = torch.cat((z, mids[i]), dim=1)
            z = blk(z, rescale=False) if (defer_last_rescale and i == len(self.up_blocks) - 1) else blk(z)
        if isinstance(self.output_block, OutputBlock):
            z = self.output_block(z)
        elif self.output_block is not None:
            z = torch.cat([ob(z) for ob in self.output_block], dim=1)
        return (z, mids) if return_intermediate else z


class UNet2d(BaseUNet):
    def __init__(self, in_channels, out_channels, block_config):
        super().__init__(in_channels, out_channels, block_config, conv_module=EqualizedConv2d)


class UNet3d(BaseUNet):
    def __init__(self, in_channels, out_channels, block_config):
        super().__init__(in_channels, out_channels, block_config, conv_module=EqualizedConv3d)
