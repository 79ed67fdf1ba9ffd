"""Sparse building blocks of the ResUNet: norm factory and the two-conv residual block.

Interface parity with the reference: `get_norm` (model/common.py:4-10), `get_block` /
`BasicBlockBN` / `BasicBlockIN` (model/residual_block.py:9-77) -- same constructor arguments,
same sub-module names (conv1, norm1, conv2, norm2 => identical state_dict keys).

Execution differs: in eval mode ResUNet2 calls `BasicBlockBase.fused`, which issues exactly two
imf_spconv_fwd launches with BatchNorm, the residual add and both ReLUs folded into the MFMA
kernel's epilogue.  `forward` keeps the layer-by-layer semantics for generic use and tests.
"""
import torch.nn as nn

from .. import sparse as ME

_NORMS = {
    'BN': lambda c, momentum, D: ME.MinkowskiBatchNorm(c, momentum=momentum),
    'IN': lambda c, momentum, D: ME.MinkowskiInstanceNorm(c, dimension=D),
}


def get_norm(norm_type, num_feats, bn_momentum=0.05, D=-1):
    if norm_type not in _NORMS:
        raise ValueError(f'Type {norm_type}, not defined')
    return _NORMS[norm_type](num_feats, bn_momentum, D)


class BasicBlockBase(nn.Module):
    expansion = 1
    NORM_TYPE = 'BN'

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
        super().__init__()
        conv = ME.MinkowskiConvolution
        self.conv1 = conv(inplanes, planes, kernel_size=3, stride=stride, dimension=D)
        self.norm1 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.conv2 = conv(planes, planes, kernel_size=3, stride=1, dilation=dilation, bias=False, dimension=D)
        self.norm2 = get_norm(self.NORM_TYPE, planes, bn_momentum=bn_momentum, D=D)
        self.downsample = downsample

    def forward(self, x):
        relu = ME.MinkowskiFunctional.relu
        y = relu(self.norm1(self.conv1(x)))
        y = self.norm2(self.conv2(y))
        y += x if self.downsample is None else self.downsample(x)
        return relu(y)

    def fused(self, x, bn1, bn2):
        """x: SparseTensor; bn1 / bn2: folded eval-BatchNorm (scale, shift) pairs."""
        if self.downsample is not None or self.conv1.stride != 1:
            return self.forward(x)
        mid, _ = self.conv1.run(x, scale=bn1[0], shift=bn1[1], relu=True)
        out, _ = self.conv2.run(x._like(mid), scale=bn2[0], shift=bn2[1], residual=x.F, relu=True)
        return x._like(out)


class BasicBlockBN(BasicBlockBase):
    NORM_TYPE = 'BN'


class BasicBlockIN(BasicBlockBase):
    NORM_TYPE = 'IN'


def get_block(norm_type, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, D=3):
    blocks = {'BN': BasicBlockBN, 'IN': BasicBlockIN}
    if norm_type not in blocks:
        raise ValueError(f'Type {norm_type}, not defined')
    return blocks[norm_type](inplanes, planes, stride, dilation, downsample, bn_momentum, D)
