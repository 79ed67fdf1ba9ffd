"""TEST INFRASTRUCTURE ONLY -- a CPU stand-in exposing exactly the MinkowskiEngine 0.5.4
symbols the reference touches, backed by oracle/imf_oracle.py.

Purpose: let tests/golden/gen_golden.py import the reference's own model/*.py and
util/misc.py VERBATIM (in the build container only) and run them end to end, so the
committed golden vectors pin the wiring of the restatement and of the HIP path.
It is never imported by imfnet_amd/ and never shipped as product.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imf_oracle as _o  # noqa: E402

from . import MinkowskiFunctional  # noqa: E402,F401
from . import utils  # noqa: E402,F401


class CoordinateManager:
    def __init__(self, coords):
        self.levels = {1: np.asarray(coords, np.int32)}
        self._rb = {}

    def coords(self, ts):
        if ts not in self.levels:
            self.levels[ts] = _o.downsample(self.coords(ts // 2), ts)[0]
        return self.levels[ts]

    def conv_map(self, ts_in, ksize, stride):
        key = ("c", ts_in, ksize, stride)
        if key not in self._rb:
            self._rb[key] = _o.rulebook(self.coords(ts_in), self.coords(ts_in * stride), ts_in, ksize)
        return self._rb[key]

    def tr_map(self, ts_in, ksize, stride):
        key = ("t", ts_in, ksize, stride)
        if key not in self._rb:
            ts_out = ts_in // stride
            self._rb[key] = _o.rulebook_transpose(self.coords(ts_in), self.coords(ts_out), ts_out, ksize)
        return self._rb[key]


class SparseTensor:
    def __init__(self, features, coordinates=None, device=None, coordinate_map_key=None,
                 coordinate_manager=None, **_):
        self._F = features if device is None else features.to(device)
        if coordinates is not None:
            c = coordinates.cpu().numpy() if torch.is_tensor(coordinates) else np.asarray(coordinates)
            assert len(np.unique(_o.pack_keys(c))) == len(c), "shim expects unique coordinates"
            coordinate_manager = CoordinateManager(c)
            coordinate_map_key = 1
        self.coordinate_manager = coordinate_manager
        self.coordinate_map_key = coordinate_map_key          # == tensor stride in this shim

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return torch.as_tensor(self.coordinate_manager.coords(self.coordinate_map_key))

    def __len__(self):
        return self._F.shape[0]

    def _like(self, feats):
        return SparseTensor(feats, coordinate_map_key=self.coordinate_map_key,
                            coordinate_manager=self.coordinate_manager)

    def __add__(self, other):
        return self._like(self._F + other._F)

    def __iadd__(self, other):
        self._F = self._F + other._F
        return self


class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1,
                 bias=False, kernel_generator=None, expand_coordinates=False, dimension=None):
        super().__init__()
        assert dilation == 1 and dimension == 3
        self.kernel_size, self.stride = kernel_size, stride
        K = kernel_size ** 3
        shape = (K, in_channels, out_channels) if K > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        with torch.no_grad():
            b = 1.0 / np.sqrt(K * in_channels)
            self.kernel.uniform_(-b, b)
            if bias:
                self.bias.uniform_(-b, b)

    def forward(self, x):
        cm, ts = x.coordinate_manager, x.coordinate_map_key
        if self.kernel.dim() == 2:
            nbr, ts_out = None, ts
        elif self.transposed:
            nbr, ts_out = cm.tr_map(ts, self.kernel_size, self.stride), ts // self.stride
        else:
            nbr, ts_out = cm.conv_map(ts, self.kernel_size, self.stride), ts * self.stride
        out = _o.spconv(x.F, self.kernel, nbr)
        if self.bias is not None:
            out = out + self.bias
        return SparseTensor(out, coordinate_map_key=ts_out, coordinate_manager=cm)


class MinkowskiConvolution(_ConvBase):
    pass


class MinkowskiConvolutionTranspose(_ConvBase):
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(self.bn(x.F))


class MinkowskiInstanceNorm(nn.Module):
    def __init__(self, num_features, dimension=-1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x):
        raise NotImplementedError("IN variants are not on the ResUNetBN2C path")


def cat(*tensors):
    return tensors[0]._like(torch.cat([t.F for t in tensors], dim=1))
