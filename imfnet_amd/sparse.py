"""The slice of the MinkowskiEngine 0.5.4 Python surface that IMFNet's descriptor path uses
(SURVEY §2.3 lists every call site), re-implemented over libimfnet_hip.so.

Same names, argument meaning and error behaviour as the reference's dependency, so
model/resunet.py-style code reads the same:
    ME.SparseTensor(feats, coordinates=coords, device=device)      util/misc.py:95
    ME.MinkowskiConvolution / MinkowskiConvolutionTranspose        model/resunet.py:42-158
    ME.MinkowskiBatchNorm, ME.cat, MEF.relu, x += y                model/common.py:6, resunet.py:197
    ME.utils.sparse_quantize / batched_coordinates / fnv_hash_vec  util/misc.py:83-86

Design difference from MinkowskiEngine: a fragment is static geometry, so the coordinate manager
builds the whole pyramid and every rulebook once (in HBM, by HIP kernels) and the convolutions
only read them.  There is no CPU backend: tensors must live on the GPU.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import ImfError


class CoordinateMapKey:
    __slots__ = ("tensor_stride",)

    def __init__(self, tensor_stride):
        self.tensor_stride = int(tensor_stride)

    def get_tensor_stride(self):
        return [self.tensor_stride] * 3

    def __eq__(self, o):
        return isinstance(o, CoordinateMapKey) and o.tensor_stride == self.tensor_stride

    def __hash__(self):
        return hash(self.tensor_stride)

    def __repr__(self):
        return f"CoordinateMapKey(tensor_stride={self.tensor_stride})"


class CoordinateManager:
    """Owns the pyramid levels (coordinates + voxel hash per tensor stride) and the rulebooks."""

    @classmethod
    def from_levels(cls, levels):
        cm = cls(levels[0])
        for lv in levels[1:]:
            cm.levels[lv.ts] = lv
        return cm

    def __init__(self, level0, meta=None):
        self.levels = {1: level0}
        self._rulebooks = {}
        self.meta = meta              # optional shared [n_levels,2] count block (row 0 = level0)
        self.meta_used = 1

    # -- levels ---------------------------------------------------------------------------------
    def build_pyramid(self, max_stride=8, before_sync=None):
        """Build all missing levels up to `max_stride` with ONE host synchronisation.  `before_sync`
        runs after every geometry kernel is queued and right before the host blocks."""
        pending = [lv for lv in self.levels.values() if lv.n is None]
        ts = max(self.levels)
        n_bound = None
        meta, mi = self.meta, self.meta_used
        while ts < max_stride:
            src = self.levels[ts]
            if src.n is not None:
                n_bound = src.n
            elif n_bound is None:
                n_bound = src.coords_buf.shape[0]
            row = None
            if meta is not None and mi < meta.shape[0]:
                row, mi = meta[mi], mi + 1
            lv = ops.downsample(src, ts * 2, n_in_max=n_bound, meta=row)
            self.levels[ts * 2] = lv
            pending.append(lv)
            ts *= 2
        self.meta_used = mi
        if before_sync is not None:
            before_sync()
        if pending:
            order = [self.levels[t] for t in sorted(self.levels)]
            shared = meta is not None and len(order) <= meta.shape[0] and \
                all(lv.n_dev.data_ptr() == meta[i].data_ptr() for i, lv in enumerate(order))
            if shared:
                ops.sync_levels(order, meta_block=meta)
            else:
                ops.sync_levels(pending)

    def level(self, ts):
        if ts not in self.levels or self.levels[ts].n is None:
            self.build_pyramid(max(ts, 1))
        return self.levels[ts]

    def coords(self, ts):
        return self.level(ts).coords

    # -- rulebooks ------------------------------------------------------------------------------
    def conv_rulebook(self, ts_in, ksize, stride):
        key = ("conv", ts_in, ksize, stride)
        if key not in self._rulebooks:
            if ksize == 1 and stride == 1:
                lv = self.level(ts_in)
                rb = ops.rulebook_identity(lv.n, None)
            else:
                rb = ops.rulebook_conv(self.level(ts_in), self.level(ts_in * stride), ksize)
            self._rulebooks[key] = rb
        return self._rulebooks[key]

    def transpose_rulebook(self, ts_in, ksize, stride):
        if stride != 2 or ksize != 3 or ts_in % 2:
            raise ImfError("MinkowskiConvolutionTranspose: only kernel_size=3, stride=2 is supported")
        key = ("tr", ts_in, ksize, stride)
        if key not in self._rulebooks:
            self._rulebooks[key] = ops.rulebook_transpose(self.level(ts_in), self.level(ts_in // 2), ksize)
        return self._rulebooks[key]


class SparseTensor:
    """Carrier of `.F` ([M,C] float32, row i <-> coordinate row i) and `.C` ([M,4] int32)."""

    def __init__(self, features, coordinates=None, device=None, coordinate_map_key=None,
                 coordinate_manager=None, tensor_stride=1):
        if device is None:
            device = features.device if features.is_cuda else torch.device("cuda")
        device = torch.device(device)
        if device.type != "cuda":
            raise ImfError("imfnet_amd.SparseTensor lives on the GPU only (no CPU backend)")
        if not torch.is_tensor(features):
            features = torch.as_tensor(np.asarray(features))
        features = features.to(device=device, dtype=torch.float32).contiguous()
        if coordinates is not None:
            if coordinate_manager is not None:
                raise ValueError("pass either coordinates or coordinate_manager, not both")
            if not torch.is_tensor(coordinates):
                coordinates = torch.as_tensor(np.asarray(coordinates))
            if coordinates.dim() != 2 or coordinates.shape[1] != 4:
                raise ValueError(f"coordinates must be [N,4] (batch,x,y,z), got {tuple(coordinates.shape)}")
            if coordinates.shape[0] != features.shape[0]:
                raise ValueError("features and coordinates must have the same number of rows")
            coordinates = coordinates.to(device=device, dtype=torch.int32).contiguous()
            level0 = ops.level_from_coords(coordinates)
            ops.sync_levels([level0])
            if level0.n != coordinates.shape[0]:
                # util/misc.py never hits this: sparse_quantize already made the rows unique
                raise ImfError("duplicate coordinates: quantise with utils.sparse_quantize first")
            coordinate_manager = CoordinateManager(level0)
            coordinate_map_key = CoordinateMapKey(tensor_stride)
        elif coordinate_manager is None or coordinate_map_key is None:
            raise ValueError("SparseTensor needs coordinates or (coordinate_map_key, coordinate_manager)")
        self._F = features
        self.coordinate_manager = coordinate_manager
        self.coordinate_map_key = coordinate_map_key

    # reference reads .F / .C and assigns ._F (model/resunet.py:189,230)
    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.coords(self.coordinate_map_key.tensor_stride)

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def device(self):
        return self._F.device

    def __len__(self):
        return self._F.shape[0]

    def _like(self, feats, ts=None):
        key = self.coordinate_map_key if ts is None else CoordinateMapKey(ts)
        return SparseTensor(feats, coordinate_map_key=key, coordinate_manager=self.coordinate_manager)

    def __add__(self, other):
        return self._like(self._F + other._F)

    def __iadd__(self, other):                     # model/residual_block.py:50
        self._F = self._F + other._F
        return self

    def __repr__(self):
        return f"SparseTensor(F={tuple(self._F.shape)}, {self.coordinate_map_key})"


def cat(*tensors):
    """ME.cat: channel concat on the same coordinate map, first argument's channels first."""
    k = tensors[0].coordinate_map_key
    if any(t.coordinate_map_key != k for t in tensors):
        raise ValueError("ME.cat: tensors must share a coordinate map")
    return tensors[0]._like(torch.cat([t.F for t in tensors], dim=1))


class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


class _ConvBase(nn.Module):
    """Parameter layout of ME: `kernel` [kvol, Cin, Cout] ([Cin, Cout] when kvol == 1), `bias` [1, Cout]."""
    _transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, dimension=None):
        super().__init__()
        if dimension != 3:
            raise ImfError("only dimension=3 is implemented")
        if dilation != 1 or kernel_generator is not None or expand_coordinates:
            raise ImfError("dilation / kernel_generator / expand_coordinates are not on IMFNet's path")
        if kernel_size not in (1, 3, 5) or stride not in (1, 2):
            raise ImfError(f"unsupported kernel_size={kernel_size} stride={stride}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dimension = kernel_size, stride, dimension
        self.kernel_volume = kernel_size ** 3
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 \
            else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()
        self._packed = None

    def reset_parameters(self):
        with torch.no_grad():
            n = (self.out_channels if self._transposed else self.in_channels) * self.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def kernel3(self):
        k = self.kernel
        return k.unsqueeze(0) if k.dim() == 2 else k

    def packed(self):
        variant = ops.conv_variant_for(self.kernel_volume)
        tag = (self.kernel.data_ptr(), self.kernel._version, variant)
        if self._packed is None or self._packed[0] != tag:
            self._packed = (tag, ops.pack_weights(self.kernel, variant=variant))
        return self._packed[1]

    def rulebook(self, x):
        cm, ts = x.coordinate_manager, x.coordinate_map_key.tensor_stride
        if self._transposed and self.kernel_volume > 1:
            return cm.transpose_rulebook(ts, self.kernel_size, self.stride), ts // self.stride
        return cm.conv_rulebook(ts, self.kernel_size, self.stride), ts * self.stride

    def run(self, x, in_b=None, scale=None, shift=None, residual=None, relu=False, l2norm=False):
        """Fused convolution on raw feature matrices; returns (features, output tensor stride)."""
        if torch.is_grad_enabled() and (self.kernel.requires_grad or x.F.requires_grad):
            # the FUSED call has no backward (its epilogue folds BatchNorm / ReLU / residual): fail loudly instead of
            # silently cutting the graph.  Differentiable path: forward() -> autograd.SparseConvFunction.
            raise ImfError("the fused convolution call is inference-only: use the module's forward() under autograd "
                           "(model.forward_layers), or run under torch.no_grad()")
        rb, ts_out = self.rulebook(x)
        feat = x.F
        cin = feat.shape[1] + (0 if in_b is None else in_b.shape[1])
        if cin != self.in_channels:
            raise ImfError(f"expected {self.in_channels} input channels, got {cin}")
        if shift is None and self.bias is not None:
            shift = self.bias.detach().reshape(-1)
        elif self.bias is not None:
            raise ImfError("bias + fused shift is not supported")
        if cin <= 4:
            if in_b is not None or residual is not None or l2norm:
                raise ImfError("small-Cin convolution supports only scale/shift/relu epilogues")
            out = ops.spconv_small_cin(feat, self.kernel3(), rb, scale, shift, relu)
        else:
            out = ops.spconv(feat, self.packed(), self.out_channels, rb, in_b=in_b, scale=scale,
                             shift=shift, residual=residual, relu=relu, l2norm=l2norm,
                             variant=ops.conv_variant_for(self.kernel_volume))
        return out, ts_out

    def forward(self, x):
        if torch.is_grad_enabled() and (self.kernel.requires_grad or x.F.requires_grad):
            from .autograd import SparseConvFunction                      # training: differentiable convolution
            _, ts_out = self.rulebook(x)
            out = SparseConvFunction.apply(x.F, self.kernel, self, x)
            if self.bias is not None:
                out = out + self.bias
            return x._like(out, ts_out)
        out, ts_out = self.run(x)
        return x._like(out, ts_out)

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}")


class MinkowskiConvolution(_ConvBase):
    pass


class MinkowskiConvolutionTranspose(_ConvBase):
    _transposed = True


class MinkowskiBatchNorm(nn.Module):
    """Wraps nn.BatchNorm1d as `self.bn` (state_dict keys '*.bn.weight' ...)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(self.bn(x.F))

    def folded(self):
        """Eval-mode affine form y = x*scale + shift."""
        bn = self.bn
        scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias.detach() - bn.running_mean * scale
        return scale.float().contiguous(), shift.float().contiguous()


class MinkowskiInstanceNorm(nn.Module):
    """ME.MinkowskiInstanceNorm (the `IN` blocks of ResUNetIN2*, model/common.py:7-8): per batch item and channel,
    (x - mean) / sqrt(var + 1e-8) over that item's rows (biased variance: ME computes both with a global average pooling),
    then weight * x + bias with [1, C] parameters.  Plain torch on the rows -- these variants are not the checkpoint's
    (ResUNetBN2C) and take the per-layer path, not the fused plan.  [ME 0.5.4 conventions RECALLED: eps, biased variance.]"""

    EPS = 1e-8

    def __init__(self, num_features, dimension=-1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x):
        f = x.F
        item = x.C[:, 0].long()
        n_items = int(item.max().item()) + 1 if item.numel() else 0
        if n_items <= 1:
            mean = f.mean(0, keepdim=True)
            cen = f - mean
            var = (cen * cen).mean(0, keepdim=True)
            out = cen * torch.rsqrt(var + self.EPS)
        else:
            cnt = torch.zeros(n_items, 1, dtype=f.dtype, device=f.device).index_add_(
                0, item, torch.ones(f.shape[0], 1, dtype=f.dtype, device=f.device)).clamp_(min=1)
            mean = torch.zeros(n_items, f.shape[1], dtype=f.dtype, device=f.device).index_add_(0, item, f) / cnt
            cen = f - mean[item]
            var = torch.zeros_like(mean).index_add_(0, item, cen * cen) / cnt
            out = cen * torch.rsqrt(var + self.EPS)[item]
        return x._like(out * self.weight + self.bias)


class MinkowskiFunctional:
    @staticmethod
    def relu(x, *a, **k):
        return x._like(F.relu(x.F))


class utils:
    """ME.utils -- the three helpers the reference calls."""

    @staticmethod
    def sparse_quantize(coordinates, features=None, return_index=False, device="cuda", **_):
        """util/misc.py:83.  Input: already-floored coordinates [N,3] (numpy or tensor); returns the
        unique integer rows and (optionally) the index of each voxel's FIRST point, ascending
        (SURVEY A.1).  Computed on the GPU; results come back as numpy like ME's."""
        c = torch.as_tensor(np.asarray(coordinates) if not torch.is_tensor(coordinates) else coordinates)
        xyz = c.to(device=device, dtype=torch.float64).contiguous()
        lv = ops.voxelize(xyz, 1.0, 0)
        ops.sync_levels([lv])
        coords = lv.coords[:, 1:].cpu().numpy()
        inds = lv.first_idx.cpu().numpy().astype(np.int64)
        if features is not None:
            f = features[inds]
            return (coords, f, inds) if return_index else (coords, f)
        return (coords, inds) if return_index else coords

    @staticmethod
    def batched_coordinates(coords_list, dtype=torch.int32, device=None):
        rows = []
        for b, c in enumerate(coords_list):
            c = torch.as_tensor(np.asarray(c) if not torch.is_tensor(c) else c).to(dtype)
            rows.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=dtype), c], dim=1))
        out = torch.cat(rows, 0)
        return out if device is None else out.to(device)

    @staticmethod
    def fnv_hash_vec(arr):
        """FNV-1a-64 over the columns (scripts/evaluation_3dmatch.py:164-168)."""
        a = np.asarray(arr).astype(np.uint64, copy=True)
        h = np.full(a.shape[0], np.uint64(14695981039346656037), dtype=np.uint64)
        for j in range(a.shape[1]):
            h *= np.uint64(1099511628211)
            h = np.bitwise_xor(h, a[:, j])
        return h
