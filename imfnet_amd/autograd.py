"""Training backward of the sparse convolutions (last item; the reference trains through
MinkowskiEngine's autograd, lib/trainer.py:495-569).

`SparseConvFunction` wraps one ME.MinkowskiConvolution / ConvolutionTranspose:
  forward   imf_spconv_fwd (the inference kernel, no fused epilogue -- BatchNorm / ReLU / residual stay torch ops in
            training mode, so their gradients are torch's);
  d input   imf_spconv_fwd again over the OPPOSITE kernel map with transposed weights:
              stride 1      the same map, W'[k] = W[K-1-k]^T          (offsets are symmetric: off[K-1-k] = -off[k])
              stride 2      the transposed map coarse -> fine, W'[k] = W[k]^T
              transposed    the strided map fine -> coarse, W'[k] = W[k]^T
              1x1x1         grad @ W^T
  d kernel  imf_spconv_wgrad (csrc/backward.hip): per offset the sum of in[i]^T grad[o] over the map's pairs.
Everything stays on the GPU; the maps come from the coordinate manager's cache (built once per fragment).
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import ImfError, check


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _opposite_rulebook(cm, ts_in, ksize, stride, transposed):
    """Kernel map that carries gradients from the convolution's outputs back to its inputs."""
    if ksize == 1:
        return None
    if not transposed:
        if stride == 1:
            return cm.conv_rulebook(ts_in, ksize, 1)                  # symmetric: same map, flipped offsets
        return cm.transpose_rulebook(ts_in * stride, ksize, stride)  # outputs live at ts_in * stride: coarse -> fine
    return cm.conv_rulebook(ts_in // stride, ksize, stride)          # transposed conv: fine (outputs) -> coarse (inputs)


def spconv_wgrad(feat, grad_out, rb, kvol):
    """dW [kvol, cin, cout] of out = spconv(feat, W, rb)."""
    L = _lib.lib()
    cin, cout = feat.shape[1], grad_out.shape[1]
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=feat.device)
    nbytes = L.imf_spconv_wgrad_workspace_bytes(rb.n_slots, kvol, cin, cout)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=feat.device)
    check(L.imf_spconv_wgrad(feat.data_ptr(), cin, grad_out.data_ptr(), cout,
                             None if rb.tile_rows is None else rb.tile_rows.data_ptr(),
                             None if rb.nbr is None else rb.nbr.data_ptr(), rb.n_slots, rb.n_out, kvol, dw.data_ptr(),
                             ws.data_ptr(), nbytes, _stream()), "imf_spconv_wgrad")
    return dw


class SparseConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, kernel, module, x):
        rb, _ = module.rulebook(x)
        k3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
        feat_c = feat.detach().contiguous()
        if module.in_channels <= 4:
            out = ops.spconv_small_cin(feat_c, k3.detach(), rb)
        else:
            variant = ops.conv_variant_for(module.kernel_volume)
            out = ops.spconv(feat_c, ops.pack_weights(k3.detach(), variant=variant), module.out_channels, rb,
                             variant=variant)
        ctx.save_for_backward(feat_c, kernel)
        ctx.module, ctx.x, ctx.rb = module, x, rb
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feat, kernel = ctx.saved_tensors
        m, x, rb = ctx.module, ctx.x, ctx.rb
        g = grad_out.contiguous().float()
        K = m.kernel_volume
        k3 = kernel.detach() if kernel.dim() == 3 else kernel.detach().unsqueeze(0)
        grad_feat = grad_kernel = None
        if ctx.needs_input_grad[0]:
            if m.in_channels % 32 or m.out_channels % 32:
                raise ImfError("input gradient needs channel counts that are multiples of 32 (the first layer's input "
                               "features do not require grad on IMFNet's path)")
            cm, ts = x.coordinate_manager, x.coordinate_map_key.tensor_stride
            if K == 1:
                grad_feat = g @ k3[0].t()
            else:
                rbt = _opposite_rulebook(cm, ts, m.kernel_size, m.stride, m._transposed)
                wt = k3.transpose(1, 2)
                if not m._transposed and m.stride == 1:
                    wt = wt.flip(0)
                variant = ops.conv_variant_for(K)
                grad_feat = ops.spconv(g, ops.pack_weights(wt.contiguous(), variant=variant), m.in_channels, rbt,
                                       variant=variant)
        if ctx.needs_input_grad[1]:
            dw = spconv_wgrad(feat, g, rb, K)
            grad_kernel = dw if kernel.dim() == 3 else dw[0]
        return grad_feat, grad_kernel, None, None
