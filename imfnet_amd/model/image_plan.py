"""Native image branch: the truncated ResNet-34 trunk + the K/V projection of the image tokens as ONE
library call (`imf_image_branch`, csrc/image.hip) on the sparse-convolution kernel.

Built once per model in eval mode from the torch modules that carry the checkpoint's parameters
(`img_encoder.backbone.*`, `attention_fusion.cross_attend_blocks.0.{norm_context,fn.to_kv}`):
BatchNorm folded to scale / shift (model/resnet.py:54-66 in eval mode), conv weights re-laid
[ky*3+kx][ci][co] and packed for the chosen kernel variant.  Per (device, image shape): the static pixel
tables + feature workspace and the output buffers.  Reference: model/resnet.py:195-216,
model/attention_fusion.py:36-46,84.
"""
import ctypes as C
import os

import torch

from .. import _lib, ops
from .._lib import ImageDesc, check


def _fold(bn):
    scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)).contiguous()
    shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
    return scale, shift


class ImagePlan:
    def __init__(self, img_encoder, cross_block, variant):
        self.L = _lib.lib()
        bb = img_encoder.backbone
        self.supported = (len(bb.layer1) == 3 and len(bb.layer2) == 4 and bb.conv1.weight.is_cuda and
                          bb.conv1.weight.shape == (64, 3, 7, 7) and bb.layer2[0].downsample is not None and
                          all(b.downsample is None for b in list(bb.layer1) + list(bb.layer2)[1:]))
        if not self.supported:
            return
        self.variant = int(variant)
        self._keep = []
        d = self.desc = ImageDesc()
        d.variant = self.variant

        def keep(t):
            self._keep.append(t)
            return t.data_ptr()

        def pack(w):                                   # torch [co,ci,kh,kw] -> [kh*kw, ci, co] -> fragment-major
            co, ci, kh, kw = w.shape
            k = w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw, ci, co).contiguous()
            return keep(ops.pack_weights(k, variant=self.variant))

        w = bb.conv1.weight.detach().float().permute(2, 3, 1, 0).reshape(147, 64)
        stem = torch.zeros((1, 160, 64), dtype=torch.float32, device=w.device)
        stem[0, :147] = w
        d.stem_w = keep(ops.pack_weights(stem, variant=self.variant))
        sc, sh = _fold(bb.bn1)
        d.stem_scale, d.stem_shift = keep(sc), keep(sh)

        def entry(i, conv, bn, relu):
            c = d.conv[i]
            c.w_packed = pack(conv.weight)
            c.kvol, c.cin, c.cout = conv.kernel_size[0] * conv.kernel_size[1], conv.in_channels, conv.out_channels
            s, b = _fold(bn)
            c.scale, c.shift = keep(s), keep(b)
            c.relu, c.l2norm, c.variant = int(relu), 0, self.variant

        i = 0
        for blk in bb.layer1:
            entry(i, blk.conv1, blk.bn1, True)
            entry(i + 1, blk.conv2, blk.bn2, True)      # relu after the residual add
            i += 2
        b0 = bb.layer2[0]
        entry(6, b0.conv1, b0.bn1, True)
        entry(7, b0.downsample[0], b0.downsample[1], False)
        entry(8, b0.conv2, b0.bn2, True)
        i = 9
        for blk in list(bb.layer2)[1:]:
            entry(i, blk.conv1, blk.bn1, True)
            entry(i + 1, blk.conv2, blk.bn2, True)
            i += 2
        self.with_kv = cross_block is not None
        if self.with_kv:
            nc, att = cross_block.norm_context, cross_block.fn
            self.with_kv = (nc is not None and nc.normalized_shape == (128,) and abs(nc.eps - 1e-5) < 1e-12 and
                            att.to_kv.weight.shape == (256, 128) and att.to_kv.bias is None)
        if self.with_kv:
            d.ln_g, d.ln_b = keep(nc.weight.detach().float().contiguous()), keep(nc.bias.detach().float().contiguous())
            d.kv_w = keep(ops.pack_weights(att.to_kv.weight.detach().float().t().contiguous().unsqueeze(0), variant=self.variant))
        self._shapes = {}

    def buffers(self, dev, B, H, W, private=False):
        """Per (device, shape): workspace with the static tables built, and the output buffers.  `private`: a fresh
        set the caller owns (a graph bucket's), not the shared one of the eager path."""
        key = (dev, B, H, W)
        b = None if private else self._shapes.get(key)
        if b is None:
            L = self.L
            nbytes = L.imf_image_workspace_bytes(B, H, W)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            check(L.imf_image_tables_build(B, H, W, ws.data_ptr(), nbytes, stream), "imf_image_tables_build")
            T = L.imf_image_tokens(H, W)
            tp = (T + 63) // 64 * 64
            feat = torch.empty((B * T, 128), dtype=torch.float32, device=dev)
            kt = torch.empty(B * 128 * tp, dtype=torch.float32, device=dev)
            vp = torch.empty(B * 128 * tp, dtype=torch.float32, device=dev)
            per = 128 * tp
            if os.environ.get("IMF_POISON"):           # debugging aid: a read-before-write shows up as NaN
                for t in (feat, kt, vp):
                    t.fill_(float("nan"))
                ws.view(torch.float32)[L.imf_image_workspace_bytes(B, H, W) // 8:].fill_(float("nan"))
            b = dict(ws=ws, nbytes=nbytes, T=T, tp=tp, feat=feat, kt=kt, vp=vp,
                     kt_items=[kt[i * per:(i + 1) * per] for i in range(B)],
                     vp_items=[vp[i * per:(i + 1) * per] for i in range(B)])
            if not private:
                self._shapes[key] = b
        return b

    def run(self, image, want_kv=True, flags=None):
        """image: CUDA float32 [B,3,H,W] (contiguous).  Launches on the CURRENT stream.  Returns
        (feat [B*T,128] NHWC rows, packed) with packed = ([K^T per item], [V per item], T, tokens_padded) or None."""
        B, _, H, W = image.shape
        b = self.buffers(image.device, B, H, W)
        kv = want_kv and self.with_kv and b["tp"] <= 320
        check(self.L.imf_image_branch(C.byref(self.desc), image.data_ptr(), B, H, W, b["ws"].data_ptr(), b["nbytes"],
                                      b["feat"].data_ptr(), b["kt"].data_ptr() if kv else None,
                                      b["vp"].data_ptr() if kv else None, b["tp"],
                                      None if flags is None else flags.data_ptr(),
                                      torch.cuda.current_stream(image.device).cuda_stream), "imf_image_branch")
        packed = (b["kt_items"], b["vp_items"], b["T"], b["tp"]) if kv else None
        return b["feat"], packed
