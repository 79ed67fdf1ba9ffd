"""IMFNet's sparse ResUNet with bottleneck image fusion, executed on MI355X.

Interface parity with the reference's model/resunet.py:
  * class names `ResUNet2`, `ResUNetBN2[B-E]`, `ResUNetIN2[B-E]` with the same CHANNELS /
    TR_CHANNELS tables (resunet.py:276-326);
  * constructor `(in_channels, out_channels, bn_momentum, normalize_feature, conv1_kernel_size,
    D, config)` (resunet.py:25-32) and sub-module names, hence the 361-key state_dict of SURVEY
    App. B loads with strict=True;
  * `forward(x, image) -> SparseTensor` with `.F` [M,out] row-aligned with the input coordinates
    (resunet.py:163-235) and `transformer(images, F, xyz)` (resunet.py:237-273).

Execution is NOT a layer-by-layer walk: in eval mode `forward` runs the fused plan
  geometry (hash pyramid + 8 rulebooks, built once per fragment in HBM)
  -> 23 sparse-conv launches whose epilogues carry BatchNorm (folded), ReLU, residual add,
     the three ME.cat's (two-source gather), the `final` bias and the L2 normalisation.
`forward_layers` keeps the reference's op-by-op order for tests and training-mode statistics.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F_

from .. import ops
from .. import sparse as ME
from .fusion import AttentionFusion
from .image_encoder import ImageEncoder
from .layers import get_block, get_norm

MEF = ME.MinkowskiFunctional


def _down8(h):
    """Height / width of the stride-8 map (conv7x7/2 pad 3, maxpool3/2 pad 1, conv3x3/2 pad 1)."""
    for k, p in ((7, 3), (3, 1), (3, 1)):
        h = (h + 2 * p - k) // 2 + 1
    return h


def _lib_max_batch():
    from .._lib import MAX_BATCH
    return MAX_BATCH


class ResUNet2(ME.MinkowskiNetwork):
    NORM_TYPE = None
    BLOCK_NORM_TYPE = 'BN'
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 32, 64, 64, 128]
    IMG_CHANNELS = [None, 0, 0, 0, 0]       # multi-scale fusion is disabled upstream (resunet.py:20-21)

    def __init__(self, in_channels=3, out_channels=32, bn_momentum=0.1, normalize_feature=None,
                 conv1_kernel_size=None, D=3, config=None):
        super().__init__(D)
        C, T, I = self.CHANNELS, self.TR_CHANNELS, self.IMG_CHANNELS
        self.normalize_feature = normalize_feature
        conv, conv_tr = ME.MinkowskiConvolution, ME.MinkowskiConvolutionTranspose

        def stage(idx, suffix, cls, cin, cout, ksize, stride):
            setattr(self, f'conv{idx}{suffix}', cls(in_channels=cin, out_channels=cout, kernel_size=ksize,
                                                    stride=stride, dilation=1, bias=False, dimension=D))
            setattr(self, f'norm{idx}{suffix}', get_norm(self.NORM_TYPE, cout, bn_momentum=bn_momentum, D=D))
            setattr(self, f'block{idx}{suffix}',
                    get_block(self.BLOCK_NORM_TYPE, cout, cout, bn_momentum=bn_momentum, D=D))

        # encoder: conv(k5|k3, stride 1|2) - norm - residual block, tensor stride 1 -> 8
        stage(1, '', conv, in_channels, C[1], conv1_kernel_size, 1)
        stage(2, '', conv, C[1], C[2], 3, 2)
        stage(3, '', conv, C[2], C[3], 3, 2)
        stage(4, '', conv, C[3], C[4], 3, 2)
        # bottleneck: image tokens (128-d) attend into the stride-8 point features (resunet.py:91-100)
        self.attention_fusion = AttentionFusion(dim=128, depth=0, latent_dim=C[4], cross_heads=1,
                                                latent_heads=8, cross_dim_head=C[4] // 2,
                                                latent_dim_head=C[4] // 2)
        # decoder: transposed conv (stride 2) - norm - block, skip concatenation before each
        stage(4, '_tr', conv_tr, C[4], T[4], 3, 2)
        stage(3, '_tr', conv_tr, C[3] + T[4] + I[1], T[3], 3, 2)
        stage(2, '_tr', conv_tr, C[2] + T[3] + I[2], T[2], 3, 2)
        self.conv1_tr = conv(in_channels=C[1] + T[2] + I[3], out_channels=T[1], kernel_size=1, stride=1,
                             dilation=1, bias=False, dimension=D)
        self.final = conv(in_channels=T[1], out_channels=out_channels, kernel_size=1, stride=1,
                          dilation=1, bias=True, dimension=D)
        self.img_encoder = ImageEncoder()
        self._folded = None
        self._plan = None                 # arena executor (model/plan.py), built lazily in eval mode
        self._native_plan = None          # its native twin: one C call per fragment (csrc/executor.hip)
        self._pending_image = None        # (image, features, kv, event, packed K/V) queued by start_image_branch
        self._kv_packed = {}              # (device, image shape) -> packed K^T / V buffers
        self._fuse_done = None            # event: last fusion finished reading the image branch outputs
        self.after_fusion_hook = None     # one-shot callable run once the bottleneck fusion is queued
        self._side = {}                   # device -> side stream
        self._img_graph = {}              # (device, shape) -> captured image branch
        self._fw = None                   # packed weights of the fused fusion kernel
        self._img_plan = None             # native image branch (model/image_plan.py, csrc/image.hip)
        self._runner = None               # whole-fragment capacity-mode / hipGraph runner (model/graph.py)
        self._flag_words = {}             # device -> int32[1]: IMF_FLAG_* bits OR-ed by the kernels (sticky)
        self._fp_tensors, self._fp_value = None, None   # parameter / buffer version fingerprint (see _stale)
        self.image_branch_mode = None     # how the last image branch ran: native-hip | torch-graph | torch-eager

    # ---- folded BatchNorm cache (eval) ----------------------------------------------------------
    def _stale(self):
        """In-place edits of parameters / buffers (optimizer step in eval mode, submodule.load_state_dict,
        param.data.copy_, BatchNorm statistics set by hand) bump the tensors' version counters: the folded
        BatchNorm terms, packed weights and plans (which hold raw pointers) are rebuilt when the sum moves."""
        ts = self._fp_tensors
        if ts is None:
            ts = self._fp_tensors = list(self.parameters()) + list(self.buffers())
        v = 0
        for t in ts:
            v += t._version
        if v != self._fp_value:
            stale = self._fp_value is not None
            self._fp_value = v
            return stale
        return False

    def invalidate(self):
        """Drop every derived copy of the parameters (folded BatchNorm, packed weights, plans, captured graphs).
        Needed only after edits torch cannot see -- `param.data.<op>_()`; everything else is detected (see _stale)."""
        self._invalidate()

    def _refresh(self):
        # (packed weights, plans and runners are built for ONE arithmetic: a changed ops.CONV_VARIANT rebuilds them too)
        # (`pinned_variant`: a model built for one arithmetic beside the process default -- bench.build_model(variant=...))
        want = self.effective_variant()
        if self._stale() or getattr(self, "_built_variant", want) != want:
            self._invalidate()

    pinned_variant = None

    def effective_variant(self):
        """The arithmetic THIS model's packed weights, plans and flag semantics belong to: its pinned one, else the process
        default (the flag mask and the fusion images followed the global even for a pinned model)."""
        return self.pinned_variant if self.pinned_variant is not None else ops.CONV_VARIANT

    def _invalidate(self):
        self._built_variant = self.effective_variant()
        self._plan = None
        self._native_plan = None
        self._folded = None
        self._pending_image = None
        self._img_graph = {}
        self._fw = None
        self._img_plan = None
        self._runner = None

    def _apply(self, fn, *a, **k):
        self._invalidate()
        self._fp_tensors = self._fp_value = None      # .to() / .cuda() replace the tensors
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        if bool(mode) != self.training:               # extract_features calls model.eval() per fragment:
            self._invalidate()                        # only a real mode change drops the plan / image graph
        return super().train(mode)

    # ---- image branch: independent of the sparse encoder until the bottleneck ------------------
    _warned_grad = False

    def _image_branch(self, image):
        """Image encoder + the context half of the cross attention (LayerNorm + K/V projection of
        the image tokens): everything that depends on the image only."""
        # MIOpen picks atomically-accumulating algorithms for batch > 1 unless told not to: run-to-run 1e-5
        # differences in the image features; the descriptor path is bit-reproducible everywhere else
        prev, torch.backends.cudnn.deterministic = torch.backends.cudnn.deterministic, True
        try:
            feat = self.img_encoder(image)
        finally:
            torch.backends.cudnn.deterministic = prev
        kv = kt = vp = None
        blk = self.attention_fusion.cross_attend_blocks[0]
        if blk.fn.heads == 1 and len(self.attention_fusion.layers) == 0:
            tokens = feat.flatten(2).transpose(1, 2)                          # [B, H*W, C]
            kv = blk.fn.to_kv(blk.norm_context(tokens))                       # [B, T, 2*d]
            T, d = kv.shape[1], kv.shape[2] // 2
            tp = (T + 63) // 64 * 64
            if feat.shape[0] <= _lib_max_batch() and tp <= 320:
                # zero-padded K^T [B, d, tp] and V [B, tp, d] for the fused fusion kernel (packed right after)
                B = feat.shape[0]
                kt = torch.zeros((B, d, tp), dtype=kv.dtype, device=kv.device)
                kt[:, :, :T] = kv[:, :, :d].transpose(1, 2)
                vp = torch.zeros((B, tp, d), dtype=kv.dtype, device=kv.device)
                vp[:, :T] = kv[:, :, d:]
        return feat, kv, kt, vp

    def start_image_branch(self, image, device=None, inputs_ready=False):
        """Queue the image branch on a side HIP stream (as a captured hipGraph when the shape is
        static) so it overlaps the geometry build and the sparse encoder.  forward() collects it.
        `image`: [B,3,H,W] host array / CPU tensor (uploaded on the side stream itself) or a device
        tensor; `inputs_ready=True` promises a device tensor is complete, so the branch need not wait
        for the main stream.  Returns the device tensor to hand to forward()."""
        if not self._can_fuse():
            return None
        self._refresh()
        on_device = torch.is_tensor(image) and image.is_cuda
        dev = image.device if on_device else torch.device(device if device is not None else "cuda")
        side = ops.aux_streams(dev)[1][2]        # the package's image-branch stream (ops.aux_streams)
        cur = torch.cuda.current_stream(dev)
        if on_device and not inputs_ready:
            side.wait_stream(cur)
        elif self._fuse_done is not None:
            side.wait_event(self._fuse_done)              # previous fragment still reads the static outputs
        with torch.cuda.stream(side), torch.no_grad():
            if not on_device:
                image = torch.as_tensor(image, dtype=torch.float32).to(dev, non_blocking=True)
            plan = None if os.environ.get("IMFNET_TORCH_IMAGE") else self._native_image()
            packed = kt = None
            if (plan is not None and plan.supported and image.dtype == torch.float32 and image.dim() == 4 and
                    image.shape[0] <= _lib_max_batch() and image.shape[2] >= 8 and image.shape[3] >= 8):
                # ~22 launches of the sparse-conv kernel over static pixel tables (csrc/image.hip)
                rows, packed = plan.run(image.contiguous(), want_kv=self._fusion_weights().supported,
                                        flags=self.flag_word(dev))
                B, h8, w8 = image.shape[0], _down8(image.shape[2]), _down8(image.shape[3])
                feat, kv = rows.view(B, h8, w8, rows.shape[1]).permute(0, 3, 1, 2), None
                self.image_branch_mode = "native-hip"
            else:
                feat, kv, kt, vp = self._run_image_graph(image, side)
            if packed is None and kt is not None and self._fusion_weights().supported:
                key = (dev, tuple(image.shape))
                bufs = self._kv_packed.get(key)
                B = kt.shape[0]
                if bufs is None:
                    bufs = self._kv_packed[key] = ([torch.empty(kt[0].numel(), dtype=torch.float32, device=dev) for _ in range(B)],
                                                   [torch.empty(vp[0].numel(), dtype=torch.float32, device=dev) for _ in range(B)])
                for b in range(B):                         # fragment-major K^T / V per image, on the side stream
                    ops.pack_weights(kt[b], out=bufs[0][b])
                    ops.pack_weights(vp[b], out=bufs[1][b])
                packed = (bufs[0], bufs[1], kv.shape[1], kt.shape[2])
            ev = torch.cuda.Event()
            ev.record(side)
        image.record_stream(side)
        self._pending_image = (image, feat, kv, ev, packed)
        return image

    # ---- f16 range guard of the split-f16 convolutions -------------------------------------------------
    def flag_word(self, device):
        """Device int32[1] the kernels OR IMF_FLAG_RANGE (32) into when an activation that feeds a split-f16
        convolution is NaN or >= 65504 in magnitude (it would become inf as an f16 operand)."""
        device = _norm_device(device)
        w = self._flag_words.get(device)
        if w is None:
            w = self._flag_words[device] = torch.zeros(1, dtype=torch.int32, device=device)
        return w

    def take_flags(self, device):
        """Read and clear the flag word (one 4-byte readback: synchronises with the current stream).  IMF_FLAG_RANGE is
        only meaningful while the convolutions run on split-f16 operands (variant 6): the fp32 / bf16x3 variants carry any
        fp32 value, so the bit (which the fusion kernel raises in every mode) is dropped for them."""
        w = self._flag_words.get(_norm_device(device))
        if w is None:
            return 0
        v = int(w.item())
        if v:
            w.zero_()
        if self.effective_variant() != 6:
            v &= ~32                                   # IMF_FLAG_RANGE (_lib.FLAG_RANGE)
        return v

    def forward_fp32(self, fn):
        """Run `fn()` with every convolution on the true-fp32 matrix instructions (variant 0): the fallback for
        activations outside the f16 range.  Plans and packed weights are rebuilt on entry and exit."""
        prev = ops.CONV_VARIANT
        ops.CONV_VARIANT = 0
        self._invalidate()
        try:
            out = fn()
            torch.cuda.synchronize()
            return out
        finally:
            ops.CONV_VARIANT = prev
            self._invalidate()
            for w in self._flag_words.values():
                w.zero_()

    def fragment_runner(self):
        """The capacity-mode / hipGraph executor of whole fragments (model/graph.py), or None when this model
        configuration is not covered (training mode, non-BatchNorm blocks, per-point input features, ...)."""
        if os.environ.get("IMFNET_NO_FRAGMENT_GRAPH") or not self._can_fuse():
            return None
        self._refresh()
        if self._runner is None:
            try:
                if next(self.parameters()).device.type != "cuda":
                    return None
                from .graph import FragmentRunner
                r = FragmentRunner(self)
                self._runner = r if r.supported else False
            except ME.ImfError:
                self._runner = False
        return self._runner or None

    def _native_image(self):
        if self._img_plan is None:
            from .image_plan import ImagePlan
            blk = self.attention_fusion.cross_attend_blocks[0]
            one_head = blk.fn.heads == 1 and len(self.attention_fusion.layers) == 0
            self._img_plan = ImagePlan(self.img_encoder, blk if one_head else None, ops.conv_variant_for(9))
            torch.cuda.synchronize()                 # once per model: packed weights visible to every stream
        return self._img_plan

    def _run_image_graph(self, image, side):
        key = (image.device, tuple(image.shape))
        g = self._img_graph.get(key)
        if g is None:
            g = self._img_graph[key] = self._capture_image_graph(image, side)
        if g is False:                                   # capture unavailable: eager on the side stream
            self.image_branch_mode = "torch-eager"
            return self._image_branch(image)
        self.image_branch_mode = "torch-graph"
        graph, static_in, outs = g
        static_in.copy_(image)
        graph.replay()
        return outs

    def _capture_image_graph(self, image, side):
        if os.environ.get("IMFNET_NO_GRAPH"):
            return False
        # diagnostic path (IMFNET_TORCH_IMAGE=1; the default image branch is csrc/image.hip): a failed capture raises --
        # nothing here falls back silently; IMFNET_NO_GRAPH=1 asks for eager launches explicitly
        static_in = image.clone()
        for _ in range(3):                               # warm-up: MIOpen / hipBLASLt pick their kernels
            self._image_branch(static_in)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            outs = self._image_branch(static_in)
        return graph, static_in, outs

    def _bn(self):
        if self._folded is None:
            with torch.no_grad():
                self._folded = {name: m.folded() for name, m in self.named_modules()
                                if isinstance(m, ME.MinkowskiBatchNorm)}
        return self._folded

    def _can_fuse(self):
        return (not self.training and self.NORM_TYPE == 'BN' and self.BLOCK_NORM_TYPE == 'BN')

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x, image):
        if not self._can_fuse() or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            if self._can_fuse() and not ResUNet2._warned_grad:
                ResUNet2._warned_grad = True
                import warnings
                warnings.warn("imfnet_amd: eval-mode forward called with autograd enabled and trainable parameters: running "
                              "the per-layer training path (10-100x slower than the fused inference plan); wrap the call in "
                              "torch.no_grad() for descriptor extraction")
            return self.forward_layers(x, image)          # training mode, or fine-tuning with frozen statistics
        if self._pending_image is None:
            self._refresh()
        bn = self._bn()
        pend, self._pending_image = self._pending_image, None
        kv = None
        if pend is None or pend[0] is not image:
            self.start_image_branch(image)
            pend, self._pending_image = self._pending_image, None
        _, image_feat, kv, ev, packed = pend               # queued on the side stream
        if image_feat.device != x.F.device:
            raise ME.ImfError("image and sparse tensor live on different devices")

        # rows of every batch item at the bottleneck: known for pyramids built by imf_pyramid_build(_batched)
        lv8 = x.coordinate_manager.level(8)
        items = getattr(lv8, "items", None)
        if items is None and image_feat.shape[0] == 1:
            items = [(0, lv8.n)]
        if packed is not None and (items is None or len(items) != len(packed[0])):
            packed = None                                  # batched tensor built from raw coordinates: torch path

        def fuse(f8):                                                                 # :189
            cur = torch.cuda.current_stream(f8.device)
            cur.wait_event(ev)                            # join the image branch
            image_feat.record_stream(cur)
            if packed is not None:                         # one HIP kernel: attention + GEGLU feed-forward
                for t in packed[0] + packed[1]:
                    t.record_stream(cur)
                ev_ = self.effective_variant()
                out = ops.fusion_attention_batched(f8, items, packed[0], packed[1], packed[2], packed[3],
                                                   self._fusion_weights(), flags=self.flag_word(f8.device),
                                                   variant=ev_ if ev_ in (6, 3) else 0)
            elif kv is not None and image_feat.shape[0] == 1:
                kv.record_stream(cur)
                out = self._fusion_fast(f8, kv[0])
            else:
                out = self.transformer(images=image_feat, F=f8, xyz=x.coordinate_manager.coords(8))
            self._fuse_done = torch.cuda.Event()
            self._fuse_done.record(cur)
            return out

        if self._plan is None:
            from .plan import FusedPlan
            self._plan = FusedPlan(self)
            torch.cuda.synchronize()                 # once per model: packed weights visible to every stream
        hook, self.after_fusion_hook = self.after_fusion_hook, None      # one-shot
        native = (packed is not None and hook is None and x.F.is_cuda and
                  not os.environ.get("IMFNET_PYTHON_EXECUTOR"))
        if native:                                      # one C call: csrc/executor.hip
            if self._native_plan is None:
                from .plan import NativePlan
                self._native_plan = NativePlan(self, self._plan)
                torch.cuda.synchronize()
            cur = torch.cuda.current_stream(x.F.device)
            image_feat.record_stream(cur)
            for t in packed[0] + packed[1]:
                t.record_stream(cur)
            self._fuse_done = torch.cuda.Event()
            return x._like(self._native_plan.run(x, packed, items, ev, self._fuse_done))
        return x._like(self._plan.run(x, fuse, hook, n_items=len(items) if items else int(image_feat.shape[0])))

    def forward_layers(self, x, image):
        """Op-by-op order of the reference's forward (resunet.py:163-235)."""
        image = self.img_encoder(image)
        out_s1 = self.block1(self.norm1(self.conv1(x)))
        out = MEF.relu(out_s1)
        out_s2 = self.block2(self.norm2(self.conv2(out)))
        out = MEF.relu(out_s2)
        out_s4 = self.block3(self.norm3(self.conv3(out)))
        out = MEF.relu(out_s4)
        out_s8 = self.block4(self.norm4(self.conv4(out)))
        out = MEF.relu(out_s8)
        out._F = self.transformer(images=image, F=out.F, xyz=out.C)
        out = MEF.relu(self.block4_tr(self.norm4_tr(self.conv4_tr(out))))
        out = ME.cat(out, out_s4)
        out = MEF.relu(self.block3_tr(self.norm3_tr(self.conv3_tr(out))))
        out = ME.cat(out, out_s2)
        out = MEF.relu(self.block2_tr(self.norm2_tr(self.conv2_tr(out))))
        out = ME.cat(out, out_s1)
        out = MEF.relu(self.conv1_tr(out))
        out = self.final(out)
        if self.normalize_feature:
            return out._like(out.F / torch.norm(out.F, p=2, dim=1, keepdim=True))
        return out

    def _fusion_weights(self):
        self._refresh()
        if self._fw is None:
            self._fw = ops.FusionKernelWeights(self.attention_fusion, variant=self.effective_variant())
        return self._fw

    def _fusion_fast(self, x, kv):
        """attention_fusion.py:132-154 for one image, single head, depth 0, with the image-only half
        (K, V) already computed by the image branch.  x [N,256], kv [T,256] -> [N,256]."""
        blk0, blk1 = self.attention_fusion.cross_attend_blocks
        att, ff = blk0.fn, blk1.fn.net
        d = att.dim_head
        q = F_.linear(blk0.norm(x), att.to_q.weight)
        p = torch.softmax((q @ kv[:, :d].t()) * att.scale, dim=-1)
        x = F_.linear(p @ kv[:, d:], att.to_out.weight, att.to_out.bias) + x
        h = F_.linear(blk1.norm(x), ff[0].weight, ff[0].bias)
        a, g = h.chunk(2, dim=-1)
        return F_.linear(a * F_.gelu(g), ff[2].weight, ff[2].bias) + x

    def transformer(self, images, F, xyz):
        """Per batch item: the item's stride-8 rows attend over that item's image tokens
        (resunet.py:237-273).  Rows are grouped by batch index.  With one image there is no
        device->host traffic (the reference syncs three times here)."""
        tokens = images.flatten(2).transpose(1, 2)                    # [B, H*W, C]  (:257-261)
        if images.shape[0] == 1:
            return self.attention_fusion(tokens, queries_encoder=F.unsqueeze(0))[0]
        counts = torch.bincount(xyz[:, 0].long(), minlength=images.shape[0]).cpu().tolist()
        parts, start = [], 0
        for b, n in enumerate(counts):
            parts.append(self.attention_fusion(tokens[b:b + 1], queries_encoder=F[start:start + n].unsqueeze(0))[0])
            start += n
        return torch.cat(parts, dim=0)


def _norm_device(device):
    """`cuda` and `cuda:<current>` are the same device but different dict keys: always carry the index."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def _variant(name, base, **attrs):
    return type(name, (base,), dict(attrs, __doc__=f"{name}: see model/resunet.py:276-326 of the reference."))


ResUNetBN2 = _variant('ResUNetBN2', ResUNet2, NORM_TYPE='BN')
ResUNetBN2B = _variant('ResUNetBN2B', ResUNet2, NORM_TYPE='BN', TR_CHANNELS=[None, 64, 64, 64, 64])
ResUNetBN2C = _variant('ResUNetBN2C', ResUNet2, NORM_TYPE='BN', TR_CHANNELS=[None, 64, 64, 64, 128])
ResUNetBN2D = _variant('ResUNetBN2D', ResUNet2, NORM_TYPE='BN', TR_CHANNELS=[None, 64, 64, 128, 128])
ResUNetBN2E = _variant('ResUNetBN2E', ResUNet2, NORM_TYPE='BN', CHANNELS=[None, 128, 128, 128, 256],
                       TR_CHANNELS=[None, 64, 128, 128, 128])
ResUNetIN2 = _variant('ResUNetIN2', ResUNet2, NORM_TYPE='BN', BLOCK_NORM_TYPE='IN')
ResUNetIN2B = _variant('ResUNetIN2B', ResUNetBN2B, BLOCK_NORM_TYPE='IN')
ResUNetIN2C = _variant('ResUNetIN2C', ResUNetBN2C, BLOCK_NORM_TYPE='IN')
ResUNetIN2D = _variant('ResUNetIN2D', ResUNetBN2D, BLOCK_NORM_TYPE='IN')
ResUNetIN2E = _variant('ResUNetIN2E', ResUNetBN2E, BLOCK_NORM_TYPE='IN')
