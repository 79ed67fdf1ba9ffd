"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy + torch-CPU) of IMFNet's
descriptor-generation hot path.  Nothing under imfnet_amd/ may import this file;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only
as the checker.

Parity status
-------------
* voxelize(): PINNED by the reference's own artefact files/3D_head_map.ply
  (tests/test_oracle_golden.py::test_voxelize_matches_reference_head_map).
* model wiring (resunet / residual_block / attention_fusion / resnet): PINNED
  against the reference's own model/*.py executed verbatim over
  oracle/me_shim (tests/golden/gen_golden.py writes the vectors).
* sparse-conv primitive semantics (kernel-offset order, strided / transposed
  kernel maps): these live in MinkowskiEngine 0.5.4 (requirements.txt:5), which is
  NOT under /root/reference and is not installable here.  They are restated from
  its published algorithm (kernel_region.hpp: axis 0 fastest; coordinate_map
  stride(); kernel_map() swap for transposed conv).  For those primitives the
  oracle is "parity unpinned" by the reference -- see DESIGN.md §Oracle.

All file:line citations are into /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------

_BITS = 18                      # per spatial coordinate (two's complement)
_BMASK = (1 << _BITS) - 1


def pack_keys(c4):
    """(b,x,y,z) int rows -> unique int64 key (b:9 | x:18 | y:18 | z:18)."""
    c4 = np.asarray(c4, dtype=np.int64)
    lim = 1 << (_BITS - 1)
    assert c4.shape[1] == 4
    assert (c4[:, 1:] >= -lim).all() and (c4[:, 1:] < lim).all(), "coordinate out of +-2^17"
    assert (c4[:, 0] >= 0).all() and (c4[:, 0] < 512).all()
    return ((c4[:, 0] << (3 * _BITS)) | ((c4[:, 1] & _BMASK) << (2 * _BITS))
            | ((c4[:, 2] & _BMASK) << _BITS) | (c4[:, 3] & _BMASK))


def first_occurrence_unique(keys):
    """Indices of the first occurrence of every distinct key, ascending.
    == ME.utils.sparse_quantize(..., return_index=True) order (SURVEY A.1,
    verified against files/3D_head_map.ply)."""
    _, first = np.unique(keys, return_index=True)   # stable -> first occurrence
    return np.sort(first)


def voxelize(xyz, voxel_size, batch_index=0):
    """util/misc.py:82-87.  coords = floor(xyz / voxel) in float64, unique rows in
    first-occurrence order.  Returns (coords int32 [M,4] = (b,x,y,z), inds int64 [M])."""
    xyz = np.asarray(xyz, dtype=np.float64)
    c = np.floor(xyz / voxel_size).astype(np.int64)
    c4 = np.concatenate([np.full((len(c), 1), batch_index, np.int64), c], axis=1)
    inds = first_occurrence_unique(pack_keys(c4))
    return c4[inds].astype(np.int32), inds.astype(np.int64)


def downsample(coords, out_stride):
    """Strided coordinate map (ME coordinate_map stride()):
    coarse = floor(c / out_stride) * out_stride, unique in first-occurrence order
    (that is the row order ME's CPU map produces; on its GPU map the order is
    unspecified and unobservable).  Returns (coarse int32 [Mc,4], parent int32 [M])."""
    c = np.asarray(coords, dtype=np.int64)
    q = c.copy()
    q[:, 1:] = (c[:, 1:] // out_stride) * out_stride          # numpy // floors
    keys = pack_keys(q)
    first = first_occurrence_unique(keys)
    coarse = q[first]
    order = np.argsort(keys[first], kind="stable")
    parent = order[np.searchsorted(keys[first][order], keys)]
    return coarse.astype(np.int32), parent.astype(np.int32)


def kernel_offsets(ksize):
    """Hyper-cube kernel region, odd size: k = (dx+r) + K1*(dy+r) + K1^2*(dz+r),
    x fastest (ME kernel_region.hpp iterator).  [K,3] int64."""
    assert ksize % 2 == 1
    r = ksize // 2
    rng = np.arange(-r, r + 1)
    dz, dy, dx = np.meshgrid(rng, rng, rng, indexing="ij")
    return np.stack([dx.ravel(), dy.ravel(), dz.ravel()], axis=1).astype(np.int64)


def _lookup(table_coords, query4):
    keys = pack_keys(table_coords)
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    q = pack_keys(query4)
    pos = np.searchsorted(sk, q)
    pos_c = np.minimum(pos, len(sk) - 1)
    hit = sk[pos_c] == q
    return np.where(hit, order[pos_c], -1).astype(np.int32)


def rulebook(in_coords, out_coords, ts_in, ksize):
    """Neighbour table nbr[o,k] = row of in_coords at out_coords[o] + off_k*ts_in,
    or -1 (covers stride-1 and stride-2 convs)."""
    offs = kernel_offsets(ksize) * ts_in
    out = np.asarray(out_coords, np.int64)
    nbr = np.empty((len(out), len(offs)), np.int32)
    lim = (1 << (_BITS - 1))
    for k, o in enumerate(offs):
        q = out.copy()
        q[:, 1:] += o
        ok = ((q[:, 1:] >= -lim) & (q[:, 1:] < lim)).all(axis=1)
        q[~ok, 1:] = 0
        r = _lookup(in_coords, q)
        r[~ok] = -1
        nbr[:, k] = r
    return nbr


def rulebook_transpose(coarse_coords, fine_coords, ts_fine, ksize):
    """Kernel map of MinkowskiConvolutionTranspose(k, stride 2) from tensor stride
    2*ts_fine to ts_fine: the forward (fine -> coarse) map with in/out
    swapped and the same k: out[f] += in[c] @ W[k] for f = c + off_k*ts_fine.
    Returns nbr_t[f,k] = coarse row c or -1."""
    offs = kernel_offsets(ksize) * ts_fine
    fine = np.asarray(fine_coords, np.int64)
    nbr = np.empty((len(fine), len(offs)), np.int32)
    for k, o in enumerate(offs):
        q = fine.copy()
        q[:, 1:] -= o
        nbr[:, k] = _lookup(coarse_coords, q)
    return nbr


def occupancy_sorted_slots(nbr, n_out, window=16384):
    """The slot order of csrc/rulebook_sort.hip (no reference counterpart: MinkowskiEngine's kernel maps have no tile
    structure; an ordering of the OUTPUT slots, the rows keep their numbers).  nbr [kvol, n_slots] in identity slot order.
    Stable sort inside windows of `window` slots by key = gray^-1(r): r = occupancy bits of the 12 edge offsets of the 3x3x3
    kernel (exactly two non-zero coordinates: offsets 1, 3, .., 25 without 13) in bits 17..6 and of the 6 face offsets in
    bits 5..0; slots >= n_out last in their window.  Returns perm (new slot s takes old slot perm[s]) and the boolean
    `valid` of the new slots."""
    nbr = np.asarray(nbr)
    kvol, n_slots = nbr.shape
    occ = (nbr >= 0).astype(np.uint64)
    r = np.zeros(n_slots, np.uint64)
    if kvol == 27:
        edges, faces = [1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25], [4, 10, 12, 14, 16, 22]
        for i, k in enumerate(edges):
            r |= occ[k] << np.uint64(17 - i)
        for i, k in enumerate(faces):
            r |= occ[k] << np.uint64(5 - i)
    else:
        for k in range(min(kvol, 18)):
            r |= occ[k] << np.uint64(k)
    key = r.copy()
    sh = 1
    while sh < 32:                                    # gray^-1: key = r ^ r >> 1 ^ r >> 2 ^ ...
        key ^= key >> np.uint64(sh)
        sh *= 2
    key &= np.uint64(0x3FFFF)
    slot = np.arange(n_slots)
    key = np.where(slot < n_out, key, np.uint64(1 << 18)) | ((slot // window).astype(np.uint64) << np.uint64(19))
    perm = np.argsort(key, kind="stable")
    return perm, perm < n_out


class Geometry:
    """All geometry products of one (possibly batched) fragment: 4 pyramid levels and the
    8 rulebooks ResUNetBN2C needs."""

    def __init__(self, coords, conv1_kernel_size=5):
        self.levels = [np.asarray(coords, np.int32)]
        self.parents = []
        for lv in range(3):
            c, p = downsample(self.levels[-1], 2 << lv)
            self.levels.append(c)
            self.parents.append(p)
        L = self.levels
        self.k_first = rulebook(L[0], L[0], 1, conv1_kernel_size)
        self.k3 = [rulebook(L[i], L[i], 1 << i, 3) for i in range(4)]
        self.down = [rulebook(L[i], L[i + 1], 1 << i, 3) for i in range(3)]
        self.up = [rulebook_transpose(L[i + 1], L[i], 1 << i, 3) for i in range(3)]


# --------------------------------------------------------------------------
# sparse-conv arithmetic  (ME ConvolutionForwardKernelCPU: per kernel offset
# gather -> GEMM -> scatter-add, fp32)
# --------------------------------------------------------------------------

SPCONV_IMPL = "torch"      # "c": route every convolution through the C / OpenMP twin (imf_oracle_cbind.spconv)


def spconv(feat, kernel, nbr):
    """out[o] = sum_k feat[nbr[o,k]] @ kernel[k]; kernel [K,Cin,Cout] or [Cin,Cout]."""
    if SPCONV_IMPL == "c":
        import imf_oracle_cbind as OC
        return OC.spconv(feat, kernel, nbr)
    feat = torch.as_tensor(feat, dtype=torch.float32)
    kernel = torch.as_tensor(kernel, dtype=torch.float32)
    if kernel.dim() == 2:
        return feat @ kernel
    nbr_t = torch.as_tensor(np.asarray(nbr), dtype=torch.int64)
    out = torch.zeros(nbr_t.shape[0], kernel.shape[2], dtype=torch.float32)
    for k in range(kernel.shape[0]):
        col = nbr_t[:, k]
        o = torch.nonzero(col >= 0).squeeze(1)
        if o.numel() == 0:
            continue
        out.index_add_(0, o, feat[col[o]] @ kernel[k])
    return out


def spconv_f64(feat, kernel, nbr):
    """float64 version -- the error yardstick for tolerance tests."""
    feat = torch.as_tensor(feat).double()
    kernel = torch.as_tensor(kernel).double()
    if kernel.dim() == 2:
        return feat @ kernel
    nbr_t = torch.as_tensor(np.asarray(nbr), dtype=torch.int64)
    out = torch.zeros(nbr_t.shape[0], kernel.shape[2], dtype=torch.float64)
    for k in range(kernel.shape[0]):
        col = nbr_t[:, k]
        o = torch.nonzero(col >= 0).squeeze(1)
        if o.numel():
            out.index_add_(0, o, feat[col[o]] @ kernel[k])
    return out


def batchnorm_eval(x, sd, prefix, eps=1e-5):
    """ME.MinkowskiBatchNorm in eval mode = BatchNorm1d on .F (model/common.py:6; A.5)."""
    return F.batch_norm(x, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"],
                        sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"], False, 0.0, eps)


def instance_norm(x, item, weight, bias, eps=1e-8):
    """ME.MinkowskiInstanceNorm (model/common.py:7-8, the `IN` blocks of ResUNetIN2*): per batch item and channel
    (x - mean) / sqrt(var + eps), mean and (biased) variance over the item's rows -- ME computes both with a global average
    pooling -- then weight * x + bias.  float64 inside.  [RECALLED: ME 0.5.4's eps = 1e-8 and the biased variance.]"""
    x64 = x.double().numpy()
    out = np.empty_like(x64)
    item = np.asarray(item)
    for b in np.unique(item):
        rows = item == b
        v = x64[rows]
        cen = v - v.mean(0, keepdims=True)
        out[rows] = cen / np.sqrt((cen * cen).mean(0, keepdims=True) + eps)
    out = out * weight.double().numpy().reshape(1, -1) + bias.double().numpy().reshape(1, -1)
    return torch.from_numpy(out).float()


def block_norm(x, sd, prefix, item):
    """The block's norm layer by what the state dict holds: BatchNorm ('<prefix>.bn.*') or InstanceNorm ('<prefix>.weight')."""
    if prefix + ".bn.weight" in sd:
        return batchnorm_eval(x, sd, prefix)
    return instance_norm(x, item, sd[prefix + ".weight"], sd[prefix + ".bias"])


def basic_block(x, sd, prefix, nbr, item=None):
    """model/residual_block.py:37-53 (BasicBlockBN / BasicBlockIN; `item`: the rows' batch indices, for IN)."""
    out = spconv(x, sd[prefix + ".conv1.kernel"], nbr)
    out = F.relu(block_norm(out, sd, prefix + ".norm1", item))
    out = spconv(out, sd[prefix + ".conv2.kernel"], nbr)
    out = block_norm(out, sd, prefix + ".norm2", item)
    return F.relu(out + x)


# --------------------------------------------------------------------------
# dense parts
# --------------------------------------------------------------------------

def image_encoder(image, sd, prefix="img_encoder.backbone."):
    """model/resnet.py:195-216 -- ResNet-34 truncated after layer2, BN in eval mode."""
    def bn(x, p):
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                            sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)

    def block(x, p, stride, has_down):
        idt = x
        out = F.relu(bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1), p + ".bn1"))
        out = bn(F.conv2d(out, sd[p + ".conv2.weight"], padding=1), p + ".bn2")
        if has_down:
            idt = bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), p + ".downsample.1")
        return F.relu(out + idt)

    x = torch.as_tensor(image, dtype=torch.float32)
    x = F.relu(bn(F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3), prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for i in range(3):
        x = block(x, f"{prefix}layer1.{i}", 1, False)
    for i in range(4):
        x = block(x, f"{prefix}layer2.{i}", 2 if i == 0 else 1, i == 0)
    return x


def attention_fusion(tokens, x, sd, prefix="attention_fusion.cross_attend_blocks."):
    """model/attention_fusion.py:132-154 with depth=0 (resunet.py:91-100).
    tokens [T,128] image context, x [N,256] point queries -> [N,256]."""
    p0, p1 = prefix + "0.", prefix + "1."
    xn = F.layer_norm(x, (x.shape[1],), sd[p0 + "norm.weight"], sd[p0 + "norm.bias"], 1e-5)
    cn = F.layer_norm(tokens, (tokens.shape[1],), sd[p0 + "norm_context.weight"],
                      sd[p0 + "norm_context.bias"], 1e-5)
    q = xn @ sd[p0 + "fn.to_q.weight"].t()
    kv = cn @ sd[p0 + "fn.to_kv.weight"].t()
    d = q.shape[1]
    k, v = kv[:, :d], kv[:, d:]
    sim = (q @ k.t()) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = (attn @ v) @ sd[p0 + "fn.to_out.weight"].t() + sd[p0 + "fn.to_out.bias"]
    x = out + x
    xn = F.layer_norm(x, (x.shape[1],), sd[p1 + "norm.weight"], sd[p1 + "norm.bias"], 1e-5)
    h = xn @ sd[p1 + "fn.net.0.weight"].t() + sd[p1 + "fn.net.0.bias"]
    a, g = h.chunk(2, dim=-1)
    h = a * F.gelu(g)
    return h @ sd[p1 + "fn.net.2.weight"].t() + sd[p1 + "fn.net.2.bias"] + x


# --------------------------------------------------------------------------
# full forward -- model/resunet.py:163-235
# --------------------------------------------------------------------------

def resunet_forward(sd, coords, image, feats=None, normalize_feature=True,
                    conv1_kernel_size=5, geometry=None, taps=None):
    """ResUNetBN2C.forward(x, image).  coords int32 [M,4] unique rows grouped by batch,
    image f32 [B,3,H,W], feats f32 [M,Cin] (default ones, util/misc.py:76-79).
    Returns F f32 [M,32].  `taps` (dict) collects intermediate tensors."""
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    g = geometry or Geometry(coords, conv1_kernel_size)
    M = len(g.levels[0])
    x = torch.ones(M, 1) if feats is None else torch.as_tensor(feats, dtype=torch.float32)
    tap = (lambda n, t: taps.__setitem__(n, t.clone())) if taps is not None else (lambda n, t: None)

    img = image_encoder(image, sd)                                        # :166
    tap("image_feat", img)

    out = batchnorm_eval(spconv(x, sd["conv1.kernel"], g.k_first), sd, "norm1")     # :168-169
    out_s1 = basic_block(out, sd, "block1", g.k3[0], g.levels[0][:, 0])
    out = F.relu(out_s1)
    tap("out_s1", out_s1)
    out = batchnorm_eval(spconv(out, sd["conv2.kernel"], g.down[0]), sd, "norm2")   # :173-174
    out_s2 = basic_block(out, sd, "block2", g.k3[1], g.levels[1][:, 0])
    out = F.relu(out_s2)
    tap("out_s2", out_s2)
    out = batchnorm_eval(spconv(out, sd["conv3.kernel"], g.down[1]), sd, "norm3")   # :178-179
    out_s4 = basic_block(out, sd, "block3", g.k3[2], g.levels[2][:, 0])
    out = F.relu(out_s4)
    tap("out_s4", out_s4)
    out = batchnorm_eval(spconv(out, sd["conv4.kernel"], g.down[2]), sd, "norm4")   # :183-184
    out_s8 = basic_block(out, sd, "block4", g.k3[3], g.levels[3][:, 0])
    out = F.relu(out_s8)
    tap("out_s8", out)

    # transformer(): :237-273 -- per batch item, rows are grouped by batch index
    b_idx = g.levels[3][:, 0]
    parts = []
    for b in range(int(b_idx.max()) + 1):
        rows = np.nonzero(b_idx == b)[0]
        assert (np.diff(rows) == 1).all()
        tok = img[b].reshape(img.shape[1], -1).t()                        # [H*W, C]  :257-261
        parts.append(attention_fusion(tok, out[rows[0]:rows[-1] + 1], sd))
    out = torch.cat(parts, 0)
    tap("fused", out)

    out = batchnorm_eval(spconv(out, sd["conv4_tr.kernel"], g.up[2]), sd, "norm4_tr")
    out = F.relu(basic_block(out, sd, "block4_tr", g.k3[2], g.levels[2][:, 0]))
    out = torch.cat([out, out_s4], 1)                                     # ME.cat :197
    out = batchnorm_eval(spconv(out, sd["conv3_tr.kernel"], g.up[1]), sd, "norm3_tr")
    out = F.relu(basic_block(out, sd, "block3_tr", g.k3[1], g.levels[1][:, 0]))
    out = torch.cat([out, out_s2], 1)                                     # :208
    out = batchnorm_eval(spconv(out, sd["conv2_tr.kernel"], g.up[0]), sd, "norm2_tr")
    out = F.relu(basic_block(out, sd, "block2_tr", g.k3[0], g.levels[0][:, 0]))
    tap("out_s1_tr", out)
    out = torch.cat([out, out_s1], 1)                                     # :219
    out = F.relu(spconv(out, sd["conv1_tr.kernel"], None))                # :224-225
    out = spconv(out, sd["final.kernel"], None) + sd["final.bias"]        # :226
    tap("final", out)
    if normalize_feature:
        out = out / torch.norm(out, p=2, dim=1, keepdim=True)             # :228-233 (no eps)
    return out


def extract_features(sd, xyz, voxel_size, image, conv1_kernel_size=5, normalize_feature=True):
    """util/misc.py:21-104 (rgb=None, normal=None path)."""
    coords, inds = voxelize(xyz, voxel_size)
    Fo = resunet_forward(sd, coords, image, None, normalize_feature, conv1_kernel_size)
    return np.asarray(xyz, np.float64)[inds], Fo


# --------------------------------------------------------------------------
# helpers shared by tests / bench: seeded weights with the reference's schema
# --------------------------------------------------------------------------

RESUNETBN2C = dict(CH=[None, 32, 64, 128, 256], TR=[None, 64, 64, 64, 128])


def resize_bilinear(img_hwc, H, W):
    """util/uio.py:31-40 cv2.resize(INTER_LINEAR) == half-pixel bilinear without
    antialias.  float32 HWC in, float32 HWC out."""
    t = torch.as_tensor(np.asarray(img_hwc, np.float32)).permute(2, 0, 1)[None]
    o = F.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)
    return o[0].permute(1, 2, 0).contiguous().numpy()


def fnv_hash_vec(arr):
    """ME.utils.fnv_hash_vec: FNV-1a-64 over the columns."""
    arr = np.asarray(arr).copy().astype(np.uint64, copy=False)
    h = np.full(arr.shape[0], np.uint64(14695981039346656037), dtype=np.uint64)
    for j in range(arr.shape[1]):
        h *= np.uint64(1099511628211)
        h = np.bitwise_xor(h, arr[:, j])
    return h


def seeded_state_dict(seed=0, conv1_kernel_size=5, in_channels=1, out_channels=32,
                      cfg=RESUNETBN2C, with_unused_image_layers=False):
    """Random weights in the reference's state_dict schema (SURVEY App. B, config 2 of
    §8d): conv kernels U(+-1/sqrt(K*Cin)), BN gamma U(.5,1.5), beta U(-.1,.1),
    mean N(0,.1), var U(.5,1.5).  Deterministic in `seed`."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def U(shape, a, b):
        return torch.rand(shape, generator=g) * (b - a) + a

    def conv(name, K, cin, cout):
        bound = 1.0 / math.sqrt(K * cin)
        sd[name + ".kernel"] = U((K, cin, cout) if K > 1 else (cin, cout), -bound, bound)

    def bn(name, c, mid=".bn"):
        sd[f"{name}{mid}.weight"] = U((c,), 0.5, 1.5)
        sd[f"{name}{mid}.bias"] = U((c,), -0.1, 0.1)
        sd[f"{name}{mid}.running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[f"{name}{mid}.running_var"] = U((c,), 0.5, 1.5)
        sd[f"{name}{mid}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def block(name, c):
        conv(name + ".conv1", 27, c, c); bn(name + ".norm1", c)
        conv(name + ".conv2", 27, c, c); bn(name + ".norm2", c)

    CH, TR = cfg["CH"], cfg["TR"]
    conv("conv1", conv1_kernel_size ** 3, in_channels, CH[1]); bn("norm1", CH[1]); block("block1", CH[1])
    conv("conv2", 27, CH[1], CH[2]); bn("norm2", CH[2]); block("block2", CH[2])
    conv("conv3", 27, CH[2], CH[3]); bn("norm3", CH[3]); block("block3", CH[3])
    conv("conv4", 27, CH[3], CH[4]); bn("norm4", CH[4]); block("block4", CH[4])
    p = "attention_fusion.cross_attend_blocks."
    ld, dim, inner = CH[4], 128, CH[4] // 2

    def lin(name, cout, cin, bias=True):
        b = 1.0 / math.sqrt(cin)
        sd[name + ".weight"] = U((cout, cin), -b, b)
        if bias:
            sd[name + ".bias"] = U((cout,), -b, b)

    def ln(name, c):
        sd[name + ".weight"] = U((c,), 0.5, 1.5)
        sd[name + ".bias"] = U((c,), -0.1, 0.1)

    lin(p + "0.fn.to_q", inner, ld, False); lin(p + "0.fn.to_kv", 2 * inner, dim, False)
    lin(p + "0.fn.to_out", ld, inner)
    ln(p + "0.norm", ld); ln(p + "0.norm_context", dim)
    lin(p + "1.fn.net.0", ld * 8, ld); lin(p + "1.fn.net.2", ld, ld * 4); ln(p + "1.norm", ld)
    conv("conv4_tr", 27, CH[4], TR[4]); bn("norm4_tr", TR[4]); block("block4_tr", TR[4])
    conv("conv3_tr", 27, CH[3] + TR[4], TR[3]); bn("norm3_tr", TR[3]); block("block3_tr", TR[3])
    conv("conv2_tr", 27, CH[2] + TR[3], TR[2]); bn("norm2_tr", TR[2]); block("block2_tr", TR[2])
    conv("conv1_tr", 1, CH[1] + TR[2], TR[1])
    conv("final", 1, TR[1], out_channels)
    sd["final.bias"] = U((1, out_channels), -0.1, 0.1)

    ip = "img_encoder.backbone."

    def conv2d(name, cout, cin, k):
        std = math.sqrt(2.0 / (cout * k * k))
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std

    def bn2d(name, c):
        bn(name, c, mid="")

    conv2d(ip + "conv1", 64, 3, 7); bn2d(ip + "bn1", 64)
    layers = [(1, 64, 64, 3), (2, 64, 128, 4)]
    if with_unused_image_layers:
        layers += [(3, 128, 256, 6), (4, 256, 512, 3)]
    for li, cin, cout, nb in layers:
        for i in range(nb):
            q = f"{ip}layer{li}.{i}"
            conv2d(q + ".conv1", cout, cin if i == 0 else cout, 3); bn2d(q + ".bn1", cout)
            conv2d(q + ".conv2", cout, cout, 3); bn2d(q + ".bn2", cout)
            if i == 0 and li > 1:
                conv2d(q + ".downsample.0", cout, cin, 1); bn2d(q + ".downsample.1", cout)
    if with_unused_image_layers:
        lin(ip + "fc", 1000, 512)
    return sd


# ---------------------------------------------------------------------------------------------
# Descriptor matching for feature-match recall -- test infrastructure like the rest
# of this file.  Parity status: the reference's KD-tree (Open3D 0.12 KDTreeFlann) is absent here; an
# exact 1-NN is uniquely defined up to exact ties, and this restatement is cross-checked against
# scipy's independent exact KD-tree in tests/test_oracle_golden.py.
# ---------------------------------------------------------------------------------------------
def knn_search(points_src, points_dst, k=1, chunk=256):
    """util/uio.py:245-258: exact nearest row of points_dst for every row of points_src, fp64
    `sum((a-b)^2)` distances; on exact ties the lowest index (first minimum)."""
    assert k == 1
    src = np.asarray(points_src, dtype=np.float64)
    dst = np.asarray(points_dst, dtype=np.float64)
    out = np.empty(len(src), dtype=np.int32)
    for s in range(0, len(src), chunk):
        d = ((src[s:s + chunk, None, :] - dst[None, :, :]) ** 2).sum(-1)
        out[s:s + chunk] = d.argmin(1)
    return out


def mutual_match_indices(frag21_nnindices, frag12_nnindices):
    """scripts/evaluation_3dmatch.py:212-217."""
    return np.flatnonzero(np.equal(np.arange(len(frag21_nnindices)), frag12_nnindices[frag21_nnindices]))


def transform_points(points, T):
    """Open3D PointCloud::transform (scripts/evaluation_3dmatch.py:223-225): homogeneous 4x4
    multiply, then divide by w."""
    p = np.asarray(points, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    h = np.concatenate([p, np.ones((len(p), 1))], 1) @ T.T
    return h[:, :3] / h[:, 3:4]


def feature_match(frag1_kpts, frag1_descs, frag2_kpts, frag2_descs, gt_pose, inlier_thresh=0.1):
    """scripts/evaluation_3dmatch.py:207-234 -> (num_inliers, inlier_ratio, frag2_match_indices,
    frag21_nnindices)."""
    nn21 = knn_search(frag2_descs, frag1_descs)
    nn12 = knn_search(frag1_descs, frag2_descs)
    m2 = mutual_match_indices(nn21, nn12)
    k2 = transform_points(np.asarray(frag2_kpts)[m2], gt_pose)
    k1 = np.asarray(frag1_kpts, dtype=np.float64)[nn21[m2]]
    distances = np.sqrt(np.sum(np.square(k1 - k2), axis=1))
    num_inliers = int(np.sum(distances < inlier_thresh))
    ratio = num_inliers / len(distances) if len(distances) else float("nan")
    return num_inliers, ratio, m2.astype(np.int32), nn21


def select_keypoints(sample_points, coords, voxel_size):
    """scripts/evaluation_3dmatch.py:162-171: rows of `coords` whose FNV key is among
    the keys of the sampled points."""
    key_points = fnv_hash_vec(np.floor(np.asarray(sample_points, dtype=np.float64) / voxel_size))
    key_coords = fnv_hash_vec(np.floor(np.asarray(coords, dtype=np.float64) / voxel_size))
    return np.where(np.isin(key_coords, key_points))[0]


# ---------------------------------------------------------------------------------------------
# RANSAC registration.  Parity status: UNPINNED against the reference -- Open3D 0.12 is
# absent here and seeds its generator from std::random_device, so not even the reference reproduces its
# own draws.  The published algorithm (Open3D 0.12 Registration.cpp, RegistrationRANSACBasedOnCorrespondence;
# CorrespondenceChecker.cpp; Eigen::umeyama) is restated over a counter-based generator shared with the
# HIP path.  The rigid fit is cross-checked against a closed-form construction in the tests.
# ---------------------------------------------------------------------------------------------
def _splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def rigid_fit(src, dst):
    """TransformationEstimationPointToPoint(False) = Eigen::umeyama without scale, batched:
    src, dst [..., n, 3] -> [..., 4, 4] with dst ~ R src + t."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    ms, md = src.mean(-2, keepdims=True), dst.mean(-2, keepdims=True)
    H = np.swapaxes(src - ms, -1, -2) @ (dst - md)
    U, _, Vt = np.linalg.svd(H)
    V = np.swapaxes(Vt, -1, -2)
    d = np.sign(np.linalg.det(V @ np.swapaxes(U, -1, -2)))
    D = np.zeros(H.shape)
    D[..., 0, 0] = D[..., 1, 1] = 1.0
    D[..., 2, 2] = d
    R = V @ D @ np.swapaxes(U, -1, -2)
    T = np.zeros(H.shape[:-2] + (4, 4))
    T[..., :3, :3] = R
    T[..., :3, 3] = md[..., 0, :] - (R @ ms[..., 0, :, None])[..., 0]
    T[..., 3, 3] = 1.0
    return T


def ransac_registration(src, dst, corres, ransac_n=3, max_corr_dist=0.075, edge_similarity=0.9,
                        max_iter=50000, seed=0):
    """scripts/benchmark_util.py:16-34 with the correspondences given (corres[i] = nearest target
    feature of source point i).  Returns (T 4x4 source->target, winning iteration or -1, inlier count,
    hypotheses that passed the checkers, fitness, inlier_rmse)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    corres = np.asarray(corres)
    n = len(src)
    it = np.arange(max_iter, dtype=np.uint64)
    picks = np.stack([(_splitmix64(np.uint64(seed) ^ (it * np.uint64(4) + np.uint64(j))) % np.uint64(n)).astype(np.int64)
                      for j in range(ransac_n)], 1)                       # [iter, ransac_n]
    S, D = src[picks], dst[corres[picks]]
    ok = np.ones(max_iter, bool)
    for i in range(ransac_n):                                             # CorrespondenceCheckerBasedOnEdgeLength
        for j in range(i + 1, ransac_n):
            ds = np.linalg.norm(S[:, i] - S[:, j], axis=1)
            dd = np.linalg.norm(D[:, i] - D[:, j], axis=1)
            ok &= ~((ds < dd * edge_similarity) | (dd < ds * edge_similarity))
    T = rigid_fit(S, D)
    moved = S @ np.swapaxes(T[:, :3, :3], 1, 2) + T[:, None, :3, 3]
    ok &= ~(np.linalg.norm(moved - D, axis=2) > max_corr_dist).any(1)     # CorrespondenceCheckerBasedOnDistance
    best = (-1, 0, 0.0)
    tgt = dst[corres]
    for h in np.flatnonzero(ok):                                          # ascending: ties keep the earliest
        dis = np.linalg.norm(src @ T[h, :3, :3].T + T[h, :3, 3] - tgt, axis=1)
        inl = dis < max_corr_dist
        c = int(inl.sum())
        if c == 0:
            continue
        rm = float(np.sqrt((dis[inl] ** 2).sum() / c))
        if c > best[1] or (c == best[1] and rm < best[2]):
            best = (int(h), c, rm)
    Tb = T[best[0]] if best[0] >= 0 else np.eye(4)
    return Tb, best[0], best[1], int(ok.sum()), best[1] / n, best[2]


def compute_transform_error(transform, covariance, estimated_transform):
    """util/uio.py:191-197 (nibabel's mat2quat restated: w-first quaternion of the relative rotation)."""
    rel = np.linalg.inv(transform) @ estimated_transform
    R, t = rel[:3, :3], rel[:3, 3]
    K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], 0, 0, 0],
                  [R[0, 1] + R[1, 0], R[1, 1] - R[0, 0] - R[2, 2], 0, 0],
                  [R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], R[2, 2] - R[0, 0] - R[1, 1], 0],
                  [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = -q
    er = np.concatenate([t, q[1:]], axis=0)
    return (er.reshape(1, 6) @ covariance @ er.reshape(6, 1) / covariance[0, 0]).item()


def compute_registration_error(gt_transform, est_transform):
    """util/uio.py:143-176: (RRE in degrees, RTE)."""
    x = 0.5 * (np.trace(est_transform[:3, :3].T @ gt_transform[:3, :3]) - 1.0)
    rre = 180.0 * np.arccos(np.clip(x, -1.0, 1.0)) / np.pi
    return rre, float(np.linalg.norm(gt_transform[:3, 3] - est_transform[:3, 3]))
