"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (imfnet_amd.ops ->
libimfnet_hip.so), against the CPU oracle on the same seeded inputs and against the committed
golden vectors.  Integer work (voxel indices, pyramid, rulebooks) must be bit-exact; fp32 features
within the tolerances written in each test (north_star: 1e-4 on descriptors)."""
import numpy as np
import pytest
import torch

import imf_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from imfnet_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def _voxelize_gpu(ops, xyz_t, vs):
    lv = ops.voxelize(xyz_t, vs)
    ops.sync_levels([lv])
    return lv


def _build_levels(ops, lv0):
    from imfnet_amd import sparse as ME
    cm = ME.CoordinateManager(lv0)
    cm.build_pyramid(8)
    return cm


# ------------------------------------------------------------------ voxelisation (bit-exact)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_voxelize_reference_head_map(ops, clouds, head_map, dtype):
    """The reference-pinned vector: files/3D_head_map.ply rows == xyz[inds] @ 2.5 cm."""
    xyz = torch.as_tensor(clouds[0]).to(DEV, dtype)
    lv = _voxelize_gpu(ops, xyz, 0.025)
    assert lv.n == 18977
    inds = lv.first_idx.cpu().numpy()
    assert (clouds[0][inds] == head_map).all()
    c_ref, i_ref = O.voxelize(clouds[0].astype(np.float64), 0.025)
    assert (lv.coords.cpu().numpy() == c_ref).all()
    assert (inds == i_ref).all()


@pytest.mark.parametrize("cloud,scale,vs,m", [(0, 1.0, 0.05, 5182), (1, 1.0, 0.05, 5140),
                                               (1, 1.0, 0.025, 19082), (0, 1.7, 0.025, 51232)])
def test_voxelize_counts_and_order(ops, clouds, cloud, scale, vs, m):
    xyz64 = clouds[cloud].astype(np.float64) * scale
    lv = _voxelize_gpu(ops, torch.as_tensor(xyz64).to(DEV), vs)
    c_ref, i_ref = O.voxelize(xyz64, vs)
    assert lv.n == m == len(c_ref)
    assert (lv.coords.cpu().numpy() == c_ref).all()
    assert (lv.first_idx.cpu().numpy() == i_ref).all()


def test_voxelize_edge_cases(ops):
    # negatives, exact voxel boundaries, duplicates, a single point, batch index
    pts = np.array([[-0.025, 0.0, 0.05], [-1e-12, 0.0249999, 0.05], [0.0, 0.0, 0.0], [-0.025, 0.0, 0.05],
                    [3.2, -3.2, 1e-9], [-0.0250001, 0.0, 0.05]], np.float64)
    lv = ops.voxelize(torch.as_tensor(pts).to(DEV), 0.025, batch_index=3)
    ops.sync_levels([lv])
    c_ref, i_ref = O.voxelize(pts, 0.025, batch_index=3)
    assert (lv.coords.cpu().numpy() == c_ref).all() and (lv.first_idx.cpu().numpy() == i_ref).all()
    one = ops.voxelize(torch.zeros(1, 3, dtype=torch.float64, device=DEV), 0.3)
    ops.sync_levels([one])
    assert one.n == 1 and one.coords.cpu().tolist() == [[0, 0, 0, 0]]
    from imfnet_amd import ImfError
    far = ops.voxelize(torch.full((4, 3), 1e6, dtype=torch.float64, device=DEV), 0.025)
    with pytest.raises(ImfError):
        ops.sync_levels([far])
    with pytest.raises(ImfError):
        ops.voxelize(torch.zeros(0, 3, dtype=torch.float64, device=DEV), 0.025)
    with pytest.raises(ImfError):
        ops.voxelize(torch.zeros(4, 3, dtype=torch.float64), 0.025)      # CPU tensor: no fallback


def test_voxelize_random_large(ops):
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(400_000, 3)) * 3.0
    lv = _voxelize_gpu(ops, torch.as_tensor(pts).to(DEV), 0.1)
    c_ref, i_ref = O.voxelize(pts, 0.1)
    assert lv.n == len(c_ref)
    assert (lv.coords.cpu().numpy() == c_ref).all() and (lv.first_idx.cpu().numpy() == i_ref).all()


# ------------------------------------------------------------------ pyramid + rulebooks (bit-exact)
@pytest.fixture(scope="module")
def geom_s5(ops, clouds):
    xyz64 = clouds[0].astype(np.float64)
    cm = _build_levels(ops, ops.voxelize(torch.as_tensor(xyz64).to(DEV), 0.05))
    g = O.Geometry(O.voxelize(xyz64, 0.05)[0])
    return cm, g


def test_pyramid_levels(ops, geom_s5):
    cm, g = geom_s5
    for i, ts in enumerate((1, 2, 4, 8)):
        assert (cm.coords(ts).cpu().numpy() == g.levels[i]).all(), f"level {ts}"


def _check_rulebook(rb, nbr_ref, identity_rows):
    n_out, kvol = nbr_ref.shape
    rows = rb.tile_rows.cpu().numpy()
    nbr = rb.nbr.cpu().numpy().reshape(kvol, rb.n_slots)
    valid = rows >= 0
    assert sorted(rows[valid].tolist()) == list(range(n_out))
    if identity_rows:
        assert (rows[:n_out] == np.arange(n_out)).all()
    assert (nbr[:, valid].T == nbr_ref[rows[valid]]).all()
    assert (nbr[:, ~valid] == -1).all()
    mask = rb.tile_mask.cpu().numpy().view(np.uint32).reshape(-1, 4)
    act = (nbr.reshape(kvol, -1, 64) >= 0).any(axis=2)                   # [kvol, tiles]
    for k in range(kvol):
        assert (((mask[:, k // 32] >> (k % 32)) & 1).astype(bool) == act[k]).all()
    return mask


def test_rulebooks_conv(ops, geom_s5):
    cm, g = geom_s5
    _check_rulebook(cm.conv_rulebook(1, 5, 1), g.k_first, True)
    for i in range(4):
        _check_rulebook(cm.conv_rulebook(1 << i, 3, 1), g.k3[i], True)
    for i in range(3):
        _check_rulebook(cm.conv_rulebook(1 << i, 3, 2), g.down[i], True)


def test_rulebooks_transpose(ops, geom_s5):
    cm, g = geom_s5
    for i in range(3):
        mask = _check_rulebook(cm.transpose_rulebook(2 << i, 3, 2), g.up[i], False)
        pop = np.array([bin(int(w)).count("1") for w in mask.reshape(-1)]).reshape(-1, 4).sum(1)
        assert pop.max() <= 8          # parity-class grouping: at most 8 offsets per tile


# ------------------------------------------------------------------ sparse convolution (fp32)
def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_pack_weights_layout(ops):
    for kvol, cin, cout in ((27, 32, 32), (3, 64, 128), (1, 96, 64), (2, 256, 64)):
        w = _rand((kvol, cin, cout), 1)
        CI = 64 if cin % 64 == 0 else 32
        J, CB = CI // 16, (4 if cout % 64 == 0 else 2)
        ref = (w.view(kvol, cin // CI, J, 4, 4, cout // (16 * CB), CB, 16)      # k cc j q t y cb c
                .permute(5, 0, 1, 2, 6, 3, 7, 4).reshape(-1))                  # y k cc j cb q c t
        got = ops.pack_weights(w.to(DEV)).cpu()
        assert torch.equal(got, ref)


CONV_CASES = [  # (cin_a, cin_b, cout, which rulebook)
    (32, 0, 32, ("k3", 0)), (32, 0, 64, ("down", 0)), (64, 0, 64, ("k3", 1)), (64, 0, 128, ("down", 1)),
    (128, 0, 128, ("k3", 2)), (128, 0, 256, ("down", 2)), (256, 0, 256, ("k3", 3)),
    (256, 0, 128, ("up", 2)), (128, 128, 64, ("up", 1)), (64, 64, 64, ("up", 0)),
    (64, 32, 64, ("k1", 0)), (64, 0, 32, ("k1", 0)),
]


def _rb_and_ref(cm, g, which):
    kind, i = which
    if kind == "k3":
        return cm.conv_rulebook(1 << i, 3, 1), g.k3[i], len(g.levels[i])
    if kind == "down":
        return cm.conv_rulebook(1 << i, 3, 2), g.down[i], len(g.levels[i])
    if kind == "up":
        return cm.transpose_rulebook(2 << i, 3, 2), g.up[i], len(g.levels[i + 1])
    return cm.conv_rulebook(1, 1, 1), None, len(g.levels[0])


@pytest.mark.parametrize("mode", ["auto", "split1", "split5", "split5_fused", "simple", "f32_regs", "f32_regs_split5",
                                  "f32_wave8", "f32_wave4", "b3", "b3_split1", "b3_split5", "b3_wave8", "b3_wave4", "b3_wave4h", "b3_wave8u", "b3_wave4u", "b3_wave4o", "b3_wave4h4", "b3_wave8h4",
                                  "h3", "h3_split1", "h3_split5", "h3_wave8", "h3_wave4"])
@pytest.mark.parametrize("ca,cb,cout,which", CONV_CASES)
def test_spconv_matches_oracle(ops, geom_s5, ca, cb, cout, which, mode):
    """Plain convolution (no epilogue) vs the oracle; error measured against an fp64 evaluation and
    required to be of fp32-roundoff size: <= 2e-6 * sum|a*b| bound (MFMA = ordered fmaf chain)."""
    cm, g = geom_s5
    rb, nbr_ref, n_in = _rb_and_ref(cm, g, which)
    kvol = 1 if nbr_ref is None else nbr_ref.shape[1]
    fa, fb = _rand((n_in, ca), 10), (_rand((n_in, cb), 11) if cb else None)
    w = _rand((kvol, ca + cb, cout), 12, 1.0 / np.sqrt(kvol * (ca + cb)))
    # variant 0 (fp32 MFMA, the reference's arithmetic) runs on the LDS-DMA kernels since round 5 (AR = kArF32): "auto" /
    # "split1" / "split5" = k_spconv_g, "f32_wave8" / "f32_wave4" = k_spconv_w; "f32_regs*" / "split5_fused" = round 1's
    # register-staged k_spconv_mfma (kernel_tag bit 1, or the in-launch combine it alone implements)
    kw = {"auto": {}, "split1": {"split_k": 1}, "split5": {"split_k": 5}, "simple": {"variant": 1},
          "split5_fused": {"split_k": 5, "fused_reduce": True},
          "f32_regs": {"staging": "regs"}, "f32_regs_split5": {"staging": "regs", "split_k": 5},
          "f32_wave8": {"staging": "wave8"}, "f32_wave4": {"staging": "wave4"},
          # variant 3 = bf16x3: fp32 operands as three bf16 parts each (exact), six bf16 MFMAs per 32 channels
          "b3": {"variant": 3}, "b3_split1": {"variant": 3, "split_k": 1}, "b3_split5": {"variant": 3, "split_k": 5},
          "b3_wave8": {"variant": 3, "staging": "wave8"}, "b3_wave4": {"variant": 3, "staging": "wave4"},
          "b3_wave4h": {"variant": 3, "staging": "wave4h"}, "b3_wave8u": {"variant": 3, "staging": "wave8u"},
          # round 6: 48-row units of 4 wavefronts; whole tiles under the three-wavefronts-per-SIMD register budget
          "b3_wave4u": {"variant": 3, "staging": "wave4u"}, "b3_wave4o": {"variant": 3, "staging": "wave4o"},
          "b3_wave4h4": {"variant": 3, "staging": "wave4h4"},          # half tiles built for four wavefronts per SIMD
          "b3_wave8h4": {"variant": 3, "staging": "wave8h4"},          # ... of eight wavefronts (a single fragment's coarse levels)
          "h3": {"variant": 6},
          "h3_split1": {"variant": 6, "split_k": 1}, "h3_split5": {"variant": 6, "split_k": 5},
          # variant 6 = the LDS-DMA kernel k_spconv_g (the register-staged k_spconv_h3 lives in diagnostic builds only)
          # the wave-split kernel of the coarse levels (csrc/spconv_w.hip): whole tile per workgroup, 8 / 4 wavefronts
          "h3_wave8": {"variant": 6, "staging": "wave8"}, "h3_wave4": {"variant": 6, "staging": "wave4"}}[mode]
    if mode.endswith(("wave8", "wave4", "wave4h", "wave8u", "wave4u", "wave4o", "wave4h4", "wave8h4")) and (kvol == 1 or cout % 64):
        pytest.skip("the wave-split kernel covers kvol > 1 and cout % 64 == 0")
    if mode in ("split5", "split5_fused", "h3_split5", "f32_regs_split5", "b3_split5") and kvol == 1:
        pytest.skip("pointwise convolution has a single offset")
    out = ops.spconv(fa.to(DEV), ops.pack_weights(w.to(DEV), variant=kw.get("variant", 0)), cout, rb,
                     in_b=None if fb is None else fb.to(DEV), **kw).cpu()
    fin = fa if fb is None else torch.cat([fa, fb], 1)
    wk = w if kvol > 1 else w[0]
    ref32 = O.spconv(fin, wk, nbr_ref)
    ref64 = O.spconv_f64(fin, wk, nbr_ref)
    bound = O.spconv_f64(fin.abs(), wk.abs(), nbr_ref) * 2e-6 + 1e-7
    assert out.shape == ref32.shape
    assert ((out.double() - ref64).abs() <= bound).all()
    assert (out - ref32).abs().max() < 5e-5


def test_spconv_chip_filling_launches_match_fp32_mfma(ops, clouds):
    """Launches of more than 512 workgroups take k_spconv_g's 2-deep buffer ring (four workgroups per CU), smaller ones
    the 4-deep ring -- the small oracle geometry above only ever sees the latter.  Here the chip-filling path (the fixture
    fragment x1.7 at 2.5 cm: ~51 k voxels, 800 tiles; the level-1 shapes through the wave-split kernel too) is held
    against the INDEPENDENT fp32-MFMA kernel (variant 0: different staging, different matrix instruction, an exact f32
    FMA chain) on the same inputs: <= 4e-6 * sum|a*b| per element (both within 2e-6 of the fp64 value, see
    test_spconv_matches_oracle), with the fused epilogue as well."""
    xyz = clouds[0].astype(np.float64) * 1.7
    cm = _build_levels(ops, ops.voxelize(torch.as_tensor(xyz).to(DEV), 0.025))
    n0 = cm.level(1).n
    assert n0 > 40_000
    for ca, cb, cout, rb, n_in, staging in ((32, 0, 32, cm.conv_rulebook(1, 3, 1), n0, None),
                                            (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, None),
                                            (64, 64, 64, cm.transpose_rulebook(2, 3, 2), cm.level(2).n, None),
                                            (64, 32, 64, cm.conv_rulebook(1, 1, 1), n0, None),
                                            (64, 0, 64, cm.conv_rulebook(2, 3, 1), cm.level(2).n, "wave4"),
                                            (32, 0, 64, cm.conv_rulebook(1, 3, 2), n0, "wave4")):
        if staging is None:
            assert rb.n_slots // 64 * max(1, cout // 64) > 512          # the chip-filling path
        fa, fb = _rand((n_in, ca), 60).to(DEV), (_rand((n_in, cb), 61).to(DEV) if cb else None)
        w = _rand((rb.kvol, ca + cb, cout), 62, 0.05).to(DEV)
        wp6, wp0 = ops.pack_weights(w, split16=True), ops.pack_weights(w)
        fin = (fa if fb is None else torch.cat([fa, fb], 1)).abs()
        bound = ops.spconv(fin[:, :ca].contiguous(), ops.pack_weights(w.abs()), cout, rb,
                           in_b=None if fb is None else fin[:, ca:].contiguous(), split_k=1, variant=0, staging="regs") * 4e-6 + 1e-6
        sc, sh = (_rand((cout,), 63).abs() + 0.5).to(DEV), _rand((cout,), 64).to(DEV)
        a = ops.spconv(fa, wp6, cout, rb, in_b=fb, variant=6, split_k=1, staging=staging)
        b = ops.spconv(fa, wp0, cout, rb, in_b=fb, variant=0, split_k=1, staging="regs")
        c = ops.spconv(fa, wp0, cout, rb, in_b=fb, variant=0, split_k=1, staging=staging)   # fp32 on the DMA kernels (round 5)
        assert ((a - b).abs() <= bound).all(), (ca, cb, cout, float((a - b).abs().max()))
        assert ((c - b).abs() <= bound).all(), (ca, cb, cout, float((c - b).abs().max()))
        a = ops.spconv(fa, wp6, cout, rb, in_b=fb, variant=6, split_k=1, scale=sc, shift=sh, relu=True, staging=staging)
        b = ops.spconv(fa, wp0, cout, rb, in_b=fb, variant=0, split_k=1, scale=sc, shift=sh, relu=True, staging="regs")
        c = ops.spconv(fa, wp0, cout, rb, in_b=fb, variant=0, split_k=1, scale=sc, shift=sh, relu=True, staging=staging)
        assert ((a - b).abs() <= bound * sc + 1e-6).all(), (ca, cb, cout, float((a - b).abs().max()))
        assert ((c - b).abs() <= bound * sc + 1e-6).all(), (ca, cb, cout, float((c - b).abs().max()))


def test_rulebook_sorted_by_occupancy(ops, clouds):
    """imf_rulebook_sort_by_occupancy against its numpy restatement (oracle.occupancy_sorted_slots: stable sort of the slots inside
    16 k-slot windows by the gray-inverse of the edge / face / corner occupancy bits, padding last): tile_rows, the gathered neighbour table and the recomputed tile masks integer-exact on a map of several
    sort windows; a convolution over the sorted map returns the rows of the plain one (same terms, other partition: fp32
    round-off), and walks fewer (tile, offset) pairs."""
    xyz = clouds[0].astype(np.float64) * 1.7
    cm = _build_levels(ops, ops.voxelize(torch.as_tensor(xyz).to(DEV), 0.025))
    for ts in (1, 2):
        rb = cm.conv_rulebook(ts, 3, 1)
        rs = ops.rulebook_sorted(rb)
        K, S, n = rb.kvol, rb.n_slots, rb.n_out
        nbr = rb.nbr.view(K, S).cpu().numpy()
        perm, valid = O.occupancy_sorted_slots(nbr, n)
        want_rows = np.where(valid, perm, -1).astype(np.int32)
        want_nbr = np.where(valid[None, :], nbr[:, perm], -1)
        want_mask = np.zeros((S // 64, 4), dtype=np.uint32)
        want_mask[:, 0] = ((want_nbr >= 0).reshape(K, S // 64, 64).any(2).astype(np.uint32) << np.arange(K, dtype=np.uint32)[:, None]).sum(0)
        assert np.array_equal(rs.tile_rows.cpu().numpy(), want_rows)
        assert np.array_equal(rs.nbr.view(K, S).cpu().numpy(), want_nbr)
        assert np.array_equal(rs.tile_mask.cpu().numpy().view(np.uint32).reshape(-1, 4), want_mask)
        active = lambda m: np.unpackbits(m.cpu().numpy().view(np.uint8)).sum() / (K * S // 64)
        if ts == 1:
            assert n > 3 * 16384 and active(rs.tile_mask) < 0.9 * active(rb.tile_mask)
        fa = _rand((n, 64), 40).to(DEV)
        w = _rand((K, 64, 64), 41, 0.05).to(DEV)
        for variant, staging in ((3, None), (3, "wave4h"), (0, "wave8"), (6, "wave4")):
            wp = ops.pack_weights(w, variant=variant)
            a = ops.spconv(fa, wp, 64, rb, variant=variant, split_k=1, staging=staging)
            b = ops.spconv(fa, wp, 64, rs, variant=variant, split_k=1, staging=staging)
            assert float((a - b).abs().max()) < (2e-5 if variant != 6 else 5e-5) * float(a.abs().max())


def test_pack_weights_bf16x3_is_an_exact_split(ops):
    """imf_pack_weights_bf16x3: every weight as three bf16 parts whose sum IS the fp32 value (three 8-bit significands carry
    fp32's 24 bits; bf16 has fp32's exponent range, so magnitudes from 1e-30 to 1e30 survive), in the split-f16 image's lane
    order with three parts per column block."""
    for kvol, cin, cout in ((27, 32, 32), (3, 64, 128), (1, 96, 64)):
        w = _rand((kvol, cin, cout), 5) * torch.pow(10.0, _rand((kvol, cin, cout), 6) * 8)      # ~16 decades
        w[0, 0, 0], w[0, 1, 1], w[0, 2, 2] = 1e-30, -3e30, 0.0
        CB = 4 if cout % 64 == 0 else 2
        img = ops.pack_weights(w.to(DEV), variant=3).cpu().view(torch.bfloat16)
        assert img.numel() == kvol * cin * cout * 3
        parts = img.view(cout // (16 * CB), kvol, cin // 32, CB, 3, 64, 8).float()               # y k cc cb part lane t
        total = parts.double().sum(4)                                                            # exact in fp64
        ref = (w.view(kvol, cin // 32, 2, 4, 4, cout // (16 * CB), CB, 16)                        # k cc t>>2 q t&3 y cb c
                .permute(5, 0, 1, 6, 3, 7, 2, 4).reshape(cout // (16 * CB), kvol, cin // 32, CB, 64, 8))   # lane = 16 q + c
        assert torch.equal(total, ref.double())
        p0 = parts[:, :, :, :, 0]
        assert torch.equal(p0, ref.to(torch.bfloat16).float())                                   # first part = bf16(w), nearest-even


def test_spconv_bf16x3_is_fp32_class(ops, clouds):
    """Variant 3 against the fp32-MFMA kernel on the chip-filling and coarse shapes, errors measured against fp64: the
    bf16x3 products drop three terms of <= 2^-26 |a||w| together and accumulate in fp32 like the fp32 MFMA does, so its
    error must be of the fp32 kernel's own size (<= 1.5 x its worst element + roundoff floor), with no range restriction:
    inputs of magnitude 1e6 and weights of 1e-6 (either overflows / underflows an f16 operand)."""
    xyz = clouds[0].astype(np.float64) * 1.7
    cm = _build_levels(ops, ops.voxelize(torch.as_tensor(xyz).to(DEV), 0.025))
    n0, n1 = cm.level(1).n, cm.level(2).n
    for ca, cb, cout, rb, n_in, staging, amp in ((64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, None, 1.0),
                                                 (32, 0, 32, cm.conv_rulebook(1, 3, 1), n0, None, 1e6),
                                                 (64, 64, 64, cm.transpose_rulebook(2, 3, 2), n1, None, 1.0),
                                                 (64, 32, 64, cm.conv_rulebook(1, 1, 1), n0, None, 1.0),
                                                 (64, 0, 128, cm.conv_rulebook(2, 3, 1), n1, "wave4", 1e6),
                                                 (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave8", 1.0),
                                                 (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave4h", 1.0),
                                                 (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, "wave4h", 1e6),
                                                 (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave8u", 1.0),
                                                 (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, "wave8u", 1e6),
                                                 (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave4u", 1.0),
                                                 (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, "wave4u", 1e6),
                                                 (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave4o", 1.0),
                                                 (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, "wave4o", 1e6),
                                                 (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave4h4", 1.0),
                                                 (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, "wave4h4", 1e6),
                                                 (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave8h4", 1.0),
                                                 (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, "wave8h4", 1e6)):
        fa, fb = _rand((n_in, ca), 90).to(DEV) * amp, (_rand((n_in, cb), 91).to(DEV) * amp if cb else None)
        w = _rand((rb.kvol, ca + cb, cout), 92, 0.05).to(DEV) / amp
        b3 = ops.spconv(fa, ops.pack_weights(w, variant=3), cout, rb, in_b=fb, variant=3, split_k=1, staging=staging)
        f32 = ops.spconv(fa, ops.pack_weights(w), cout, rb, in_b=fb, variant=0, split_k=1, staging=staging)
        fin = fa if fb is None else torch.cat([fa, fb], 1)
        nbr = rb.nbr.view(rb.kvol, -1)[:, :rb.n_slots] if rb.nbr is not None else None
        # fp64 reference on the device: gather-GEMM per offset through the rulebook's own neighbour table
        ref = torch.zeros((rb.n_out, cout), dtype=torch.float64, device=DEV)
        rows = rb.tile_rows[:rb.n_slots].long() if rb.tile_rows is not None else torch.arange(rb.n_out, device=DEV)
        for k in range(rb.kvol):
            idx = nbr[k].long() if nbr is not None else rows
            ok = (idx >= 0) & (rows >= 0)
            ref.index_add_(0, rows[ok], fin[idx[ok]].double() @ w[k].double())
        e3, e0 = (b3.double() - ref).abs().max().item(), (f32.double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert e3 <= 1.5 * e0 + 1e-7 * scale, (ca, cb, cout, staging, e3, e0, scale)
        assert e0 <= 2e-6 * scale * 8, (ca, cb, cout, staging, e0, scale)                        # (the reference itself is sane)
        if staging in ("wave8u", "wave4u"):
            # 48-row units: same arithmetic, other partition -- fp32-class like the rest (above), and the fused epilogue
            # (scale / shift / residual / ReLU) lands on the right rows: against the whole-tile launch within 4 fp32 ulps of the sums
            w3 = ops.pack_weights(w, variant=3)
            sc, sh = (_rand((cout,), 93).abs() + 0.5).to(DEV), _rand((cout,), 94).to(DEV)
            res = _rand((rb.n_out, cout), 95).to(DEV) * amp
            kw = dict(in_b=fb, variant=3, split_k=1, scale=sc, shift=sh, residual=res, relu=True)
            u, t = ops.spconv(fa, w3, cout, rb, staging=staging, **kw), ops.spconv(fa, w3, cout, rb, staging=staging[:5], **kw)
            assert (u - t).abs().max().item() <= 4e-6 * scale * float(sc.max()) + 1e-7 * amp
        if staging in ("wave4h", "wave4o", "wave4h4", "wave8h4"):
            # half-tile workgroups walk their tile's offset list with the same four wavefront ranges: the SAME sums as the
            # whole-tile launch, bit for bit -- plain and through the fused epilogue (scale / shift / residual / ReLU); so does
            # the whole-tile kernel built for three wavefronts per SIMD (wave4o: other register budget, same instructions' sums)
            w3 = ops.pack_weights(w, variant=3)
            whole = "wave8" if staging == "wave8h4" else "wave4"
            assert torch.equal(b3, ops.spconv(fa, w3, cout, rb, in_b=fb, variant=3, split_k=1, staging=whole))
            sc, sh = (_rand((cout,), 93).abs() + 0.5).to(DEV), _rand((cout,), 94).to(DEV)
            res = _rand((rb.n_out, cout), 95).to(DEV) * amp
            kw = dict(in_b=fb, variant=3, split_k=1, scale=sc, shift=sh, residual=res, relu=True)
            assert torch.equal(ops.spconv(fa, w3, cout, rb, staging=staging, **kw), ops.spconv(fa, w3, cout, rb, staging=whole, **kw))


def test_spconv_operand_images(ops, clouds):
    """imf_conv_args.operand_format: a convolution fed the split-f16 operand image of its input forms the SAME products as
    one fed the fp32 rows (bit-identical output) -- on k_spconv_g (chip-filling and small launches, one and two sources)
    and on the wave-split kernel; an output written as an operand image carries hi + lo = the fp32 output to 2^-22; a
    residual handed over as an image is read as hi + lo.  Host-side images: ops.to_operand_image (numpy-free torch)."""
    xyz = clouds[0].astype(np.float64) * 1.7
    cm = _build_levels(ops, ops.voxelize(torch.as_tensor(xyz).to(DEV), 0.025))
    n0, n1 = cm.level(1).n, cm.level(2).n
    A, R, OUT = ops.FMT_A_SPLIT, ops.FMT_RES_SPLIT, ops.FMT_OUT_SPLIT
    x = _rand((1000, 96), 70).to(DEV)
    assert torch.equal(ops.from_operand_image(ops.to_operand_image(x)),
                       (x.half().float() + (x - x.half().float()).half().float()))
    for ca, cb, cout, rb, n_in, staging in ((32, 0, 32, cm.conv_rulebook(1, 3, 1), n0, None),
                                            (64, 0, 64, cm.conv_rulebook(1, 3, 1), n0, None),
                                            (64, 64, 64, cm.transpose_rulebook(2, 3, 2), n1, None),
                                            (64, 32, 64, cm.conv_rulebook(1, 1, 1), n0, None),
                                            (64, 0, 128, cm.conv_rulebook(2, 3, 1), n1, None),
                                            (64, 0, 64, cm.conv_rulebook(2, 3, 1), n1, "wave4"),
                                            (64, 64, 128, cm.conv_rulebook(2, 3, 1), n1, "wave8"),
                                            (32, 0, 64, cm.conv_rulebook(1, 3, 2), n0, "wave4")):
        fa, fb = _rand((n_in, ca), 71).to(DEV), (_rand((n_in, cb), 72).to(DEV) if cb else None)
        w = _rand((rb.kvol, ca + cb, cout), 73, 0.05).to(DEV)
        wp = ops.pack_weights(w, split16=True)
        sc, sh = (_rand((cout,), 74).abs() + 0.5).to(DEV), _rand((cout,), 75).to(DEV)
        res = _rand((rb.n_out, cout), 76).to(DEV)
        ia, ib = ops.to_operand_image(fa), (None if fb is None else ops.to_operand_image(fb))
        kw = dict(variant=6, split_k=1, staging=staging, scale=sc, shift=sh, relu=True)
        ref = ops.spconv(fa, wp, cout, rb, in_b=fb, **kw)
        got = ops.spconv(ia, wp, cout, rb, in_b=ib, operand_format=A, **kw)
        assert torch.equal(ref, got), (ca, cb, cout, staging)
        # residual: fp32 vs its image (the image's value is what gets added)
        res_img = ops.to_operand_image(res)
        ref_r = ops.spconv(fa, wp, cout, rb, in_b=fb, residual=ops.from_operand_image(res_img), **kw)
        got_r = ops.spconv(ia, wp, cout, rb, in_b=ib, residual=res_img, operand_format=A | R, **kw)
        assert torch.equal(ref_r, got_r), (ca, cb, cout, staging)
        full = ops.spconv(fa, wp, cout, rb, in_b=fb, residual=res, **kw)
        assert ((ref_r - full).abs() <= 1e-6 * (1 + full.abs())).all()
        # output as an image: decodes to the fp32 output's hi + lo, and is exactly the image of the fp32 output
        out_img = ops.spconv(ia, wp, cout, rb, in_b=ib, residual=res_img, operand_format=A | R | OUT, **kw)
        assert torch.equal(out_img.view(torch.int32), ops.to_operand_image(got_r).view(torch.int32)), (ca, cb, cout, staging)
        dec = ops.from_operand_image(out_img)
        # (relative 2^-22; below ~0.06 the lo half is an f16 subnormal: absolute 2^-25)
        assert ((dec - got_r).abs() <= torch.clamp(got_r.abs() * 2.0 ** -22, min=2.0 ** -25)).all()
    # the fused head on operand images == on fp32 rows
    a, b = _rand((n0, 64), 77).to(DEV), _rand((n0, 32), 78).to(DEV)
    w1, w2 = _rand((1, 96, 64), 79, 0.1).to(DEV), _rand((1, 64, 32), 80, 0.1).to(DEV)
    w1p, w2p = ops.pack_weights(w1, split16=True), ops.pack_weights(w2, split16=True)
    s1, h1, h2 = (_rand((64,), 81).abs() + 0.5).to(DEV), _rand((64,), 82).to(DEV), _rand((32,), 83).to(DEV)
    f32 = ops.pointwise_head(a, b, w1p, w2p, scale1=s1, shift1=h1, shift2=h2)
    img = ops.pointwise_head(ops.to_operand_image(a), ops.to_operand_image(b), w1p, w2p, scale1=s1, shift1=h1, shift2=h2,
                             a_split=True)
    assert torch.equal(f32, img)


def test_spconv_epilogues(ops, geom_s5):
    cm, g = geom_s5
    rb, nbr_ref = cm.conv_rulebook(1, 3, 1), g.k3[0]
    n = len(g.levels[0])
    f, w = _rand((n, 32), 20), _rand((27, 32, 32), 21, 0.05)
    sc, sh, res = _rand((32,), 22).abs() + 0.5, _rand((32,), 23), _rand((n, 32), 24)
    wp = ops.pack_weights(w.to(DEV))
    base = O.spconv(f, w, nbr_ref)
    got = ops.spconv(f.to(DEV), wp, 32, rb, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV),
                     relu=True).cpu()
    assert (got - torch.relu(base * sc + sh + res)).abs().max() < 5e-5
    ref = base + sh
    ref = ref / ref.norm(dim=1, keepdim=True)
    for kw in ({}, {"split_k": 3}, {"variant": 1}):
        got = ops.spconv(f.to(DEV), wp, 32, rb, shift=sh.to(DEV), l2norm=True, **kw).cpu()
        assert (got - ref).abs().max() < 5e-6
        assert torch.allclose(got.norm(dim=1), torch.ones(n), atol=1e-5)
    for kw in ({"split_k": 4}, {"variant": 1}):
        got = ops.spconv(f.to(DEV), wp, 32, rb, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV),
                         relu=True, **kw).cpu()
        assert (got - torch.relu(base * sc + sh + res)).abs().max() < 5e-5


@pytest.mark.parametrize("staging", ["wave8", "wave4"])
def test_spconv_wave_kernel_epilogues(ops, geom_s5, staging):
    """k_spconv_w (one workgroup per tile, sub-stages split over its wavefronts, partial tiles combined in LDS): the
    in-launch epilogue (scale / shift / residual / ReLU, L2 norm), the two-source input and the transposed map against
    the oracle; run-to-run bit reproducibility; and agreement with the unsplit k_spconv_g within fp32 roundoff."""
    cm, g = geom_s5
    for which, ca, cb, cout in ((("k3", 1), 64, 0, 64), (("up", 1), 128, 128, 64), (("down", 1), 64, 0, 128)):
        rb, nbr_ref, n_in = _rb_and_ref(cm, g, which)
        fa, fb = _rand((n_in, ca), 70), (_rand((n_in, cb), 71) if cb else None)
        w = _rand((27, ca + cb, cout), 72, 1.0 / np.sqrt(27 * (ca + cb)))
        sc, sh, res = _rand((cout,), 73).abs() + 0.5, _rand((cout,), 74), _rand((rb.n_out, cout), 75)
        wp = ops.pack_weights(w.to(DEV), split16=True)
        fin = fa if fb is None else torch.cat([fa, fb], 1)
        base = O.spconv_f64(fin, w, nbr_ref)
        kw = dict(in_b=None if fb is None else fb.to(DEV), variant=6)
        got = ops.spconv(fa.to(DEV), wp, cout, rb, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True,
                         staging=staging, **kw)
        assert (got.cpu().double() - torch.relu(base * sc + sh + res)).abs().max() < 2e-5
        again = ops.spconv(fa.to(DEV), wp, cout, rb, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True,
                           staging=staging, **kw)
        assert torch.equal(got, again)
        unsplit = ops.spconv(fa.to(DEV), wp, cout, rb, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True,
                             split_k=1, **kw)
        assert (got - unsplit).abs().max() < 2e-5
        if cout == 64:
            ref = base + sh
            ref = ref / ref.norm(dim=1, keepdim=True)
            got = ops.spconv(fa.to(DEV), wp, cout, rb, shift=sh.to(DEV), l2norm=True, staging=staging, **kw).cpu()
            assert (got.double() - ref).abs().max() < 2e-6
        flags = torch.zeros(1, dtype=torch.int32, device=DEV)          # range guard: raised by the in-launch epilogue
        ops.spconv(fa.to(DEV) * 3e5, wp, cout, rb, staging=staging, flags=flags, **kw)
        assert int(flags.item()) & 32
    with pytest.raises(Exception):                                     # 32 output channels: not served
        rb = cm.conv_rulebook(1, 3, 1)
        ops.spconv(_rand((len(g.levels[0]), 32), 76).to(DEV), ops.pack_weights(_rand((27, 32, 32), 77).to(DEV), split16=True),
                   32, rb, variant=6, staging=staging)


@pytest.mark.parametrize("ca,cb,n", [(64, 32, 5182), (64, 32, 64 * 700 + 17), (64, 0, 1000), (64, 64, 333), (96, 0, 63)])
def test_pointwise_head(ops, ca, cb, n):
    """imf_pointwise_head (conv1_tr + norm1_tr + ReLU + final + bias + L2 norm, model/resunet.py:219-233, one launch)
    against an fp64 evaluation of the same formulas (<= 2e-6 on unit rows), bit-identical to the two variant-6
    convolution launches it replaces (same MFMA order, same epilogue expressions, the intermediate rounded to fp32 at
    the same place), with the device-side row count of the capacity mode, and the range flag of the hidden block."""
    from imfnet_amd.ops import Rulebook
    fa, fb = _rand((n, ca), 80), (_rand((n, cb), 81) if cb else None)
    w1, w2 = _rand((1, ca + cb, 64), 82, 1.0 / np.sqrt(ca + cb)), _rand((1, 64, 32), 83, 0.125)
    sc, sh, bias = _rand((64,), 84).abs() + 0.5, _rand((64,), 85), _rand((32,), 86)
    w1p, w2p = ops.pack_weights(w1.to(DEV), split16=True), ops.pack_weights(w2.to(DEV), split16=True)
    a, b = fa.to(DEV), (None if fb is None else fb.to(DEV))
    got = ops.pointwise_head(a, b, w1p, w2p, scale1=sc.to(DEV), shift1=sh.to(DEV), relu1=True, shift2=bias.to(DEV),
                             l2norm=True)
    fin = (fa if fb is None else torch.cat([fa, fb], 1)).double()
    hid = torch.relu(fin @ w1[0].double() * sc.double() + sh.double())
    ref = hid @ w2[0].double() + bias.double()
    ref = ref / ref.norm(dim=1, keepdim=True)
    assert got.shape == (n, 32)
    assert (got.cpu().double() - ref).abs().max() < 2e-6
    # the two launches it replaces
    slots = (n + 63) // 64 * 64
    rb = Rulebook(None, None, None, slots, n, 1)
    h = ops.spconv(a, w1p, 64, rb, in_b=b, scale=sc.to(DEV), shift=sh.to(DEV), relu=True, variant=6)
    two = ops.spconv(h, w2p, 32, rb, shift=bias.to(DEV), l2norm=True, variant=6)
    assert torch.equal(got, two)
    # capacity mode: rows beyond the device-side count are neither read into a result nor written
    m = max(1, n - 37)
    n_dev = torch.tensor([m], dtype=torch.int32, device=DEV)
    out = torch.full((n, 32), 7.0, device=DEV)
    ops.pointwise_head(a, b, w1p, w2p, scale1=sc.to(DEV), shift1=sh.to(DEV), relu1=True, shift2=bias.to(DEV), l2norm=True,
                       out=out, n_dev=n_dev)
    assert torch.equal(out[:m], got[:m]) and bool((out[m:] == 7.0).all())
    # range guard on the hidden block; without ReLU / L2 norm
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.pointwise_head(a * 1e3, b, w1p, w2p, scale1=torch.full((64,), 1e3, device=DEV), flags=flags)   # hidden ~ 1e6
    assert int(flags.item()) & 32
    flags.zero_()
    plain = ops.pointwise_head(a, b, w1p, w2p, relu1=False, l2norm=False, flags=flags).cpu().double()
    assert int(flags.item()) == 0
    assert (plain - (fin @ w1[0].double()) @ w2[0].double()).abs().max() < 2e-5
    with pytest.raises(Exception):
        ops.pointwise_head(a[:, :32].contiguous(), None, w1p, w2p)


@pytest.mark.parametrize("ca,cb,n", [(64, 32, 5182), (64, 32, 128 * 300 + 17), (64, 0, 1000), (32, 32, 333), (96, 0, 63)])
def test_pointwise_head_bf16x3(ops, ca, cb, n):
    """imf_pointwise_head with variant 3 (bf16x3 images, fp32 rows): against fp64 (fp32-class error on unit rows), bit-identical
    to the two variant-3 convolution launches it replaces, the device-side row count of the capacity mode, no range flag for
    hidden values far outside the f16 range, and 128 input channels refused (the image would not fit beside the row buffers)."""
    from imfnet_amd.ops import Rulebook
    fa, fb = _rand((n, ca), 80), (_rand((n, cb), 81) if cb else None)
    w1, w2 = _rand((1, ca + cb, 64), 82, 1.0 / np.sqrt(ca + cb)), _rand((1, 64, 32), 83, 0.125)
    sc, sh, bias = _rand((64,), 84).abs() + 0.5, _rand((64,), 85), _rand((32,), 86)
    w1p, w2p = ops.pack_weights(w1.to(DEV), variant=3), ops.pack_weights(w2.to(DEV), variant=3)
    a, b = fa.to(DEV), (None if fb is None else fb.to(DEV))
    kw = dict(scale1=sc.to(DEV), shift1=sh.to(DEV), relu1=True, shift2=bias.to(DEV), l2norm=True, variant=3)
    got = ops.pointwise_head(a, b, w1p, w2p, **kw)
    fin = (fa if fb is None else torch.cat([fa, fb], 1)).double()
    hid = torch.relu(fin @ w1[0].double() * sc.double() + sh.double())
    ref = hid @ w2[0].double() + bias.double()
    ref = ref / ref.norm(dim=1, keepdim=True)
    assert got.shape == (n, 32)
    assert (got.cpu().double() - ref).abs().max() < 1e-6
    rb = Rulebook(None, None, None, (n + 63) // 64 * 64, n, 1)
    h = ops.spconv(a, w1p, 64, rb, in_b=b, scale=sc.to(DEV), shift=sh.to(DEV), relu=True, variant=3)
    two = ops.spconv(h, w2p, 32, rb, shift=bias.to(DEV), l2norm=True, variant=3)
    assert torch.equal(got, two)
    m = max(1, n - 37)
    n_dev = torch.tensor([m], dtype=torch.int32, device=DEV)
    out = torch.full((n, 32), 7.0, device=DEV)
    ops.pointwise_head(a, b, w1p, w2p, out=out, n_dev=n_dev, **kw)
    assert torch.equal(out[:m], got[:m]) and bool((out[m:] == 7.0).all())
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    big = ops.pointwise_head(a * 1e3, b, w1p, w2p, scale1=torch.full((64,), 1e3, device=DEV), relu1=False, l2norm=False,
                             flags=flags, variant=3).cpu().double()                            # hidden ~ 1e6: fine here
    assert int(flags.item()) == 0
    fin_big = (fa * 1e3 if fb is None else torch.cat([fa * 1e3, fb], 1)).double()
    want = (fin_big @ w1[0].double() * 1e3) @ w2[0].double()
    assert (big - want).abs().max() <= 2e-6 * want.abs().max()
    with pytest.raises(Exception):
        ops.pointwise_head(torch.zeros((64, 128), device=DEV), None, ops.pack_weights(_rand((1, 128, 64), 87).to(DEV), variant=3),
                           w2p, variant=3)


def test_spconv_split16_variant(ops, geom_s5):
    """Variant 6 (split-f16 MFMA): the packed image decodes to hi + lo == w within 2^-21, epilogues and
    determinism as the fp32 kernels, and fp32-class error on inputs spanning seven decades."""
    cm, g = geom_s5
    rb, nbr_ref = cm.conv_rulebook(1, 3, 1), g.k3[0]
    n = len(g.levels[0])
    w = _rand((27, 64, 32), 40, 0.05)
    for wscale in (1.0, 1e-3, 1e-6, 3e4):                       # trained kernels are ~1e-2 .. 1e-3: lo would be subnormal unscaled
        ws = w * wscale
        full = ops.pack_weights(ws.to(DEV), split16=True).cpu()
        assert full.numel() == 27 * 64 * 32 + 64                # image + trailer (max |w| bits, 2^-shift)
        unscale = float(full[27 * 64 * 32 + 1])
        assert unscale > 0 and np.log2(unscale) == round(np.log2(unscale))          # a power of two
        assert 2.0 ** 13 <= float(ws.abs().max()) / unscale < 2.0 ** 14
        img = full[:27 * 64 * 32].view(torch.float16)
        # [y][k][cc][q = 2 cb + h][lane][t]: ci = 32 cc + 16 (t >> 2) + 4 (lane >> 4) + (t & 3), co = 16 cb + (lane & 15)
        v = img.view(1, 27, 2, 2, 2, 4, 16, 2, 4).double()          # y k cc cb h g c th tl
        rec = (v[:, :, :, :, 0] + v[:, :, :, :, 1])[0]              # k cc cb g c th tl
        rec = rec.permute(0, 1, 5, 3, 6, 2, 4).reshape(27, 64, 32) * unscale   # k (cc th g tl) (cb c)
        # hi + lo carries >= 21 bits of every weight down to 2^-17 of the largest one (lo stays a normal f16)
        assert ((rec - ws.double()).abs() <= ws.abs().double() * 2.0 ** -21 + float(ws.abs().max()) * 2.0 ** -39).all()
    f = _rand((n, 64), 41)
    sc, sh, res = _rand((32,), 42).abs() + 0.5, _rand((32,), 43), _rand((n, 32), 44)
    wp = ops.pack_weights(w.to(DEV), split16=True)
    base = O.spconv_f64(f, w, nbr_ref)
    for kw in ({}, {"split_k": 1}, {"split_k": 3}):
        got = ops.spconv(f.to(DEV), wp, 32, rb, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV),
                         relu=True, variant=6, **kw)
        assert (got.cpu().double() - torch.relu(base * sc + sh + res)).abs().max() < 2e-5
        assert torch.equal(got, ops.spconv(f.to(DEV), wp, 32, rb, scale=sc.to(DEV), shift=sh.to(DEV),
                                           residual=res.to(DEV), relu=True, variant=6, **kw))
        ref = (base + sh)
        ref = ref / ref.norm(dim=1, keepdim=True)
        got = ops.spconv(f.to(DEV), wp, 32, rb, shift=sh.to(DEV), l2norm=True, variant=6, **kw).cpu()
        assert (got.double() - ref).abs().max() < 2e-6
    # wide dynamic range: |x| from 1e-4 to 1e3 (f16 subnormal lo parts on the small ones)
    mag = torch.pow(10.0, torch.empty(n, 64).uniform_(-4, 3, generator=torch.Generator().manual_seed(45)))
    fw = f.sign() * mag
    got = ops.spconv(fw.to(DEV), wp, 32, rb, variant=6).cpu().double()
    ref64 = O.spconv_f64(fw, w, nbr_ref)
    bound = O.spconv_f64(fw.abs(), w.abs(), nbr_ref) * 2e-6 + 1e-7
    assert ((got - ref64).abs() <= bound).all()
    with pytest.raises(Exception):                             # k5 (125 offsets) is not served by variant 6
        ops.spconv(f.to(DEV), wp, 32, cm.conv_rulebook(1, 5, 1), variant=6)


def test_spconv_deterministic(ops, geom_s5):
    cm, g = geom_s5
    rb = cm.conv_rulebook(1, 3, 1)
    f = _rand((len(g.levels[0]), 64), 30).to(DEV)
    wp = ops.pack_weights(_rand((27, 64, 64), 31, 0.03).to(DEV))
    for kw in ({}, {"split_k": 4}, {"variant": 1}):
        a = ops.spconv(f, wp, 64, rb, **kw)
        b = ops.spconv(f, wp, 64, rb, **kw)
        assert torch.equal(a, b)
    # in-kernel combine == two-pass combine, bit for bit, and stable over many launches
    ref = ops.spconv(f, wp, 64, rb, split_k=6, fused_reduce=False)
    for _ in range(20):
        assert torch.equal(ops.spconv(f, wp, 64, rb, split_k=6, fused_reduce=True), ref)


@pytest.mark.parametrize("cin,cout,ks", [(1, 32, 5), (1, 32, 3), (3, 32, 3), (4, 64, 3)])
def test_spconv_small_cin(ops, geom_s5, cin, cout, ks):
    cm, g = geom_s5
    rb = cm.conv_rulebook(1, ks, 1)
    nbr_ref = g.k_first if ks == 5 else g.k3[0]
    n = len(g.levels[0])
    f = torch.ones(n, cin) if cin == 1 else _rand((n, cin), 40)
    w = _rand((ks ** 3, cin, cout), 41, 0.1)
    sc, sh = _rand((cout,), 42).abs() + 0.5, _rand((cout,), 43)
    got = ops.spconv_small_cin(f.to(DEV), w.to(DEV), rb, sc.to(DEV), sh.to(DEV), relu=True).cpu()
    ref = torch.relu(O.spconv(f, w, nbr_ref) * sc + sh)
    assert (got - ref).abs().max() < 2e-5


@pytest.mark.parametrize("cin,cout,ks", [(1, 32, 5), (1, 32, 3), (3, 32, 5), (2, 64, 3), (1, 64, 5)])
def test_conv_first_fused(ops, geom_s5, cin, cout, ks):
    """Hash-probing first layer == table-driven small-Cin conv == oracle."""
    cm, g = geom_s5
    nbr_ref = g.k_first if ks == 5 else g.k3[0]
    n = len(g.levels[0])
    f = torch.ones(n, cin) if cin == 1 else _rand((n, cin), 50)
    w = _rand((ks ** 3, cin, cout), 51, 0.1)
    sc, sh = _rand((cout,), 52).abs() + 0.5, _rand((cout,), 53)
    ref = torch.relu(O.spconv(f, w, nbr_ref) * sc + sh)
    got = ops.conv_first_fused(cm.level(1), f.to(DEV), w.to(DEV), ks, sc.to(DEV), sh.to(DEV), relu=True).cpu()
    assert (got - ref).abs().max() < 2e-5
    if cin == 1:
        got1 = ops.conv_first_fused(cm.level(1), None, w.to(DEV), ks, sc.to(DEV), sh.to(DEV), relu=True).cpu()
        assert torch.equal(got, got1)


@pytest.mark.parametrize("cout,ks,scale,vs", [(32, 5, 1.0, 0.05), (32, 3, 1.0, 0.05), (64, 5, 1.0, 0.05),
                                               (32, 5, 1.7, 0.025)])
def test_conv_first_bitgrid(ops, clouds, cout, ks, scale, vs):
    """Occupancy bit-grid first layer (all-ones input) == oracle sparse conv over the k-offset table;
    the pyramid build reports the level-0 bounding box it needs."""
    xyz = clouds[0].astype(np.float64) * scale
    levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(DEV), vs, 4)
    c_ref, _ = O.voxelize(xyz, vs)
    assert levels[0].bbox[:4] == [0] + c_ref[:, 1:].min(0).tolist()
    assert levels[0].bbox[4:] == [0] + c_ref[:, 1:].max(0).tolist()
    nbr_ref = O.rulebook(c_ref, c_ref, 1, ks)
    w = _rand((ks ** 3, 1, cout), 61, 0.1)
    sc, sh = _rand((cout,), 62).abs() + 0.5, _rand((cout,), 63)
    got = ops.conv_first_bitgrid(levels[0], w.to(DEV), ks, sc.to(DEV), sh.to(DEV), relu=True)
    assert got is not None
    ref = torch.relu(O.spconv(torch.ones(len(c_ref), 1), w, nbr_ref) * sc + sh)
    assert (got.cpu() - ref).abs().max() < 2e-5
    fused = ops.conv_first_fused(levels[0], None, w.to(DEV), ks, sc.to(DEV), sh.to(DEV), relu=True)
    assert (got - fused).abs().max() < 2e-5


def test_spconv_argument_errors(ops, geom_s5):
    from imfnet_amd import ImfError
    cm, g = geom_s5
    rb = cm.conv_rulebook(1, 3, 1)
    f = torch.zeros(len(g.levels[0]), 48, device=DEV)
    with pytest.raises(ImfError):
        ops.spconv(f, torch.zeros(27 * 48 * 32, device=DEV), 32, rb)        # cin % 32 != 0
    with pytest.raises(ImfError):
        ops.spconv(f[:, :32].contiguous(), torch.zeros(5, device=DEV), 32, rb)   # wrong weight size
    wp = ops.pack_weights(torch.zeros(27, 32, 32, device=DEV))
    for retired in (2, 3, 4, 5):                                             # round-1 experiments, no longer built
        with pytest.raises(ImfError):
            ops.spconv(f[:, :32].contiguous(), wp, 32, rb, variant=retired)
    wp6 = ops.pack_weights(torch.zeros(27, 32, 32, device=DEV), split16=True)
    for kw in ({"staging": "regs"}, {"split_k": 4, "fused_reduce": True}):      # the register-staged twin: diagnostic builds only
        with pytest.raises(ImfError, match="diagnostic"):
            ops.spconv(f[:, :32].contiguous(), wp6, 32, rb, variant=6, **kw)


# ------------------------------------------------------------------ whole model
@pytest.fixture(scope="module")
def model(seeded_sd):
    from imfnet_amd.model import load_model
    Model = load_model("ResUNetBN2C")
    m = Model(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
    missing = m.load_state_dict(seeded_sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval().to(DEV)


def test_fused_fusion_kernel_matches_reference_module(ops, model, golden):
    """imf_fusion_attention (one HIP kernel) vs the output of the reference's own AttentionFusion
    module on the golden inputs (413 point rows -- not a multiple of 16 -- x 300 image tokens)."""
    fw = model._fusion_weights()
    assert fw.supported
    x = torch.as_tensor(golden["af_in"]).to(DEV)
    ctx = torch.as_tensor(golden["af_ctx"]).to(DEV)
    blk = model.attention_fusion.cross_attend_blocks[0]
    with torch.no_grad():
        kv = blk.fn.to_kv(blk.norm_context(ctx))                       # [300, 256]
        kt = torch.zeros(128, 320, device=DEV); kt[:, :300] = kv[:, :128].t()
        vp = torch.zeros(320, 128, device=DEV); vp[:300] = kv[:, 128:]
        out = ops.fusion_attention(x, ops.pack_weights(kt), ops.pack_weights(vp), 300, 320, fw)
        ref_torch = model._fusion_fast(x, kv)
    assert np.abs(out.cpu().numpy() - golden["af_out"]).max() < 5e-5
    assert (out - ref_torch).abs().max() < 5e-5
    out2 = ops.fusion_attention(x, ops.pack_weights(kt), ops.pack_weights(vp), 300, 320, fw)
    assert torch.equal(out, out2)                                       # deterministic
    # the fp32-MFMA feed-forward (variant 0: what the f16-range recompute runs) on the fp32 images of the same matrices
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    out0 = ops.fusion_attention(x, ops.pack_weights(kt), ops.pack_weights(vp), 300, 320, fw, flags=flags, variant=0)
    assert np.abs(out0.cpu().numpy() - golden["af_out"]).max() < 5e-5 and int(flags.item()) == 0
    assert (out0 - out).abs().max() < 5e-6


def test_fusion_range_flag_in_both_variants(ops, model, golden, fast_mode):
    """The feed-forward's f16 range guard.  Rows scaled until the GEGLU hidden passes 65504: variant 6 must raise
    IMF_FLAG_RANGE (its operands would be inf); variant 0 -- what the flagged fragment is redone with, fp32 throughout -- computes
    finite fp32 values (round 5: its convolution launches no longer raise the bit, a large value is a value in fp32)."""
    fw = model._fusion_weights()
    x = torch.as_tensor(golden["af_in"]).to(DEV)
    ctx = torch.as_tensor(golden["af_ctx"]).to(DEV)
    blk = model.attention_fusion.cross_attend_blocks[0]
    ff = model.attention_fusion.cross_attend_blocks[1].fn.net
    with torch.no_grad():
        kv = blk.fn.to_kv(blk.norm_context(ctx))
        kt = torch.zeros(128, 320, device=DEV); kt[:, :300] = kv[:, :128].t()
        vp = torch.zeros(320, 128, device=DEV); vp[:300] = kv[:, 128:]
        w0, b0 = ff[0].weight.clone(), ff[0].bias.clone()
        try:
            ff[0].weight.mul_(3000.0); ff[0].bias.mul_(3000.0)            # hidden ~ 1e7 * gelu
            from imfnet_amd.ops import FusionKernelWeights
            big = FusionKernelWeights(model.attention_fusion)
            for variant in (6, 0):
                flags = torch.zeros(1, dtype=torch.int32, device=DEV)
                out = ops.fusion_attention(x, ops.pack_weights(kt), ops.pack_weights(vp), 300, 320, big, flags=flags,
                                           variant=variant)
                if variant == 6:
                    assert int(flags.item()) & 32, "variant 6: no range flag"
                else:
                    assert torch.isfinite(out).all()
                    out_f32 = out
            # the default arithmetic (bf16x3) on the same rows: finite, no flag from its convolution launches, = fp32 to roundoff
            O_ = ops
            O_.CONV_VARIANT = 3
            big3 = FusionKernelWeights(model.attention_fusion)
            out3 = ops.fusion_attention(x, ops.pack_weights(kt), ops.pack_weights(vp), 300, 320, big3, variant=3)
            O_.CONV_VARIANT = 6
            assert torch.isfinite(out3).all() and ((out3 - out_f32).abs() <= 1e-5 * out_f32.abs().max()).all()
        finally:
            ff[0].weight.copy_(w0); ff[0].bias.copy_(b0)


def test_forward_matches_reference_golden_S5(model, clouds, images, golden):
    """Config 1: cloud_bin_0 @ 5 cm.  Golden = the reference's own model code (CPU stand-in ops)."""
    from imfnet_amd.extract import extract_features
    with torch.no_grad():
        xyz_down, F = extract_features(model, clouds[0].astype(np.float64), voxel_size=0.05,
                                       device=torch.device(DEV), skip_check=True, image=images[0])
    assert F.is_cuda and F.shape == (5182, 32)
    assert (xyz_down.astype(np.float32) == golden["S5_xyz_down_f32"]).all()
    assert np.abs(F.cpu().numpy() - golden["S5_F"]).max() < 1e-4            # north_star tolerance


@pytest.mark.parametrize("variant", [6, 0])
def test_forward_matches_reference_golden_other_arithmetics(seeded_sd, clouds, images, golden, variant):
    """The goldens of the reference's own model code against the two arithmetics that are not the default: the split-f16
    fast mode (variant 6) and fp32 MFMA (variant 0) -- config 1 (S5) and a full 2.5 cm fragment, exact path (first call) and
    capacity mode (second call), each within the north_star's 1e-4 and within 2e-6 of the default arithmetic (bf16x3)."""
    from imfnet_amd import ops as O_
    from imfnet_amd.extract import extract_features
    from imfnet_amd.model import load_model

    def run(v):
        prev, O_.CONV_VARIANT = O_.CONV_VARIANT, v
        try:
            m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
            m.load_state_dict(seeded_sd, strict=True)
            m = m.eval().to(DEV)
            outs = []
            with torch.no_grad():
                for _ in range(2):
                    outs.append(extract_features(m, clouds[0].astype(np.float64), voxel_size=0.05, device=torch.device(DEV),
                                                 skip_check=True, image=images[0])[1].cpu().numpy())
                F25 = extract_features(m, clouds[0].astype(np.float64), voxel_size=0.025, device=torch.device(DEV),
                                       skip_check=True, image=images[0])[1].cpu().numpy()
            assert m.fragment_runner().variant == v and m.fragment_runner().stats["eager"] >= 1
            return outs, F25
        finally:
            O_.CONV_VARIANT = prev

    (a, b), F25 = run(variant)
    (d, _), D25 = run(3)
    assert np.abs(a - golden["S5_F"]).max() < 1e-4 and np.abs(b - golden["S5_F"]).max() < 1e-4
    assert np.abs(a - d).max() < 2e-6 and np.abs(F25 - D25).max() < 2e-6
    M = int(golden["S25_0_M"])
    assert F25.shape == (M, 32) and np.abs(F25[:: max(1, M // 256)][:256] - golden["S25_0_rows"]).max() < 1e-4


def test_forward_matches_reference_golden_crop(model, clouds, images, golden):
    from imfnet_amd.extract import extract_features
    crop = clouds[0].astype(np.float64)[golden["crop_sel_idx"]]
    with torch.no_grad():
        _, F = extract_features(model, crop, voxel_size=0.025, device=torch.device(DEV),
                                skip_check=True, image=images[0])
    assert np.abs(F.cpu().numpy() - golden["crop_F"]).max() < 1e-4


@pytest.mark.parametrize("i", [0, 1])
def test_forward_full_fragment_S25(model, clouds, images, golden, i):
    """Config 2: the pair @ 2.5 cm; golden keeps column sums and 256 sampled rows."""
    from imfnet_amd.extract import extract_features
    with torch.no_grad():
        xyz_down, F = extract_features(model, clouds[i].astype(np.float64), voxel_size=0.025,
                                       device=torch.device(DEV), skip_check=True, image=images[i])
    F = F.cpu().numpy()
    M = int(golden[f"S25_{i}_M"])
    assert F.shape == (M, 32)
    rows = F[:: max(1, M // 256)][:256]
    assert np.abs(rows - golden[f"S25_{i}_rows"]).max() < 1e-4
    assert np.abs(F.astype(np.float64).sum(0) - golden[f"S25_{i}_colsum"]).max() < 1e-4 * M ** 0.5 * 10


def test_fused_equals_layerwise(model, clouds, images):
    """The fused plan and the op-by-op walk (forward_layers) agree."""
    from imfnet_amd.extract import sparse_tensor_from_points
    xyz = clouds[1].astype(np.float64)
    img = torch.as_tensor(images[1]).to(DEV)
    with torch.no_grad():
        st, _ = sparse_tensor_from_points(xyz, 0.05, torch.device(DEV))
        a = model(st, img).F
        st2, _ = sparse_tensor_from_points(xyz, 0.05, torch.device(DEV))
        b = model.forward_layers(st2, img).F
    assert (a - b).abs().max() < 2e-5


@pytest.fixture
def model6(model, fast_mode):
    """The module's model with its plans rebuilt for the split-f16 fast mode (variant 6), back on the default afterwards."""
    model._invalidate()
    yield model
    model._invalidate()


def test_native_executor_equals_python_plan_bf16x3(model, clouds, images, monkeypatch):
    """The default arithmetic (variant 3): imf_resunet_forward issues exactly the launches of the op-by-op Python executor on
    fp32 buffers -- bit-identical descriptors; 20 convolution launches traced + the fused head (conv1_tr + final, which the
    op-by-op executor runs as two launches with the same sums)."""
    from imfnet_amd import ops as O_
    from imfnet_amd.extract import sparse_tensor_from_points
    assert O_.CONV_VARIANT == 3
    for k, voxel in ((0, 0.05), (1, 0.025)):
        xyz = clouds[k].astype(np.float64)
        img = torch.as_tensor(images[k]).to(DEV)
        with torch.no_grad():
            monkeypatch.setenv("IMFNET_PYTHON_EXECUTOR", "1")
            st, _ = sparse_tensor_from_points(xyz, voxel, torch.device(DEV))
            a = model(st, img).F.clone()
            monkeypatch.delenv("IMFNET_PYTHON_EXECUTOR")
            O_.TRACE = []
            st2, _ = sparse_tensor_from_points(xyz, voxel, torch.device(DEV))
            b = model(st2, img).F.clone()
            torch.cuda.synchronize()
            trace, O_.TRACE = O_.TRACE, None
        assert model._native_plan is not None
        assert torch.equal(a, b)
        assert len(trace) == 21 and all(r["kernel"].endswith("/b3") for r in trace[:-1])
        assert trace[-1]["kernel"] == "k_pointwise_head_b3"          # conv1_tr + final: one launch (csrc/head.hip)


def test_native_executor_equals_python_plan(model6, clouds, images, monkeypatch):
    """Fast mode (variant 6).  imf_resunet_forward (one C call per fragment) issues the launches of the Python arena executor:
    with fp32 feature buffers (what the op-by-op executor has) descriptors must be bit-identical, for both fragments
    and two voxel sizes, also when traced; in its default mode -- layers hand split-f16 operand images on, residual
    reads see 22 of 24 bits -- within 2e-6."""
    model = model6
    from imfnet_amd import ops as O_
    from imfnet_amd.extract import sparse_tensor_from_points
    for k, voxel in ((0, 0.05), (1, 0.05), (0, 0.025)):
        xyz = clouds[k].astype(np.float64)
        img = torch.as_tensor(images[k]).to(DEV)
        with torch.no_grad():
            monkeypatch.setenv("IMFNET_PYTHON_EXECUTOR", "1")
            st, _ = sparse_tensor_from_points(xyz, voxel, torch.device(DEV))
            a = model(st, img).F.clone()
            monkeypatch.delenv("IMFNET_PYTHON_EXECUTOR")
            st4, _ = sparse_tensor_from_points(xyz, voxel, torch.device(DEV))
            d = model(st4, img).F.clone()                      # default: operand images
            monkeypatch.setenv("IMFNET_FP32_BUFFERS", "1")
            st2, _ = sparse_tensor_from_points(xyz, voxel, torch.device(DEV))
            b = model(st2, img).F.clone()
            O_.TRACE = []
            st3, _ = sparse_tensor_from_points(xyz, voxel, torch.device(DEV))
            c = model(st3, img).F.clone()
            torch.cuda.synchronize()
            trace, O_.TRACE = O_.TRACE, None
            monkeypatch.delenv("IMFNET_FP32_BUFFERS")
        assert model._native_plan is not None
        assert torch.equal(a, b) and torch.equal(a, c)
        assert (a - d).abs().max() < 2e-6 and not torch.equal(a, d)
        assert len(trace) == 21 and all(r["ev"].elapsed_ms() > 0 for r in trace)
        assert sorted(r["name"] for r in trace)[0] == "block1.conv1"


def test_forward_from_coordinates_api(model, clouds, images, golden):
    """The reference's call shape: ME.SparseTensor(feats, coordinates=coords, device) -> model(...)."""
    import imfnet_amd.sparse as ME
    xyz = clouds[0].astype(np.float64)
    coords, inds = ME.utils.sparse_quantize(np.floor(xyz / 0.05), return_index=True)
    assert (xyz[inds].astype(np.float32) == golden["S5_xyz_down_f32"]).all()
    coords = ME.utils.batched_coordinates([coords])
    st = ME.SparseTensor(torch.ones(len(coords), 1), coordinates=coords, device=DEV)
    with torch.no_grad():
        F = model(st, torch.as_tensor(images[0]).to(DEV)).F
    assert np.abs(F.cpu().numpy() - golden["S5_F"]).max() < 1e-4
    assert (st.C.cpu().numpy() == coords.numpy()).all()


def test_batched_forward_matches_single(model, clouds, images):
    """Two fragments in one batch (rows grouped by batch index, one image each, resunet.py:241-250)."""
    import imfnet_amd.sparse as ME
    from imfnet_amd.extract import extract_features
    cs, single = [], []
    with torch.no_grad():
        for i in (0, 1):
            xyz = clouds[i].astype(np.float64)[::3]
            c, _ = O.voxelize(xyz, 0.05)
            cs.append(c[:, 1:])
            _, F = extract_features(model, xyz, voxel_size=0.05, device=torch.device(DEV), skip_check=True,
                                    image=images[i])
            single.append(F)
        coords = ME.utils.batched_coordinates(cs)
        st = ME.SparseTensor(torch.ones(len(coords), 1), coordinates=coords, device=DEV)
        img = torch.as_tensor(np.concatenate([images[0], images[1]])).to(DEV)
        Fb = model(st, img).F
    assert (Fb - torch.cat(single)).abs().max() < 2e-5


def _in_state_dict(seeded_sd, seed=5):
    """The seeded BN state dict with the seven residual blocks' norms replaced by InstanceNorm parameters ([1, C] weight /
    bias, no running statistics): the schema of ResUNetIN2C (model/resunet.py:316-318)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in seeded_sd.items():
        if k.startswith("block") and ".bn." in k:
            if k.endswith(".bn.weight"):
                sd[k.replace(".bn.weight", ".weight")] = (torch.rand(1, v.numel(), generator=g) + 0.5)
            elif k.endswith(".bn.bias"):
                sd[k.replace(".bn.bias", ".bias")] = (torch.rand(1, v.numel(), generator=g) - 0.5) * 0.2
            continue
        sd[k] = v
    return sd


def test_instance_norm_variant_matches_oracle(seeded_sd, clouds, images):
    """ResUNetIN2C (BatchNorm after the strided convolutions, InstanceNorm inside the residual blocks: model/resunet.py:316-318,
    model/common.py:7-8) on a batch of two fragments -- the statistics are per batch item -- against the oracle."""
    import imfnet_amd.sparse as ME
    from imfnet_amd.model import load_model
    sd = _in_state_dict(seeded_sd)
    m = load_model("ResUNetIN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
    m.load_state_dict(sd, strict=True)
    m = m.eval().to(DEV)
    cs = [O.voxelize(clouds[i].astype(np.float64)[::4], 0.05)[0] for i in (0, 1)]
    coords = np.concatenate([np.concatenate([np.full((len(c), 1), i, np.int32), c[:, 1:]], 1) for i, c in enumerate(cs)])
    img = np.concatenate([images[0], images[1]])
    with torch.no_grad():
        st = ME.SparseTensor(torch.ones(len(coords), 1), coordinates=torch.as_tensor(coords), device=DEV)
        Fb = m(st, torch.as_tensor(img).to(DEV)).F.cpu()
    Fr = O.resunet_forward(sd, coords, img)
    assert Fb.shape == Fr.shape and float((Fb - Fr).abs().max()) < 1e-4
    # the statistics really are per item: the first fragment alone gives the same rows
    with torch.no_grad():
        st0 = ME.SparseTensor(torch.ones(len(cs[0]), 1), coordinates=torch.as_tensor(coords[:len(cs[0])]), device=DEV)
        F0 = m(st0, torch.as_tensor(images[0]).to(DEV)).F.cpu()
    assert float((F0 - Fb[:len(cs[0])]).abs().max()) < 2e-5


def test_determinism_end_to_end(model, clouds, images):
    from imfnet_amd.extract import extract_features
    with torch.no_grad():
        a = extract_features(model, clouds[0], voxel_size=0.05, device=torch.device(DEV), skip_check=True,
                             image=images[0])[1]
        b = extract_features(model, clouds[0], voxel_size=0.05, device=torch.device(DEV), skip_check=True,
                             image=images[0])[1]
    assert torch.equal(a, b)


def test_native_batched_pair_matches_single_fragments(model, clouds, images, monkeypatch):
    """A PAIR of fragments in one forward through the native batched path (imf_pyramid_build_batched, one
    image each, imf_fusion_attention_batched): voxel rows are the concatenation of the single-fragment
    results, descriptors agree with the single-fragment forwards to rounding (split-K partitions differ
    with the tile count, so not bit for bit)."""
    from imfnet_amd.extract import extract_features, extract_features_batch
    pts = [clouds[0].astype(np.float64), clouds[1].astype(np.float64)]
    imgs = np.concatenate([images[0], images[1]], 0)
    for voxel in (0.05, 0.025):
        with torch.no_grad():
            single = [extract_features(model, xyz=pts[k], voxel_size=voxel, device=DEV, skip_check=True,
                                       image=torch.as_tensor(images[k])) for k in (0, 1)]
            single = [(a, b.clone()) for a, b in single]
            batched = extract_features_batch(model, pts, voxel, DEV, imgs)
            batched = [(a, b.clone()) for a, b in batched]
            monkeypatch.setenv("IMFNET_FP32_BUFFERS", "1")                     # the op-by-op executor's arithmetic
            batched_f32 = extract_features_batch(model, pts, voxel, DEV, imgs)
            batched_f32 = [(a, b.clone()) for a, b in batched_f32]
            monkeypatch.delenv("IMFNET_FP32_BUFFERS")
            monkeypatch.setenv("IMFNET_PYTHON_EXECUTOR", "1")
            batched_py = extract_features_batch(model, pts, voxel, DEV, imgs)
            monkeypatch.delenv("IMFNET_PYTHON_EXECUTOR")
        assert len(batched) == 2
        for k in (0, 1):
            assert (batched[k][0] == single[k][0]).all()                       # same voxels, same order
            assert batched[k][1].shape == single[k][1].shape
            assert (batched[k][1] - single[k][1]).abs().max() < 2e-6
            assert (batched_py[k][1] - batched_f32[k][1]).abs().max() == 0.0   # both executors, same launches
            assert (batched_py[k][1] - batched[k][1]).abs().max() < 2e-6       # operand images: 22-bit residual reads
    # three items, different sizes, float32 points
    three = [clouds[0][::3], clouds[1][::5], clouds[0][1::7]]
    imgs3 = np.concatenate([images[0], images[1], images[0]], 0)
    with torch.no_grad():
        out3 = extract_features_batch(model, three, 0.05, DEV, imgs3)
        for k in range(3):
            xd, F = extract_features(model, xyz=three[k].astype(np.float64), voxel_size=0.05, device=DEV,
                                     skip_check=True, image=torch.as_tensor(imgs3[k:k + 1]))
            assert (out3[k][0] == xd).all() and (out3[k][1] - F).abs().max() < 2e-6

