"""The HIP library against its `imf_cpu_*` twins (oracle/imf_cpu_twins.c: the same C-ABI signatures on host pointers, SURVEY
8b B3).  Both sides write the library's own layouts, so the integer outputs are compared array against array -- voxel
coordinates and first-occurrence indices, the three coarse levels, tile_rows / offset-major neighbour tables / per-tile masks
of the strided, stride-1 and parity-grouped transposed maps -- and a fused convolution (two sources, BatchNorm, residual,
ReLU, L2 norm; fp32 MFMA and the default bf16x3) within fp32 roundoff of the twin's FMA chain."""
import numpy as np
import pytest
import torch

import imf_cpu_twins as T
from test_cpu_twins import pack_weights_f32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def both(clouds):
    from imfnet_amd import ops
    from imfnet_amd import sparse as ME
    xyz = clouds[1].astype(np.float64) * 1.3
    cm = ME.CoordinateManager(ops.voxelize(torch.as_tensor(xyz).to(DEV), 0.025))
    cm.build_pyramid(8)
    lv0, err = T.voxelize(xyz, 0.025)
    levels = [lv0]
    for i in range(3):
        levels.append(T.downsample(levels[-1], 2 << i))
    assert err == 0
    return ops, cm, levels


def _same_map(rb, twin):
    rows, nbr, mask = twin
    assert rb.n_slots == len(rows)
    assert (rb.tile_rows.cpu().numpy() == rows).all()
    assert (rb.nbr.cpu().numpy().reshape(rb.kvol, rb.n_slots) == nbr).all()
    assert (rb.tile_mask.cpu().numpy().view(np.uint32).reshape(-1, 4) == mask).all()


def test_geometry_equals_the_cpu_twins(both):
    ops, cm, levels = both
    lv = cm.level(1)
    assert lv.n == levels[0].n > 25_000
    assert (lv.coords.cpu().numpy() == levels[0].coords).all() and (lv.first_idx.cpu().numpy() == levels[0].first_idx).all()
    for i, ts in enumerate((2, 4, 8)):
        assert (cm.coords(ts).cpu().numpy() == levels[i + 1].coords).all(), ts
    for i in range(4):
        _same_map(cm.conv_rulebook(1 << i, 3, 1), T.rulebook_conv(levels[i], levels[i], 1 << i, 3))
    _same_map(cm.conv_rulebook(1, 5, 1), T.rulebook_conv(levels[0], levels[0], 1, 5))
    for i in range(3):
        _same_map(cm.conv_rulebook(1 << i, 3, 2), T.rulebook_conv(levels[i], levels[i + 1], 1 << i, 3))
        _same_map(cm.transpose_rulebook(2 << i, 3, 2), T.rulebook_transpose(levels[i + 1], levels[i], 1 << i))


def test_sorted_map_equals_the_cpu_twin(both):
    """imf_rulebook_sort_by_occupancy (three launches: keys, one LDS radix sort per 16 k-slot window, gather) against its twin (a
    stable merge sort): the same permutation -- stride-1 maps of all four levels (several windows, two, and less than one:
    only a window's valid prefix is sorted) and the three strided maps."""
    ops, cm, levels = both
    for i in range(4):
        rb = cm.conv_rulebook(1 << i, 3, 1)
        _same_map(ops.rulebook_sorted(rb), T.rulebook_sort_by_occupancy(rb.nbr.cpu().numpy().reshape(27, rb.n_slots), rb.n_out))
    for i in range(3):
        rb = cm.conv_rulebook(1 << i, 3, 2)
        _same_map(ops.rulebook_sorted(rb), T.rulebook_sort_by_occupancy(rb.nbr.cpu().numpy().reshape(27, rb.n_slots), rb.n_out))


def test_sorted_map_of_tiny_and_window_edge_sizes(both):
    """Row counts around the sort's internal sizes: fewer rows than one 64-lane group, exactly one wavefront's share, one row
    short of / exactly / one row beyond a 16 384-slot window: against the numpy restatement, run twice (same bits)."""
    import imf_oracle as O
    ops, cm, levels = both
    rb0 = cm.conv_rulebook(1, 3, 1)
    K, S0 = 27, rb0.n_slots
    full = rb0.nbr.view(K, S0)
    for n in (1, 50, 64, 1024, 16383, 16384, 16385, 20000):
        S = (n + 63) // 64 * 64
        nbr = torch.where(full[:, :S] < n, full[:, :S], torch.full_like(full[:, :S], -1)).contiguous()   # a map of the first n rows
        nbr[:, n:] = -1
        mask = torch.zeros(S // 64 * 4, dtype=torch.int32, device=nbr.device)
        rb = ops.Rulebook(None, nbr.view(-1), mask, S, n, K, K)
        a, b = ops.rulebook_sorted(rb), ops.rulebook_sorted(rb)
        perm, valid = O.occupancy_sorted_slots(nbr.cpu().numpy(), n)
        want_rows = np.where(valid, perm, -1).astype(np.int32)
        assert np.array_equal(a.tile_rows.cpu().numpy(), want_rows), n
        assert np.array_equal(a.nbr.view(K, S).cpu().numpy(), np.where(valid[None, :], nbr.cpu().numpy()[:, perm], -1)), n
        assert torch.equal(a.tile_rows, b.tile_rows) and torch.equal(a.nbr, b.nbr) and torch.equal(a.tile_mask, b.tile_mask)


@pytest.mark.parametrize("variant", [0, 3])
def test_fused_convolution_equals_the_cpu_twin(both, variant):
    from imfnet_amd._lib import ConvArgs
    ops, cm, levels = both
    rng = np.random.default_rng(1)
    for ca, cb, cout, rb, twin, n_in, staging in (
            (32, 0, 32, cm.conv_rulebook(1, 3, 1), T.rulebook_conv(levels[0], levels[0], 1, 3), levels[0].n, None),
            (64, 64, 64, cm.transpose_rulebook(2, 3, 2), T.rulebook_transpose(levels[1], levels[0], 1), levels[1].n, None),
            (64, 0, 128, cm.conv_rulebook(2, 3, 2), T.rulebook_conv(levels[1], levels[2], 2, 3), levels[1].n, "wave8")):
        rows, nbr, mask = twin
        n_out = rb.n_out
        fa = rng.normal(size=(n_in, ca)).astype(np.float32)
        fb = rng.normal(size=(n_in, cb)).astype(np.float32) if cb else None
        w = (rng.normal(size=(27, ca + cb, cout)) / np.sqrt(27 * (ca + cb))).astype(np.float32)
        sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
        res = rng.normal(size=(n_out, cout)).astype(np.float32)
        l2 = cout == 32
        wp, ref = pack_weights_f32(w), np.zeros((n_out, cout), np.float32)
        a = ConvArgs()
        a.in_a, a.c_a, a.in_b, a.c_b = fa.ctypes.data, ca, (fb.ctypes.data if cb else None), cb
        a.w_packed, a.kvol, a.cout = wp.ctypes.data, 27, cout
        a.tile_rows, a.nbr, a.tile_mask = rows.ctypes.data, nbr.ctypes.data, mask.ctypes.data
        a.n_slots, a.n_out = len(rows), n_out
        a.scale, a.shift, a.residual, a.relu, a.l2norm = sc.ctypes.data, sh.ctypes.data, res.ctypes.data, 1, int(l2)
        a.out, a.variant, a.split_k = ref.ctypes.data, 0, 1
        T.spconv_fwd(a)
        t = lambda x: None if x is None else torch.as_tensor(x).to(DEV)   # noqa: E731
        assert torch.equal(ops.pack_weights(t(w)).cpu(), torch.as_tensor(wp))                 # the twin reads the library's image
        got = ops.spconv(t(fa), ops.pack_weights(t(w), variant=variant), cout, rb, in_b=t(fb), scale=t(sc), shift=t(sh),
                         residual=t(res), relu=True, l2norm=l2, variant=variant, split_k=1, staging=staging).cpu().numpy()
        bound = 4e-6 * (np.abs(ref).max() + 27 * (ca + cb) ** 0.5 * 0.3) if not l2 else 2e-6
        assert np.abs(got - ref).max() < max(bound, 2e-5 if not l2 else 2e-6), (variant, ca, cb, cout, float(np.abs(got - ref).max()))
