"""the `imf_cpu_*` twins (oracle/imf_cpu_twins.c -- the C ABI's geometry and convolution entry points with the
same signatures on host pointers).  Here (CPU): the twins against the numpy / C oracle.  tests/test_gpu_cpu_twins.py holds the
HIP library's outputs against the twins directly (same layouts: integer outputs equal bit for bit)."""
import ctypes as C

import numpy as np
import torch

import imf_cpu_twins as T
import imf_oracle as O


def _nbr_by_row(rows, nbr, n_out):
    valid = rows >= 0
    assert sorted(rows[valid].tolist()) == list(range(n_out))
    out = np.empty((n_out, nbr.shape[0]), np.int32)
    out[rows[valid]] = nbr[:, valid].T
    assert (nbr[:, ~valid] == -1).all()
    return out


def pack_weights_f32(w):
    """imf_pack_weights' fragment-major fp32 image of W [kvol, cin, cout] (tests/test_gpu_parity.py::test_pack_weights_layout)."""
    kvol, cin, cout = w.shape
    CI = 64 if cin % 64 == 0 else 32
    J, CB = CI // 16, (4 if cout % 64 == 0 else 2)
    return (w.reshape(kvol, cin // CI, J, 4, 4, cout // (16 * CB), CB, 16).transpose(5, 0, 1, 2, 6, 3, 7, 4).reshape(-1)).copy()


def test_geometry_twins_match_the_oracle(clouds):
    xyz = clouds[0].astype(np.float64)
    lv0, err = T.voxelize(xyz, 0.05)
    coords_ref, inds_ref = O.voxelize(xyz, 0.05)
    assert err == 0 and (lv0.coords == coords_ref).all() and (lv0.first_idx == inds_ref).all()
    lv32, _ = T.voxelize(xyz.astype(np.float32), 0.05)          # float32 input is widened before the fp64 divide
    c32, _ = O.voxelize(xyz.astype(np.float32).astype(np.float64), 0.05)
    assert (lv32.coords == c32).all()
    g = O.Geometry(coords_ref)
    levels = [lv0]
    for i in range(3):
        levels.append(T.downsample(levels[-1], 2 << i))
        assert (levels[-1].coords == g.levels[i + 1]).all()
    rows, nbr, mask = T.rulebook_conv(levels[0], levels[0], 1, 5)
    assert (_nbr_by_row(rows, nbr, levels[0].n) == g.k_first).all()
    for i in range(4):
        rows, nbr, mask = T.rulebook_conv(levels[i], levels[i], 1 << i, 3)
        assert (_nbr_by_row(rows, nbr, levels[i].n) == g.k3[i]).all() and (rows[:levels[i].n] == np.arange(levels[i].n)).all()
        act = (nbr.reshape(27, -1, 64) >= 0).any(2)
        for k in range(27):
            assert (((mask[:, k // 32] >> (k % 32)) & 1).astype(bool) == act[k]).all()
    for i in range(3):
        rows, nbr, _ = T.rulebook_conv(levels[i], levels[i + 1], 1 << i, 3)
        assert (_nbr_by_row(rows, nbr, levels[i + 1].n) == g.down[i]).all()
        rows, nbr, mask = T.rulebook_transpose(levels[i + 1], levels[i], 1 << i)
        assert (_nbr_by_row(rows, nbr, levels[i].n) == g.up[i]).all()
        pop = np.array([bin(int(w)).count("1") for w in mask.reshape(-1)]).reshape(-1, 4).sum(1)
        assert pop.max() <= 8                               # parity-class grouping: at most 8 offsets per tile


def test_sorted_map_twin_matches_its_numpy_restatement(clouds):
    """imf_cpu_rulebook_sort_by_occupancy (the twin of csrc/rulebook_sort.hip) against numpy's stable argsort of the same key,
    on a map spanning several 16 k-row windows: same permutation, gathered table and tile masks; every row appears once."""
    xyz = np.concatenate([clouds[0], clouds[1] + 3.0]).astype(np.float64)
    lv0, err = T.voxelize(xyz, 0.02)
    rows, nbr, mask = T.rulebook_conv(lv0, lv0, 1, 3)
    n, S, K = lv0.n, len(rows), 27
    assert err == 0 and n > 2 * 16384
    srows, snbr, smask = T.rulebook_sort_by_occupancy(nbr, n)
    perm, valid = O.occupancy_sorted_slots(nbr, n)
    assert (srows == np.where(valid, perm, -1)).all() and sorted(srows[:n].tolist()) == list(range(n))
    assert (snbr == np.where(valid[None, :], nbr[:, perm], -1)).all()
    act = (snbr.reshape(K, -1, 64) >= 0).any(2)
    assert (smask[:, 0] == (act.astype(np.uint32) << np.arange(K, dtype=np.uint32)[:, None]).sum(0)).all() and not smask[:, 1:].any()
    assert act.mean() < 0.9 * (nbr.reshape(K, -1, 64) >= 0).any(2).mean()        # fewer active (tile, offset) pairs


def test_convolution_twin_matches_the_oracle(clouds):
    from imfnet_amd._lib import ConvArgs
    xyz = clouds[0][::3].astype(np.float64)
    lv0, _ = T.voxelize(xyz, 0.05)
    lv1 = T.downsample(lv0, 2)
    rng = np.random.default_rng(0)
    for (ca, cb, cout, in_lv, out_lv, ts, transposed) in ((32, 0, 64, lv0, lv0, 1, False), (64, 32, 32, lv0, lv0, 1, False),
                                                          (32, 0, 64, lv0, lv1, 1, False), (64, 0, 32, lv1, lv0, 1, True)):
        rows, nbr, mask = T.rulebook_transpose(in_lv, out_lv, ts) if transposed else T.rulebook_conv(in_lv, out_lv, ts, 3)
        n_in, n_out = in_lv.n, out_lv.n
        fa = rng.normal(size=(n_in, ca)).astype(np.float32)
        fb = rng.normal(size=(n_in, cb)).astype(np.float32) if cb else None
        w = (rng.normal(size=(27, ca + cb, cout)) / np.sqrt(27 * (ca + cb))).astype(np.float32)
        sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
        res = rng.normal(size=(n_out, cout)).astype(np.float32)
        wp, out = pack_weights_f32(w), np.zeros((n_out, cout), np.float32)
        a = ConvArgs()
        a.in_a, a.c_a, a.in_b, a.c_b = fa.ctypes.data, ca, (fb.ctypes.data if cb else None), cb
        a.w_packed, a.kvol, a.cout = wp.ctypes.data, 27, cout
        a.tile_rows, a.nbr, a.tile_mask = rows.ctypes.data, nbr.ctypes.data, mask.ctypes.data
        a.n_slots, a.n_out = len(rows), n_out
        a.scale, a.shift, a.residual, a.relu, a.l2norm = sc.ctypes.data, sh.ctypes.data, res.ctypes.data, 1, int(cout == 32)
        a.out, a.variant, a.split_k = out.ctypes.data, 0, 1
        T.spconv_fwd(a)
        fin = torch.as_tensor(fa if fb is None else np.concatenate([fa, fb], 1))
        ref = O.spconv(fin, torch.as_tensor(w), _nbr_by_row(rows, nbr, n_out))
        ref = torch.relu(ref * torch.as_tensor(sc) + torch.as_tensor(sh) + torch.as_tensor(res))
        if cout == 32:
            ref = ref / ref.norm(dim=1, keepdim=True)
        assert np.abs(out - ref.numpy()).max() < 2e-5, (ca, cb, cout, transposed)
