"""Capacity mode and whole-fragment hipGraphs (csrc/executor.hip imf_fragment_forward, model/graph.py):
the device-side-count path must reproduce the exact (host-count) path BIT FOR BIT -- descriptors, voxel order,
first-point indices, counts -- eagerly and as a replayed graph, for single fragments and batches, across
fragments of different size sharing one capacity bucket; overflow must be flagged, never silent."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model(seeded_sd):
    from imfnet_amd.model import load_model
    m = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3, config=None)
    m.load_state_dict(seeded_sd, strict=True)
    return m.eval().to(DEV)


def _exact(model, pts_list, imgs, voxel):
    """The exact path: geometry with one count readback, NativePlan forward.  Returns (F, first_idx, counts, bbox, items)."""
    from imfnet_amd.extract import sparse_tensor_from_points, start_geometry
    dev = torch.device(DEV)
    with torch.no_grad():
        fut = start_geometry([torch.as_tensor(p).to(dev) for p in pts_list], voxel, dev)
        st, inds = sparse_tensor_from_points(None, voxel, dev, geometry=fut)
        F = model(st, torch.as_tensor(imgs).to(dev)).F
    torch.cuda.synchronize()
    cm = st.coordinate_manager
    lv = [cm.level(ts) for ts in (1, 2, 4, 8)]
    return F.clone(), inds.clone(), [l.n for l in lv], lv[0].bbox, lv[0].items, lv[0].coords.clone()


def _runner(model):
    from imfnet_amd.model.graph import FragmentRunner
    r = FragmentRunner(model)
    assert r.supported
    r.use_graph = True
    return r


def _cat(pts_list):
    starts, n = [], 0
    for p in pts_list:
        starts.append(n)
        n += len(p)
    return torch.as_tensor(np.concatenate(pts_list, 0)).to(DEV), starts


@pytest.mark.parametrize("use_graph", [False, True])
def test_capacity_mode_equals_exact_single(model, clouds, images, use_graph):
    pts = [clouds[0].astype(np.float64)]
    F, inds, counts, bbox, items, coords = _exact(model, pts, images[0], 0.05)
    r = _runner(model)
    r.use_graph = use_graph
    assert r.run(*_cat(pts), torch.as_tensor(images[0]).to(DEV), 0.05) is None      # no capacities known yet
    r.observe(len(pts[0]), counts, bbox)
    stream = torch.cuda.Stream()
    xyz, starts = _cat(pts)
    res = r.run(xyz, starts, torch.as_tensor(images[0]).to(DEV), 0.05, stream=stream)
    assert res.flags == 0 and res.counts == counts and res.bbox == list(bbox)
    assert torch.equal(res.first_idx, inds)
    assert torch.equal(res.F, F)                                  # bit-identical descriptors
    b = res.bucket
    assert tuple(b.caps.rows) >= tuple(counts) and b.caps.rows[0] < 2 * counts[0]
    if use_graph:
        assert b.n_nodes > 80 and r.stats["captured"] == 1
        res2 = r.run(xyz, starts, torch.as_tensor(images[0]).to(DEV), 0.05, stream=stream)
        assert torch.equal(res2.F, F) and r.stats["captured"] == 1 and r.stats["graph"] == 2


def test_graph_replay_on_different_fragments_of_one_bucket(model, clouds, images):
    """One capture, then fragments of different size and geometry through the same graph: each equals its exact result."""
    r = _runner(model)
    r.MARGIN = 1.3
    cases = []
    for k, scale in ((0, 1.0), (1, 1.0), (0, 0.93), (1, 1.04)):
        pts = [clouds[k].astype(np.float64) * scale]
        cases.append((pts, images[k], _exact(model, pts, images[k], 0.05)))
        r.observe(len(pts[0]), cases[-1][2][2], cases[-1][2][3])
    stream = torch.cuda.Stream()
    keys = set()
    for pts, img, (F, inds, counts, bbox, items, coords) in cases:
        xyz, starts = _cat(pts)
        res = r.run(xyz, starts, torch.as_tensor(img).to(DEV), 0.05, stream=stream)
        keys.add(res.bucket.key)
        assert res.flags == 0 and res.counts == counts
        assert torch.equal(res.first_idx, inds) and torch.equal(res.F, F)
    assert len(keys) == 1 and r.stats["captured"] == 1 and r.stats["graph"] == len(cases)


def test_graph_batched_pair_equals_exact(model, clouds, images):
    pts = [clouds[0].astype(np.float64), clouds[1].astype(np.float64)]
    imgs = np.concatenate([images[0], images[1]], 0)
    F, inds, counts, bbox, items, coords = _exact(model, pts, imgs, 0.05)
    r = _runner(model)
    r.observe(sum(len(p) for p in pts), counts, bbox)
    xyz, starts = _cat(pts)
    res = r.run(xyz, starts, torch.as_tensor(imgs).to(DEV), 0.05, stream=torch.cuda.Stream())
    assert res.flags == 0 and res.counts == counts and res.items(0) == [tuple(i) for i in items]
    assert torch.equal(res.F, F) and torch.equal(res.first_idx, inds)
    # fine voxel size (more levels split differently): the 2.5 cm pair
    F, inds, counts, bbox, items, coords = _exact(model, pts, imgs, 0.025)
    r.observe(sum(len(p) for p in pts), counts, bbox)
    res = r.run(xyz, starts, torch.as_tensor(imgs).to(DEV), 0.025, stream=torch.cuda.Stream())
    assert res.flags == 0 and res.counts == counts and torch.equal(res.F, F)


@pytest.mark.parametrize("nb", [3, 4])
def test_graph_batches_of_three_and_four_equal_exact(model, clouds, images, nb):
    """The executors' kernel policy depends on the batch size (imf_resunet_conv_kernel_tag: 48-row units on the stride-8 level
    from two fragments on, 4-wavefront workgroups on the stride-4 level from three on): capacity mode must still equal the exact
    path bit for bit, because the batch size is static in both."""
    pts = [clouds[k % 2].astype(np.float64) * (1.0 + 0.07 * k) for k in range(nb)]
    imgs = np.concatenate([images[k % 2] for k in range(nb)], 0)
    F, inds, counts, bbox, items, coords = _exact(model, pts, imgs, 0.025)
    r = _runner(model)
    r.observe(sum(len(p) for p in pts), counts, bbox)
    xyz, starts = _cat(pts)
    res = r.run(xyz, starts, torch.as_tensor(imgs).to(DEV), 0.025, stream=torch.cuda.Stream())
    assert res.flags == 0 and res.counts == counts and res.items(0) == [tuple(i) for i in items]
    assert torch.equal(res.F, F) and torch.equal(res.first_idx, inds)


def test_sorted_twins_keep_the_modes_bit_identical():
    """The decoder's blocks walk occupancy-sorted twins of the stride-1 maps of levels 0-2 (csrc/executor.hip,
    csrc/rulebook_sort.hip; default IMF_SORTED_MAP=7, 0 = the maps as built): in a process of its own for each setting (the
    switch is read once) the native executor equals the op-by-op
    Python plan bit for bit, capacity mode equals the exact path for one fragment and for the pair, and the descriptors stay
    within round-off of the default map's (same terms, other partition)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "oracle"))
        import bench
        from imfnet_amd.extract import sparse_tensor_from_points
        dev = torch.device("cuda:0")
        model, sd = bench.build_model(dev)
        pts, imgs = bench.load_pair(1.0)
        with torch.no_grad():
            for k in (0, 1):
                img = torch.as_tensor(imgs[k:k + 1]).to(dev)
                os.environ["IMFNET_PYTHON_EXECUTOR"] = "1"
                st, _ = sparse_tensor_from_points(pts[k], 0.025, dev)
                a = model(st, img).F.clone()
                del os.environ["IMFNET_PYTHON_EXECUTOR"]
                st, _ = sparse_tensor_from_points(pts[k], 0.025, dev)
                b = model(st, img).F.clone()
                assert torch.equal(a, b), "native executor != python plan"
            for sel in ([0], [0, 1]):
                wl = bench.Workload(model, dev, [pts[i] for i in sel], imgs[sel], 0.025)
                F = wl.prepare_graph().clone()
                wl.runner.use_graph = False
                r = wl.graph_step(); torch.cuda.synchronize()
                assert r.flags == 0 and torch.equal(r.F, F), "capacity mode != exact path"
                w = (torch.arange(F.numel(), device=dev, dtype=torch.float64) %% 7 + 1).view_as(F)
                print("F", len(sel), repr(float((F.double() * w).sum())), flush=True)
        print("OK")
    """) % (ROOT, ROOT, ROOT)
    outs = {}
    for flag in ("7", "1", "0"):
        env = dict(os.environ, IMF_SORTED_MAP=flag)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
        outs[flag] = [float(l.split()[2]) for l in p.stdout.splitlines() if l.startswith("F ")]
    for flag in ("7", "1"):
        for a, b in zip(outs[flag], outs["0"]):
            assert abs(a - b) <= 1e-6 * abs(b) + 1e-3            # the same descriptors to round-off ...
        assert outs[flag] != outs["0"]                           # ... formed over another partition: the switch took effect
    assert outs["7"] != outs["1"]


def test_capacity_overflow_is_flagged(model, clouds, images):
    """Rows beyond a level's capacity: count clamped, flag 2 raised (never a silent wrong result or a fault)."""
    pts = [clouds[0].astype(np.float64)]
    F, inds, counts, bbox, items, coords = _exact(model, pts, images[0], 0.05)
    r = _runner(model)
    r.observe(len(pts[0]), [c // 3 for c in counts], bbox)        # pretend fragments are three times sparser
    r.MARGIN = 1.0
    xyz, starts = _cat(pts)
    res = r.run(xyz, starts, torch.as_tensor(images[0]).to(DEV), 0.05, stream=torch.cuda.Stream())
    assert res.flags & 2
    assert res.counts[0] == res.bucket.caps.rows[0] < counts[0]
    # the rows that fit are the first voxels in first-occurrence order
    assert torch.equal(res.first_idx, inds[: res.counts[0]])
    # too small a bit grid: flag 4
    r2 = _runner(model)
    r2.observe(len(pts[0]), counts, bbox)
    r2.grid_words = 16
    import imfnet_amd.model.graph as G
    key = list(r2.caps_for(len(pts[0]), 1, 120, 160, 0.05, True))
    key[5] = 64
    b = r2.bucket(tuple(key), torch.device(DEV))
    s = torch.cuda.Stream()
    n = r2.stage(b, xyz, starts, torch.as_tensor(images[0]).to(DEV), s)
    r2.use_graph = False
    assert r2.launch(b, n, 1, s).flags & 4
    assert G.FLAG_NAMES[4]


def test_pyramid_dyn_matches_exact_pyramid(clouds):
    """imf_pyramid_build_dyn on its own: counts, coordinates, first indices, bounding box, item starts."""
    import ctypes as C
    from imfnet_amd import _lib, ops
    L = _lib.lib()
    pts = [clouds[0].astype(np.float64), clouds[1].astype(np.float64) * 1.1]
    xyz, starts = _cat(pts)
    fut = ops.PyramidFuture(xyz, 0.05, 4, 0, inputs_ready=False, item_starts=starts)
    lv = fut.result()
    n_cap = 600000
    caps = (C.c_int64 * 4)(16384, 4096, 2048, 512)
    nbytes = L.imf_pyramid_arena_bytes_caps(n_cap, 4, caps)
    arena = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    meta = torch.zeros(_lib.META_WORDS, dtype=torch.int32, device=DEV)
    dyn = torch.tensor([xyz.shape[0], 2] + starts + [0] * 12, dtype=torch.int32, device=DEV)
    buf = torch.zeros((n_cap, 3), dtype=torch.float64, device=DEV)
    buf[: xyz.shape[0]] = xyz
    descs = (_lib.LevelDesc * 4)()
    _lib.check(L.imf_pyramid_build_dyn(buf.data_ptr(), 1, dyn.data_ptr(), n_cap, caps, 0.05, 4, arena.data_ptr(), nbytes,
                                       meta.data_ptr(), descs, torch.cuda.current_stream().cuda_stream), "dyn")
    m = meta.cpu().numpy()
    assert [int(m[2 * l]) for l in range(4)] == [l.n for l in lv] and not any(m[2 * l + 1] for l in range(4))
    assert list(m[8:16]) == list(lv[0].bbox)
    for l in range(4):
        off = descs[l].coords - arena.data_ptr()
        got = arena[off:off + 16 * lv[l].n].view(torch.int32).view(-1, 4)
        assert torch.equal(got, lv[l].coords)
        assert [int(v) for v in m[16 + 8 * l:16 + 8 * l + 2]] == [s for s, _ in lv[l].items]


def test_strict_fp32_capacity_mode_equals_its_exact_mode(seeded_sd, clouds, images):
    """Variant 0 (v_mfma_f32_16x16x4_f32 in every convolution, the image trunk and the fusion feed-forward) through
    imf_fragment_forward: bit-identical to the variant-0 exact mode, within 1e-4 of the oracle and within 2e-6 of the
    default path (bf16x3 since round 5); a variant-0 bucket does not report IMF_FLAG_RANGE (nothing there is an f16 operand)."""
    import bench
    import imf_oracle as O
    from imfnet_amd.model.graph import FragmentRunner
    dev = torch.device(DEV)
    m0, sd = bench.build_model(dev, variant=0)
    pts = [clouds[0][::3].astype(np.float64), clouds[1][::4].astype(np.float64)]
    imgs = np.concatenate([images[0], images[1]], 0)
    wl = bench.Workload(m0, dev, pts, imgs, 0.05)
    with torch.no_grad():
        F_exact = wl.prepare_graph().clone()
        assert wl.runner.variant == 0 and wl.runner.supported and wl.bucket.ignore_flags == 32
        wl.runner.use_graph = False
        res = wl.graph_step()
        torch.cuda.synchronize()
        assert res.flags == 0 and torch.equal(res.F, F_exact)
        wl.runner.use_graph = True                                   # and as a replayed hipGraph
        assert torch.equal(wl.graph_step().F, F_exact)
        m6, _ = bench.build_model(dev)
        F6 = bench.Workload(m6, dev, pts, imgs, 0.05).exact_step()   # (issued on the workload's own stream)
        torch.cuda.synchronize()
    assert float((F6 - F_exact).abs().max()) < 2e-6
    n0 = res.items()[0][1]
    _, F_ref = O.extract_features(sd, pts[0], 0.05, images[0])
    assert np.abs(F_exact[:n0].cpu().numpy() - F_ref.numpy()).max() < 1e-4


def test_side_chain_issue_order_does_not_change_the_descriptors(clouds):
    """imf_fragment_io.gpu_idle_hint (round 6): with nothing queued ahead the executor issues the side stream's pieces right
    before the first launch that waits for each, otherwise all of them ahead of conv1 -- same streams, same events, same order
    within each stream: the descriptors of a pair and of one fragment are the same bits either way (IMF_EAGER_SIDE forces the
    order per call), and both equal the exact path."""
    import bench
    dev = torch.device("cuda:0")
    model, _ = bench.build_model(dev)
    pts, imgs = bench.load_pair(1.0)
    old = os.environ.get("IMF_EAGER_SIDE")
    try:
        with torch.no_grad():
            for sel in ([0], [0, 1]):
                wl = bench.Workload(model, dev, [pts[i] for i in sel], imgs[sel], 0.025)
                F = wl.prepare_graph().clone()
                wl.runner.use_graph = False
                got = {}
                for flag in ("1", "0"):
                    os.environ["IMF_EAGER_SIDE"] = flag
                    for _ in range(2):                                     # (twice: an idle GPU, then work queued ahead)
                        r = wl.graph_step()
                    torch.cuda.synchronize()
                    assert r.flags == 0
                    got[flag] = r.F.clone()
                assert torch.equal(got["1"], got["0"]) and torch.equal(got["1"], F), sel
    finally:
        if old is None:
            os.environ.pop("IMF_EAGER_SIDE", None)
        else:
            os.environ["IMF_EAGER_SIDE"] = old
