"""The N>1 path on CPU: world_size-2 gloo processes exercise fragment sharding and the single
variable-length gather of descriptor blocks (imfnet_amd/dist.py) -- the same code that runs over
RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_fragments_lpt():
    from imfnet_amd.dist import shard_fragments
    costs = [50, 10, 40, 30, 20, 60, 5, 5]
    shards = shard_fragments(costs, 3)
    assert sorted(sum(shards, [])) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= 20 and max(loads) <= 80
    assert shard_fragments(costs, 3) == shards                     # deterministic
    assert shard_fragments([1, 2], 4)[2:] == [[], []]              # more ranks than fragments
    assert shard_fragments(costs, 1) == [list(range(8))]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from imfnet_amd import dist as idist
    r, w, _ = idist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n_frag = 7
    sizes = [13, 1, 29, 8, 0, 17, 5]                                 # ragged, one empty block
    shards = idist.shard_fragments([s + 1 for s in sizes], world)

    def desc(i):                                                   # stand-in for the GPU forward
        g = torch.Generator().manual_seed(100 + i)
        return torch.randn(sizes[i], 32, generator=g)

    mine = {i: desc(i) for i in shards[rank]}
    got = idist.gather_fragment_descriptors(mine, n_frag, shards, dst=0)
    # the shard's blocks already packed in one buffer (what the sharded bench leg sends: filled from the capacity buckets
    # as the fragments complete) -- same result, nothing concatenated inside
    rows = [sizes[i] for i in shards[rank]]
    buf = torch.cat([mine[i] for i in shards[rank]], 0) if rows else torch.empty((0, 32))
    got_packed = idist.gather_fragment_descriptors(None, n_frag, shards, dst=0, packed=(rows, buf))
    blocks = idist.gather_blocks(torch.full((rank + 2, 3), float(rank)), dst=0)
    if rank == 0:
        assert sorted(got) == list(range(n_frag)) == sorted(got_packed)
        for i in range(n_frag):
            assert torch.equal(got[i], desc(i)), i                 # bit-exact, right order, right counts
            assert torch.equal(got_packed[i], desc(i)), i
        assert [b.shape[0] for b in blocks] == [r + 2 for r in range(world)]
        assert all(float(b.mean()) == r for r, b in enumerate(blocks))
        np.save(os.path.join(out_dir, "ok.npy"), np.array([1]))
    else:
        assert got is None and blocks is None and got_packed is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_more_ranks(rank, world, port, out_dir):
    """More ranks than fragments: the ranks with an empty shard still take part in both exchanges."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from imfnet_amd import dist as idist
    idist.init_from_env("gloo")
    shards = idist.shard_fragments([5, 9], world)
    assert sum(1 for s in shards if not s) == world - 2
    mine = {i: torch.full((i + 3, 32), float(i)) for i in shards[rank]}
    got = idist.gather_fragment_descriptors(mine, 2, shards, dst=0)
    rows = [i + 3 for i in shards[rank]]
    buf = torch.cat([mine[i] for i in shards[rank]], 0) if rows else torch.empty((0, 32))
    got_packed = idist.gather_fragment_descriptors(None, 2, shards, dst=0, packed=(rows, buf), device=torch.device("cpu"))
    if rank == 0:
        assert sorted(got) == [0, 1] and got[0].shape == (3, 32) and float(got[1].mean()) == 1.0
        assert all(torch.equal(got[i], got_packed[i]) for i in (0, 1))
        np.save(os.path.join(out_dir, "ok2.npy"), np.array([1]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_with_more_ranks_than_fragments(tmp_path):
    mp.spawn(_worker_more_ranks, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert os.path.exists(tmp_path / "ok2.npy")


@pytest.mark.parametrize("world", [2, 3])
def test_gather_over_gloo(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def test_single_process_gather_is_identity():
    from imfnet_amd.dist import gather_blocks, gather_fragment_descriptors
    x = torch.arange(6.).reshape(3, 2)
    assert gather_blocks(x)[0] is x
    out = gather_fragment_descriptors({0: x, 1: x * 2}, 2, [[0, 1]])
    assert torch.equal(out[1], x * 2)


def _worker_one_rank_group(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      IMF_DIST_FORCE_INIT="1")
    from imfnet_amd import dist as idist
    r, w, _ = idist.init_from_env("gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    rows = [3, 0, 7]
    shards = idist.shard_fragments([4, 1, 8], 1)
    feats = torch.arange(sum(rows) * 4, dtype=torch.float32).view(-1, 4)
    got = idist.gather_fragment_descriptors(None, 3, shards, dst=0, packed=([rows[i] for i in shards[0]], feats))
    at = 0
    for i in shards[0]:
        assert torch.equal(got[i], feats[at:at + rows[i]])
        at += rows[i]
    res = {i: torch.full((rows[i], 4), float(i)) for i in range(3)}
    got = idist.gather_fragment_descriptors(res, 3, shards, dst=0)
    assert all(torch.equal(got[i], res[i]) for i in range(3))
    assert torch.equal(idist.gather_blocks(feats, dst=0)[0], feats)
    dist.destroy_process_group()
    open(os.path.join(out_dir, "ok"), "w").write("1")


def test_one_rank_with_a_process_group_takes_the_collective_path(tmp_path):
    """IMF_DIST_FORCE_INIT=1 (round 6): a process group of ONE rank -- what lets the RCCL half run on a single-GPU box
    (tests/test_gpu_dist_rccl.py) -- goes through the table's all_gather and the (peer-less) exchange, not the early return."""
    mp.spawn(_worker_one_rank_group, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok")
