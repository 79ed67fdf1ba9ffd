"""Multi-GPU layer: one process per GPU, fragments sharded across ranks, ONE exchange at the end
(SURVEY §8e).  The reference has no distributed code at all (scripts/generate_desc.py:8 pins
CUDA_VISIBLE_DEVICES="0"); fragments are independent units (eval-mode BatchNorm, no cross-fragment
state), so the path shards embarrassingly and the only collective is the final variable-length
gather of [M_i, 32] descriptor blocks to rank 0 -- `torch.distributed` send/recv, which is RCCL
over xGMI with backend "nccl" (each peer reaches the root over its own direct link) and gloo on CPU
for the tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns
    (rank, world_size, local_rank); a single process without the env vars is (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # single-GPU stand-in for a multi-GPU node (tests, tools/emulate_3dmatch.py): IMF_DIST_BACKEND=gloo
    # IMF_FORCE_DEVICE=0 runs N ranks on one device
    backend = os.environ.get("IMF_DIST_BACKEND", backend)
    forced = os.environ.get("IMF_FORCE_DEVICE")
    # IMF_DIST_FORCE_INIT=1: a process group also for ONE rank (tests: the RCCL code paths on a single-GPU box)
    if (world > 1 or os.environ.get("IMF_DIST_FORCE_INIT") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl" and forced is None:
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if forced is not None:
        local = int(forced)
    return rank, world, local


def shard_fragments(costs, world_size):
    """Longest-processing-time-first assignment of fragments to ranks.  costs[i] ~ work of fragment
    i (point or voxel count).  Returns world_size lists of fragment indices (each ascending);
    deterministic, every fragment assigned exactly once."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def _exchange(sends, recvs, group=None):
    """All point-to-point transfers of one exchange posted AT ONCE: `batch_isend_irecv` = one grouped
    ncclSend / ncclRecv launch under RCCL (every peer -> root transfer runs on its own xGMI link concurrently),
    independent isend / irecv under gloo.  sends: [(tensor, dst)], recvs: [(tensor, src)]."""
    ops = [dist.P2POp(dist.isend, t, d, group) for t, d in sends] + [dist.P2POp(dist.irecv, t, r, group) for t, r in recvs]
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def gather_blocks(block, dst=0, group=None):
    """Variable-length gather of 2-D row blocks (same dtype / column count, different row counts)
    to rank `dst`.  Returns the list of per-rank tensors on `dst` (rank order), None elsewhere.  One count
    all_gather, then every non-empty block in ONE grouped exchange (the root posts all receives together)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [block]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    block = block.contiguous()
    n = torch.tensor([block.shape[0]], dtype=torch.int64, device=block.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = torch.cat(counts).tolist()
    if rank != dst:
        if block.shape[0]:
            _exchange([(block, dst)], [], group)
        return None
    out = [block if r == dst else torch.empty((counts[r],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
           for r in range(world)]
    _exchange([], [(out[r], r) for r in range(world) if r != dst and counts[r]], group)
    return out


def _collective_device():
    """Where tensors of a collective must live: the rank's GPU under the nccl (RCCL) backend, the host under gloo."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_fragment_descriptors(results, n_fragments, shards, dst=0, device=None, dtype=torch.float32, packed=None):
    """results: {fragment index: F [M_i, D]} computed by this rank (its shard).  Gathers every
    rank's blocks to `dst` and returns {fragment index: F} for ALL fragments there (None elsewhere).
    ONE all_gather (per-fragment row counts + D, a fixed-size table) and ONE grouped exchange of the blocks,
    independent of the number of fragments; the root posts every receive at once.
    packed = (rows, feats): the shard's blocks ALREADY concatenated in shard order (`feats` [sum(rows), D] on the
    collective's device -- e.g. filled block by block from the capacity buckets as the fragments complete) -- nothing is
    copied or moved before the send.  A rank whose shard is empty (more ranks than fragments) still takes part, on
    `device` (default: the backend's)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = shards[rank]
    if packed is not None:
        rows, feats = [int(r) for r in packed[0]], packed[1]
        dev = device if device is not None else feats.device
        assert len(rows) == len(mine) and feats.shape[0] == sum(rows), "packed: rows / feats do not match the shard"
        dtype = feats.dtype
    else:
        dev = device if device is not None else (next(iter(results.values())).device if results else _collective_device())
        if results:
            dtype = next(iter(results.values())).dtype
        rows = [int(results[i].shape[0]) for i in mine]
        D0 = next(iter(results.values())).shape[1] if results else 0
        feats = torch.cat([results[i] for i in mine], 0).to(dev) if mine else torch.empty((0, D0), dtype=dtype, device=dev)
    D = int(feats.shape[1]) if feats.dim() == 2 else 0

    def split(table_rows, blocks):
        out = {}
        for r in range(world):
            start = 0
            for i, m in zip(shards[r], table_rows[r]):
                out[i] = blocks[r][start:start + m]
                start += m
        assert len(out) == n_fragments
        return out

    if world == 1 and not dist.is_initialized():
        return split([rows], [feats])
    # (one rank WITH a process group takes the collective path too: the table's all_gather runs under the backend -- RCCL on
    # the rank's GPU -- and the exchange below has no peers)
    L = max(len(s) for s in shards)
    t = torch.zeros(L + 1, dtype=torch.int64)
    t[0] = D
    if rows:
        t[1:1 + len(rows)] = torch.tensor(rows, dtype=torch.int64)
    t = t.to(dev)
    table = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(table, t)
    table = torch.stack(table).cpu()                     # (one readback: every rank now knows every fragment's row count)
    D = int(table[:, 0].max())
    all_rows = [table[r, 1:1 + len(shards[r])].tolist() for r in range(world)]
    if rank != dst:
        if feats.shape[0]:
            _exchange([(feats.contiguous(), dst)], [])
        return None
    blocks = [feats if r == dst else torch.empty((sum(all_rows[r]), D), dtype=dtype, device=dev) for r in range(world)]
    _exchange([], [(blocks[r], r) for r in range(world) if r != dst and blocks[r].shape[0]])
    return split(all_rows, blocks)
