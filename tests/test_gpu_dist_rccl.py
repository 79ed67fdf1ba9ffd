"""RCCL on the one GPU a test box has (the N > 1 exchange has only ever run under gloo).  Two RCCL ranks
cannot share a device, so what CAN run here runs: a process group of ONE rank on backend "nccl" (= RCCL on ROCm), through
  * imfnet_amd.dist.gather_fragment_descriptors(packed=...) -- the device-tensor branch: the row-count table's all_gather
    on the GPU under RCCL, blocks that never leave the device (the grouped send / recv has no peer at one rank);
  * bench.py's multi-rank code path: barrier(device_ids), all_reduce MAX / SUM, all_gather of the per-rank times, the
    sharded-pipeline leg's gather with its CRC check -- `rccl.rccl_ranks` in the compact line is what the group reports.
Each in a process of its own (a process group must not leak into the test process)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                IMF_DIST_FORCE_INIT="1", HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_gather_of_packed_device_blocks_under_rccl_world_one():
    code = textwrap.dedent("""
        import sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from imfnet_amd import dist as idist
        rank, world, local = idist.init_from_env("nccl")
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        dev = idist._collective_device()
        assert dev.type == "cuda"
        dist.barrier(device_ids=[local])
        t = torch.tensor([3.0, 5.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.tolist() == [3.0, 5.0]
        rows = [5, 0, 1200, 7]                                  # ragged blocks, an empty one
        shards = idist.shard_fragments([r + 1 for r in rows], 1)
        feats = torch.arange(sum(rows) * 32, dtype=torch.float32, device=dev).view(-1, 32)
        order = shards[0]
        got = idist.gather_fragment_descriptors(None, len(rows), shards, dst=0, packed=([rows[i] for i in order], feats))
        torch.cuda.synchronize()
        at = 0
        for i in order:
            assert got[i].device.type == "cuda" and torch.equal(got[i], feats[at:at + rows[i]])
            assert got[i].data_ptr() == feats[at:at + rows[i]].data_ptr() or rows[i] == 0   # views of the send buffer: nothing copied
            at += rows[i]
        blocks = idist.gather_blocks(feats[:9], dst=0)
        assert len(blocks) == 1 and torch.equal(blocks[0], feats[:9])
        dist.destroy_process_group()
        print("OK")
    """) % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_bench_multi_rank_path_under_rccl_world_one(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--repeats", "2",
           "--settle-ms", "50", "--mode", "capacity", "--no-cpu-baseline", "--no-extras", "--sharded-per-rank", "12",
           "--sharded-region-s", "0.05", "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    c = json.loads(r.stdout.splitlines()[-1])
    assert c["n_gpus"] == 1 and c["value"] > 0
    assert c["rccl"] == {"backend": "nccl", "process_group": True, "rccl_ranks": 1,
                         "per_rank_ms_per_step": [c["ms_per_step"]], "gather_crc_ok": True}
    sp = c["sharded_pipeline"]
    assert sp["ranks"] == 1 and sp["backend"] == "nccl" and sp["fragments"] == 12 and sp["gather_crc_ok"] is True
