"""bench.py's multi-rank path on ONE GPU: two ranks under gloo (IMF_DIST_BACKEND=gloo IMF_FORCE_DEVICE=0) run the replica
steps, the host-array stream and the sharded pipeline + gather leg; rank 0's line must carry all three."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_one_gpu_under_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMF_DIST_BACKEND="gloo", IMF_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--settle-ms", "50", "--mode", "capacity", "--no-cpu-baseline", "--no-extras",
           "--sharded-per-rank", "12", "--sharded-region-s", "0.05", "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = r.stdout.splitlines()[-1]                      # the compact line is the LAST stdout line, and small
    c = json.loads(line)
    assert len(line) <= 4096 and c["n_gpus"] == 2 and c["value"] > 0
    assert c["rccl"]["rccl_ranks"] == 2 and len(c["rccl"]["per_rank_ms_per_step"]) == 2 and c["rccl"]["gather_crc_ok"] is True
    assert c["sharded_pipeline"]["ranks"] == 2 and c["host_span"]["n_gpus"] == 2
    d = json.load(open(tmp_path / "full.json"))          # the full record
    assert d["value"] == c["value"] and d["roofline"]["per_kernel"]
    assert d["n_gpus"] == 2 and d["config"]["fragments_per_step"] == 4 and d["value"] > 0
    hs = d["host_span"]
    assert hs and hs["n_gpus"] == 2 and hs["value"] > 0
    sp = d["config"]["sharded_pipeline"]
    assert sp["ranks"] == 2 and sp["backend"] == "gloo" and sp["fragments"] == 24
    assert sp["verified"].startswith("every block") and sp["descriptors"] > 0 and sp["gather_ms"] > 0
    assert sp["gather"].startswith("device-resident") and len(sp["stream_s_all_passes"]) == 3
