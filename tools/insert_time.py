"""k_insert_points alone (+ the table reset before it) on the bench's pair: 527 k points, f64.  usage: [IMF_LIB=...] python tools/insert_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
import bench
from imfnet_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
pts, _ = bench.load_pair(1.7)
xyz = torch.as_tensor(np.concatenate(pts, 0)).to(dev)
n = xyz.shape[0]
cap = L.imf_hash_capacity(n)
table = torch.empty((cap, 2), dtype=torch.int64, device=dev)
ws = torch.empty(L.imf_unique_workspace_bytes(n), dtype=torch.uint8, device=dev)
coords = torch.empty((n, 4), dtype=torch.int32, device=dev); first = torch.empty(n, dtype=torch.int32, device=dev)
meta = torch.zeros(2, dtype=torch.int32, device=dev)
ts = []
for r in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.imf_voxelize(xyz.data_ptr(), 1, n, 0.025, 0, coords.data_ptr(), first.data_ptr(), meta.data_ptr(), table.data_ptr(), cap,
                   ws.data_ptr(), meta[1:].data_ptr(), torch.cuda.current_stream().cuda_stream)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("imf_voxelize (table reset + k_insert_points + k_flag_first + k_emit_unique), %d points: %.1f us" % (n, np.median(ts[2:])))
