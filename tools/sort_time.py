#!/usr/bin/env python3
"""Device time of imf_rulebook_sort_by_occupancy's two launches per level of the pair (run under rocprofv3 --kernel-trace --stats:
tools/gpu.sh ... 'cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/st -- python $R/tools/sort_time.py'), and by HIP events here."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from imfnet_amd import ops, sparse as ME, _lib
from bench import load_pair
import ctypes as C
dev = torch.device("cuda:0")
pts, imgs = load_pair(1.7)
xyz, starts = np.concatenate(pts, 0), [0, len(pts[0])]
levels = ops.PyramidFuture(torch.as_tensor(xyz).to(dev), 0.025, 4, 0, item_starts=starts).result()
cm = ME.CoordinateManager.from_levels(levels)
L = _lib.lib()
for li in range(4):
    rb = cm.conv_rulebook(1 << li, 3, 1)
    tile_rows = torch.empty(rb.n_slots, dtype=torch.int32, device=dev)
    nbr = torch.empty(rb.kvol * rb.n_slots, dtype=torch.int32, device=dev)
    mask = torch.empty(rb.n_slots // 64 * 4, dtype=torch.int32, device=dev)
    ws = torch.empty(L.imf_rulebook_sorted_workspace_bytes(rb.n_slots), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: L.imf_rulebook_sort_by_occupancy(rb.nbr.data_ptr(), rb.kvol, rb.n_slots, rb.n_out, None, tile_rows.data_ptr(),
                                                    nbr.data_ptr(), mask.data_ptr(), ws.data_ptr(), ws.numel(), st)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record(); torch.cuda.synchronize()
    print("level %d: %6d slots (%d windows): %.1f us per sort (both launches, back to back)" % (li, rb.n_slots, (rb.n_slots + 16383) // 16384, e0.elapsed_time(e1) / 20 * 1e3))
