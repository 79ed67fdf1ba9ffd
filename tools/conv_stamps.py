"""In-kernel s_memtime stamps of the pipelined conv kernel (env IMF_ABLATE=0x40000000)."""
import os, sys
os.environ["IMF_ABLATE"] = str(0x40000000)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes as C
import numpy as np, torch
from imfnet_amd import ops, _lib
from imfnet_amd import sparse as ME
from bench import load_workload
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
cm = ME.CoordinateManager.from_levels(levels)
rb = cm.conv_rulebook(1, 3, 1)
g = torch.Generator().manual_seed(0)
f = torch.randn(levels[0].n, 64, generator=g).to(dev)
w = ops.pack_weights((torch.randn(27, 64, 64, generator=g) * 0.05).to(dev))
out = torch.empty(levels[0].n, 64, device=dev)
ws = torch.zeros(1024 * 4 * 32 * 4, dtype=torch.int64, device=dev)
a = _lib.ConvArgs()
a.in_a, a.c_a, a.c_b = f.data_ptr(), 64, 0
a.w_packed, a.kvol, a.cout = w.data_ptr(), 27, 64
a.tile_rows, a.nbr, a.tile_mask = rb.tile_rows.data_ptr(), rb.nbr.data_ptr(), rb.tile_mask.data_ptr()
a.n_slots, a.n_out = rb.n_slots, rb.n_out
a.out, a.split_k, a.variant = out.data_ptr(), 1, 0
a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 8
for _ in range(3):
    ws.zero_()
    _lib.check(_lib.lib().imf_spconv_fwd(C.byref(a), torch.cuda.current_stream().cuda_stream), "conv")
    torch.cuda.synchronize()
st = ws.cpu().numpy().reshape(1024, 4, 32, 4)[:801, :, :27, :].astype(np.float64)
t0 = st[..., 0]; t1 = st[..., 1]; t2 = st[..., 2]; t3 = st[..., 3]
print("per-stage medians over all waves/stages (s_memtime ticks; 100 MHz const clock => x10 ns? see below)")
print("  top->after barrier   :", np.median(t1 - t0))
print("  prefetch issue       :", np.median(t2 - t1))
print("  MFMA section         :", np.median(t3 - t2))
print("  stage period         :", np.median(t0[:, :, 1:] - t0[:, :, :-1]))
print("  whole loop per wave  :", np.median(t3[:, :, 26] - t0[:, :, 0]))
print("  kernel span (max end - min start):", t3.max() - t0.min())
starts = t0[:, 0, 0]
print("  block start spread p5/p50/p95:", np.percentile(starts - starts.min(), [5, 50, 95]))
