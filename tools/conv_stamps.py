"""In-kernel s_memtime stamps of the split-f16 conv kernel (diagnostic build: make -C imfnet_amd/csrc stamps).
usage: python tools/conv_stamps.py [cin cout level [max_tiles]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["IMF_LIB"] = os.path.join(ROOT, "imfnet_amd", "libimfnet_hip_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes as C
import numpy as np, torch
from imfnet_amd import ops, _lib
from imfnet_amd import sparse as ME
from bench import load_workload
cin, cout, lvl = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 64, 0)
max_tiles = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
xyz, img, voxel = load_workload(1.7, 0.025)
levels = ops.pyramid_from_points(torch.as_tensor(xyz).to(dev), voxel, 4)
cm = ME.CoordinateManager.from_levels(levels)
rb = cm.conv_rulebook(1 << lvl, 3, 1)
g = torch.Generator().manual_seed(0)
f = torch.randn(levels[lvl].n, cin, generator=g).to(dev)
w = ops.pack_weights((torch.randn(27, cin, cout, generator=g) * 0.05).to(dev), split16=True)
out = torch.empty(levels[lvl].n, cout, device=dev)
n_tiles = rb.n_slots // 64
if max_tiles:
    n_tiles = min(n_tiles, max_tiles)          # launch only the first tiles: fewer workgroups per CU
ws = torch.zeros(n_tiles * 4 * 128, dtype=torch.int64, device=dev)
a = _lib.ConvArgs()
a.in_a, a.c_a, a.c_b = f.data_ptr(), cin, 0
a.w_packed, a.kvol, a.cout = w.data_ptr(), 27, cout
a.tile_rows, a.nbr, a.tile_mask = ops._ptr(rb.tile_rows), rb.nbr.data_ptr(), rb.tile_mask.data_ptr()
a.n_slots, a.n_out = n_tiles * 64, min(rb.n_out, n_tiles * 64)
a.out, a.split_k, a.variant = out.data_ptr(), 1, 6
a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 8
# NOTE: nbr is [K][n_slots] offset-major with the FULL n_slots stride; a truncated launch would read the
# wrong rows, so truncated runs rebuild a compact table
if max_tiles and n_tiles * 64 < rb.n_slots:
    nb = rb.nbr.view(27, rb.n_slots)[:, : n_tiles * 64].contiguous()
    a.nbr = nb.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ws.zero_()
    e0.record()
    _lib.check(_lib.lib().imf_spconv_fwd(C.byref(a), torch.cuda.current_stream().cuda_stream), "conv")
    e1.record()
    torch.cuda.synchronize()
print("kernel (events): %.1f us   tiles %d (%.2f per CU)" % (e0.elapsed_time(e1) * 1e3, n_tiles, n_tiles / 256))
st = ws.cpu().numpy().reshape(n_tiles, 4, 128).astype(np.float64)
ncc = cin // 32
KG = 2 if cout % 64 == 0 else 4
n_macro = (27 * ncc + KG - 1) // KG
full = st[:, :, 8 + 4 * (n_macro - 1)] > 0                      # tiles with all 27 offsets active
s = st[full]
T = lambda i: s[:, i]
loop = np.stack([s[:, 8 + 4 * n: 12 + 4 * n] for n in range(n_macro)], 1)      # [waves, n_macro, 4]
print("ticks (s_memtime = shader clock), medians over %d waves:" % len(s))
print("  prologue (start -> loop)    :", np.median(T(3) - T(0)))
print("  stage: LDS write + split    :", np.median(loop[:, :, 1] - loop[:, :, 0]))
print("  stage: barrier wait         :", np.median(loop[:, :, 2] - loop[:, :, 1]))
print("  stage: prefetch issue       :", np.median(loop[:, :, 3] - loop[:, :, 2]))
print("  stage: LDS reads + MFMAs    :", np.median(loop[:, 1:, 0] - loop[:, :-1, 3]))
print("  stage period                :", np.median(loop[:, 1:, 0] - loop[:, :-1, 0]))
print("  whole loop per wave         :", np.median(T(4) - T(3)))
print("  epilogue                    :", np.median(T(5) - T(4)))
print("  wave lifetime               :", np.median(T(5) - T(0)), " p95", np.percentile(T(5) - T(0), 95))
