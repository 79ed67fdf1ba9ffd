"""In-kernel shader-clock stamps of the LDS-DMA conv kernel k_spconv_g (diagnostic build: make -C imfnet_amd/csrc stamps_g).
usage: python tools/conv_stamps_g.py [cin cout level [max_tiles]]      (split 1, unsplit launch of the k3 map of `level`)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["IMF_LIB"] = os.path.join(ROOT, "imfnet_amd", "libimfnet_hip_stamps_g.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes as C
import numpy as np, torch
from imfnet_amd import ops, _lib
from imfnet_amd import sparse as ME
from bench import load_workload, load_pair
cin, cout, lvl = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 64, 0)
max_tiles = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
pts, imgs = load_pair(1.7)
xyz, starts = np.concatenate(pts, 0), [0, len(pts[0])]
levels = ops.PyramidFuture(torch.as_tensor(xyz).to(dev), 0.025, 4, 0, item_starts=starts).result()
cm = ME.CoordinateManager.from_levels(levels)
rb = cm.conv_rulebook(1 << lvl, 3, 1)
g = torch.Generator().manual_seed(0)
f = torch.randn(levels[lvl].n, cin, generator=g).to(dev)
w = ops.pack_weights((torch.randn(27, cin, cout, generator=g) * 0.05).to(dev), split16=True)
out = torch.empty(levels[lvl].n, cout, device=dev)
n_tiles = rb.n_slots // 64
if max_tiles:
    n_tiles = min(n_tiles, max_tiles)
ws = torch.zeros(n_tiles * 4 * 256, dtype=torch.int64, device=dev)
a = _lib.ConvArgs()
a.in_a, a.c_a, a.c_b = f.data_ptr(), cin, 0
a.w_packed, a.kvol, a.cout = w.data_ptr(), 27, cout
a.tile_rows, a.nbr, a.tile_mask = ops._ptr(rb.tile_rows), rb.nbr.data_ptr(), rb.tile_mask.data_ptr()
a.n_slots, a.n_out = n_tiles * 64, min(rb.n_out, n_tiles * 64)
a.out, a.split_k, a.variant = out.data_ptr(), 1, 6
a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 8
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
st = ws.cpu().numpy().reshape(n_tiles, 4, 256).astype(np.float64)
n_sub = 27 * (cin // 32)
full = st[:, :, 8 + 4 * (n_sub - 1)] > 0                      # tiles with all 27 offsets active
s = st[full]
T = lambda i: s[:, i]
loop = np.stack([s[:, 8 + 4 * n: 12 + 4 * n] for n in range(n_sub)], 1)      # [waves, n_sub, 4]
med = lambda x: float(np.median(x))
print("cycles (shader clock), medians over %d wavefronts, %d sub-stages each:" % (len(s), n_sub))
print("  prologue (start -> loop)                 : %8.0f" % med(T(1) - T(0)))
print("  sub-stage: vmcnt wait + barrier          : %8.0f" % med(loop[:, :, 1] - loop[:, :, 0]))
print("  sub-stage: DMA issue + bookkeeping reads : %8.0f" % med(loop[:, :, 2] - loop[:, :, 1]))
print("  sub-stage: fragment reads + split        : %8.0f" % med(loop[:, :, 3] - loop[:, :, 2]))
print("  sub-stage: 12 MFMAs issued -> next top   : %8.0f" % med(loop[:, 1:, 0] - loop[:, :-1, 3]))
print("  sub-stage period                         : %8.0f" % med(loop[:, 1:, 0] - loop[:, :-1, 0]))
print("  whole loop per wavefront                 : %8.0f" % med(T(2) - T(1)))
print("  epilogue                                 : %8.0f" % med(T(3) - T(2)))
print("  wavefront lifetime                       : %8.0f   p95 %8.0f" % (med(T(3) - T(0)), float(np.percentile(T(3) - T(0), 95))))
