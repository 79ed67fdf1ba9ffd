#!/bin/bash
# experiment: ONE wavefront per half tile (IMF_W1=1 in the experimental library) vs four
L=$PWD/gpurun_ab/libimf_w1.so
for w1 in 0 1 0 1; do
  echo "== IMF_W1=$w1"; IMF_LIB=$L IMF_W1=$w1 IMF_SORTED_MAP=0 timeout 300 python tools/step_pair.py 2>&1 | grep -v amdgpu.ids | head -3
done
IMF_LIB=$L IMF_W1=1 python - <<'PY' 2>&1 | grep -v amdgpu
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from imfnet_amd import ops, sparse as ME
from bench import load_pair
dev = torch.device("cuda:0")
pts, imgs = load_pair(1.7)
xyz, starts = np.concatenate(pts, 0), [0, len(pts[0])]
levels = ops.PyramidFuture(torch.as_tensor(xyz).to(dev), 0.025, 4, 0, item_starts=starts).result()
cm = ME.CoordinateManager.from_levels(levels)
rb = cm.conv_rulebook(1, 3, 1)
g = torch.Generator().manual_seed(0)
fa = torch.randn(levels[0].n, 64, generator=g).to(dev)
w = ops.pack_weights((torch.randn(27, 64, 64, generator=g) * 0.05).to(dev), variant=3)
a = ops.spconv(fa, w, 64, rb, variant=3, split_k=1, staging="wave4h")      # W1 path (env)
b = ops.spconv(fa, w, 64, rb, variant=3, split_k=1, staging="wave4")       # 4 wavefronts, whole tiles
print("W1 vs W4 max rel diff", float((a - b).abs().max() / b.abs().max()))
PY
