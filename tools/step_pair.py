"""Pair step (the bench workload, one bucket, eager capacity mode) + per-family convolution times for the current IMF_CONV_VARIANT / IMF_LIB:
the A/B tool behind the kernel-policy decisions of round 5 (run it twice per arm on ONE box: boxes differ by up to 6 %)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, '.')
import bench, torch
from imfnet_amd import ops
dev = torch.device('cuda', 0)
pts2, imgs2 = bench.load_pair(1.7)
if os.environ.get('SINGLE'):          # SINGLE=1: one fragment per forward (the reference's call pattern)
    pts2, imgs2 = pts2[:1], imgs2[:1]
sync = torch.cuda.synchronize
with torch.no_grad():
    m0, _ = bench.build_model(dev)
    wl0 = bench.Workload(m0, dev, pts2, imgs2, 0.025)
    wl0.prepare_graph(replicate=True)
    wl0.runner.use_graph = False
    for _ in range(200):
        r0 = wl0.graph_step()
    sync()
    ts = sorted(bench.timed(wl0.graph_step, 30, sync) * 1e3 for _ in range(7))
    print(os.environ.get('IMF_LIB', 'default'), 'variant', ops.CONV_VARIANT, 'ms/step median %.4f min %.4f' % (ts[3], ts[0]))
    tr = []
    for _ in range(3):
        wl0.graph_step(tr)
    sync()
    agg = {}
    for rec in tr:
        g = agg.setdefault(rec['kernel'], [0, 0.0]); g[0] += 1; g[1] += rec['ev'].elapsed_ms()
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('   %-28s n/step %2d  avg %7.1f us  total/step %7.1f us' % (k, n // 3, ms * 1e3 / n, ms * 1e3 / 3))
