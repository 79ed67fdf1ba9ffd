"""Pair step (one bucket, eager capacity mode) and the time of EVERY traced launch by layer name, for the current environment
(e.g. IMF_SORTED_MAP=0/1, IMF_CONV_VARIANT): which layers move when a switch is flipped, and does the step follow?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
dev = torch.device('cuda', 0)
pts2, imgs2 = bench.load_pair(1.7)
sync = torch.cuda.synchronize
with torch.no_grad():
    m0, _ = bench.build_model(dev)
    wl = bench.Workload(m0, dev, pts2, imgs2, 0.025)
    wl.prepare_graph(replicate=True)
    wl.runner.use_graph = False
    for _ in range(150):
        wl.graph_step()
    sync()
    ts = sorted(bench.timed(wl.graph_step, 30, sync) * 1e3 for _ in range(7))
    print('step ms median %.4f min %.4f' % (ts[3], ts[0]), {k: v for k, v in os.environ.items() if k.startswith('IMF_')})
    tr = []
    for _ in range(5):
        wl.graph_step(tr)
    sync()
    agg = {}
    for rec in tr:
        g = agg.setdefault(rec['name'], [0, 0.0, rec['kernel']]); g[0] += 1; g[1] += rec['ev'].elapsed_ms()
    tot = 0.0
    for k, (n, ms, kern) in agg.items():
        tot += ms / n
        print('   %-18s %-24s %7.1f us' % (k, kern, ms * 1e3 / n))
    print('   sum of traced launches %.1f us' % (tot * 1e3))
