"""batched step time (NB fragments per forward) for the current env"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, '.')
import bench, torch, numpy as np
dev = torch.device('cuda', 0)
sync = torch.cuda.synchronize
nb = int(sys.argv[1])
pts2, imgs2 = bench.load_pair(1.7)
with torch.no_grad():
    m0, _ = bench.build_model(dev)
    wl = bench.Workload(m0, dev, pts2 * (nb // 2), np.concatenate([imgs2] * (nb // 2), 0), 0.025)
    wl.prepare_graph()
    wl.runner.use_graph = False
    for _ in range(3):
        r = wl.graph_step()
    sync()
    if r.flags:
        wl.runner.observe_batch(nb, r.bbox); wl.prepare_graph()
    for _ in range(40):
        wl.graph_step()
    sync()
    ts = sorted(bench.timed(wl.graph_step, 12, sync) * 1e3 for _ in range(5))
    print('batch %d ms/step median %.4f' % (nb, ts[2]), {k: os.environ.get(k) for k in ('IMF_L2_TAG', 'IMF_L3_TAG')})
