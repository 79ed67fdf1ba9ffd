"""single-fragment step time for the current env"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, '.')
import bench, torch
dev = torch.device('cuda', 0)
sync = torch.cuda.synchronize
xyz1, img1, voxel = bench.load_workload(1.7, 0.025)
with torch.no_grad():
    m0, _ = bench.build_model(dev)
    wl1 = bench.Workload(m0, dev, [xyz1], img1, voxel)
    wl1.prepare_graph()
    wl1.runner.use_graph = False
    for _ in range(100):
        wl1.graph_step()
    sync()
    ts = sorted(bench.timed(wl1.graph_step, 30, sync) * 1e3 for _ in range(7))
    print('single fragment ms/step median %.4f min %.4f' % (ts[3], ts[0]), {k: os.environ.get(k) for k in ('IMF_L1_TAG', 'IMF_L2_TAG', 'IMF_L3_TAG')})
