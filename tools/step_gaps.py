"""Pair step (one bucket, eager capacity mode): per traced launch its in-situ duration AND the gap to the next traced launch
(end event -> next begin event on the main stream): what the step spends BETWEEN its convolutions (dispatch of a dependent
kernel, event records, waits for the side streams' maps, the fusion between conv4's block and conv4_tr)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from imfnet_amd import _lib
dev = torch.device('cuda', 0)
pts2, imgs2 = bench.load_pair(1.7)
if os.environ.get('SINGLE'):
    pts2, imgs2 = pts2[:1], imgs2[:1]
sync = torch.cuda.synchronize
L = _lib.lib()
with torch.no_grad():
    m0, _ = bench.build_model(dev)
    wl = bench.Workload(m0, dev, pts2, imgs2, 0.025)
    wl.prepare_graph(replicate=True)
    wl.runner.use_graph = False
    for _ in range(150):
        wl.graph_step()
    sync()
    ts = sorted(bench.timed(wl.graph_step, 30, sync) * 1e3 for _ in range(7))
    print('step ms median %.4f min %.4f (untraced)' % (ts[3], ts[0]))
    N = 5
    steps = []
    for _ in range(N):
        tr = []
        wl.graph_step(tr)
        steps.append(tr)
    sync()
    names = [r['name'] for r in steps[0]]
    dur = [0.0] * len(names); gap = [0.0] * len(names); span = 0.0
    for tr in steps:
        for i, r in enumerate(tr):
            dur[i] += r['ev'].elapsed_ms()
            if i + 1 < len(tr):
                gap[i] += L.imf_event_elapsed_ms(r['ev'].end, tr[i + 1]['ev'].begin)
        span += L.imf_event_elapsed_ms(tr[0]['ev'].begin, tr[-1]['ev'].end)
    for i, n in enumerate(names):
        print('   %-18s %-24s %7.1f us   gap after %6.1f us' % (n, steps[0][i]['kernel'], dur[i] * 1e3 / N, gap[i] * 1e3 / N))
    print('   sum of launches %.1f us, sum of gaps %.1f us, first begin -> last end %.1f us (traced steps carry 2 event records per launch)'
          % (sum(dur) * 1e3 / N, sum(gap) * 1e3 / N, span * 1e3 / N))
