cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "spconv_matches_oracle or chip_filling or bf16x3" 2>&1 | tail -8
timeout 600 python - <<'PY' 2>&1 | tail -40
import sys, time, json
sys.path.insert(0, '.')
import bench, torch, numpy as np
dev = torch.device('cuda', 0)
pts2, imgs2 = bench.load_pair(1.7)
sync = torch.cuda.synchronize
res = {}
with torch.no_grad():
    for variant in (0, 3, 6):
        m0, _ = bench.build_model(dev, variant=variant)
        from imfnet_amd import ops
        prev = ops.CONV_VARIANT; ops.CONV_VARIANT = variant
        try:
            wl0 = bench.Workload(m0, dev, pts2, imgs2, 0.025)
            F0 = wl0.prepare_graph().clone()
            wl0.runner.use_graph = False
            for _ in range(5):
                r0 = wl0.graph_step()
            sync()
            print('variant', variant, 'runner.variant', wl0.runner.variant, 'flags', r0.flags, 'equal exact', bool(torch.equal(r0.F, F0)))
            ts = [bench.timed(wl0.graph_step, 20, sync) * 1e3 for _ in range(5)]
            print('variant', variant, 'ms/step', [round(t, 4) for t in ts])
            tr = []
            for _ in range(3):
                wl0.graph_step(tr)
            sync()
            agg = {}
            for rec in tr:
                g = agg.setdefault(rec['kernel'], [0, 0.0]); g[0] += 1; g[1] += rec['ev'].elapsed_ms()
            for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print('   %-28s n/step %2d  avg %7.1f us  total/step %7.1f us' % (k, n // 3, ms * 1e3 / n, ms * 1e3 / 3))
            res[variant] = r0.F.clone()
        finally:
            ops.CONV_VARIANT = prev
    print('max |F3 - F0| =', float((res[3] - res[0]).abs().max()), ' max |F6 - F0| =', float((res[6] - res[0]).abs().max()))
PY
