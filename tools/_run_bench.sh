cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fusion or forward_matches" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_harness.py -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/r4/bench.json 2> gpurun_out/r4/bench.err; echo "bench rc=$?"; tail -5 gpurun_out/r4/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype')})
print('host_span', json.dumps(d['host_span'])[:1800])
c=d['config']
for k in ('single_fragment','batch_4','batch_8','e2e_extract_features','host_span_f32_valued','graph_replay','strict_fp32','sharded_pipeline'):
    print(k, json.dumps(c.get(k))[:600])
r=d['roofline']; print({k:v for k,v in r.items() if k not in('per_kernel','traffic_note','timing','rocprof_note')})
print(json.dumps(r['per_kernel'])[:1500])
print(d['cpu_baseline'])
PY
