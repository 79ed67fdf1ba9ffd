cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_range.py tests/test_gpu_bench_sharded.py -x -q 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "native_executor or fusion_range" 2>&1 | tail -5
timeout 900 python bench.py --steps 10 --repeats 3 --no-cpu-baseline > gpurun_out/r5/bench_dbg.json 2> gpurun_out/r5/bench_dbg.err; echo "bench rc=$?"; tail -8 gpurun_out/r5/bench_dbg.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_dbg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype')})
r=d['roofline']; print({k:v for k,v in r.items() if k not in('per_kernel','traffic_note','timing','rocprof_note','peak_note','achieved_note','step_frac_note','notional_hbm')}); print(r['notional_hbm'])
print(json.dumps(r['per_kernel'])[:1200])
for k,v in d['arithmetics'].items():
    print(k, v['value'], v['ms_per_step'], v['max_abs_diff_vs_headline_descriptors'], {kk:vv for kk,vv in v['roofline'].items() if kk in ('kernel','achieved','peak','frac','avg_launch_us','step_frac')})
print('sharded', json.dumps(d['config']['sharded_pipeline'])[:1500])
print('host_span', d['host_span']['value'], d['host_span']['ms_per_step'])
c=d['config']
for k in ('single_fragment','batch_4','batch_8','e2e_extract_features','graph_replay'):
    print(k, json.dumps(c.get(k))[:300])
PY
