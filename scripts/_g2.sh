set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_host_boundary_gpu.py tests/test_example_cpp.py tests/test_lqr_gpu.py -m gpu -q -s --timeout 600 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r04/pytest_host_2.txt
cat gpurun_out/r04/pytest_host_2.txt
timeout 900 python bench.py > gpurun_out/r04/bench_2.json 2> gpurun_out/r04/bench_2.err; tail -c 1500 gpurun_out/r04/bench_2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
e=d.get('extra',{})
print(json.dumps(e.get('ekf_host_boundary'),indent=1))
print(json.dumps(e.get('single_vehicle_call_latency'),indent=1))
print(json.dumps(d.get('roofline_dare_dense'),indent=1))
print(json.dumps(e.get('dare5_throughput_regime'),indent=1)); print(json.dumps(e.get('mpc_T21_throughput_regime'),indent=1)); print(e.get('throughput_regime_error'))
print(json.dumps(d.get('parity_secondary'),indent=1))
PY
