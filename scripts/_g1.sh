set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04/pytest_gpu_1.txt
cat gpurun_out/r04/pytest_gpu_1.txt
timeout 300 python -m pytest tests/test_datan2.py tests/test_track_gpu.py -m gpu -q -s -k "datan2 or adversarial" 2>&1 | grep -E "OCML|adversarial|passed|failed" > gpurun_out/r04/datan2_gpu.txt
cat gpurun_out/r04/datan2_gpu.txt
timeout 600 python bench.py > gpurun_out/r04/bench_1.json 2> gpurun_out/r04/bench_1.err; tail -c 600 gpurun_out/r04/bench_1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
PY
