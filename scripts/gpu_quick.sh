# quick perf probe: EKF parity tests + bench line (no cpu baseline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ekf_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('value %.2f G/s  kernel_ms %.4f  frac %.4f'%(r['value']/1e9, r['roofline']['kernel_ms'], r['roofline']['frac']))"; done
