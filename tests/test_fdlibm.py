"""crx_fdlibm.h (atanf / atan2f / tanf as the engine evaluates them on the GPU) against the host libm on a strided
sweep of the float domain; the full 2^32 sweep is `tests/tools/fdlibm_exhaustive.cpp` with stride 1 (0 mismatches,
53 s on 8 cores)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fdlibm_matches_host_libm(tmp_path):
    exe = str(tmp_path / "fde")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-o", exe,
                           os.path.join(HERE, "tools", "fdlibm_exhaustive.cpp")])
    out = subprocess.run([exe, "509"], capture_output=True, text=True)   # ~8.4 M points per function, prime stride
    assert out.returncode == 0, out.stdout
    assert "atanf mismatches 0" in out.stdout and "tanf mismatches 0" in out.stdout and "atan2f mismatches 0" in out.stdout
