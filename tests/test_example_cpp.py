"""examples/ekf_fleet.cpp: the reference's EKF demo for a whole fleet, in C++ against the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory, crx):
    out = str(tmp_path_factory.mktemp("ex") / "ekf_fleet")
    libdir = os.path.join(ROOT, "cpprobotics_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ekf_fleet.cpp"),
                           "-o", out, "-L", libdir, "-lcrx", f"-Wl,-rpath,{libdir}"])
    return out


def test_example_builds_and_fails_loudly_without_gpu(exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe, "64", "10"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_example_runs(exe):
    r = subprocess.run([exe, "4096", "200"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "EKF updates/s" in r.stdout
