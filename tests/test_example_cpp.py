"""examples/*.cpp: the reference's EKF demo, its two sampling planners and its two tracking demos for a whole fleet, in C++
against the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path_factory, name, compiler="/opt/rocm/bin/hipcc"):
    out = str(tmp_path_factory.mktemp("ex") / name)
    libdir = os.path.join(ROOT, "cpprobotics_amd")
    subprocess.check_call([compiler, "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".cpp"), "-o", out, "-L", libdir, "-lcrx", f"-Wl,-rpath,{libdir}"])
    return out


@pytest.fixture(scope="module")
def host_exe(tmp_path_factory, crx):
    return _build(tmp_path_factory, "ekf_fleet_host", compiler="g++")        # plain C++: no HIP header, no hipcc


@pytest.fixture(scope="module")
def exe(tmp_path_factory, crx):
    return _build(tmp_path_factory, "ekf_fleet")


@pytest.fixture(scope="module")
def planner_exe(tmp_path_factory, crx):
    return _build(tmp_path_factory, "planner_fleet")


@pytest.fixture(scope="module")
def tracking_exe(tmp_path_factory, crx):
    return _build(tmp_path_factory, "tracking_fleet")


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
    assert "EKF updates/s" in r.stdout and "GPU(s)" in r.stdout


def test_host_example_builds_with_a_plain_cxx_compiler_and_fails_loudly_without_gpu(host_exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([host_exe, "64", "10"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", ["0", "1"])
def test_host_example_runs(host_exe, pinned):
    """The fleet through the host-pointer entry point (std::vector or pinned arrays), sharded over every visible GPU by
    crx_set_devices: the estimate tracks the truth."""
    import re
    r = subprocess.run([host_exe, "20000", "120", "99", pinned], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GB/s across the boundary" in r.stdout
    # the same fleet split into three shards (device 0 named three times: the code path of a 3-GPU C++ host): the same trajectory, bit for bit
    r3 = subprocess.run([host_exe, "20000", "120", "99", pinned, "3"], capture_output=True, text=True)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    c1, c3 = (re.search(r"trajectory checksum ([0-9a-f]{16})", o.stdout) for o in (r, r3))
    assert c1 and c3 and c1.group(1) == c3.group(1), (r.stdout, r3.stdout)


def test_planner_example_builds_and_fails_loudly_without_gpu(planner_exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([planner_exe, "64"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_planner_example_runs(planner_exe):
    r = subprocess.run([planner_exe, "512"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Frenet:" in r.stdout and "DWA:" in r.stdout


def test_tracking_example_builds_and_fails_loudly_without_gpu(tracking_exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([tracking_exe, "64"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_tracking_example_runs(tracking_exe):
    """The two tracking mains for a fleet: agent 0 is the reference's own vehicle — it reaches the LQR goal, and advances along the
    MPC course."""
    import re
    r = subprocess.run([tracking_exe, "512"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"LQR: .* reached the goal after (\d+) ticks; (\d+) of 512 reached it", r.stdout)
    assert m and 100 < int(m.group(1)) < 1000 and int(m.group(2)) > 450, r.stdout
    m = re.search(r"MPC: .* agent 0 is at course index (\d+)", r.stdout)
    assert m and int(m.group(1)) > 30, r.stdout


@pytest.fixture(scope="module")
def mgpu_exe(tmp_path_factory, crx):
    return _build(tmp_path_factory, "ekf_fleet_mgpu")


def test_mgpu_example_builds_and_fails_loudly_without_gpu(mgpu_exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([mgpu_exe, "0", "1", str(tmp_path / "id")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_mgpu_example_runs_as_a_one_rank_fleet(mgpu_exe, tmp_path):
    """examples/ekf_fleet_mgpu.cpp: one process per GPU, the final estimates concatenated by crx_allgather_dev (RCCL behind the C ABI) —
    here with world = 1, the only world a one-GPU box offers; the N > 1 launch is the shell loop in the file's header."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([mgpu_exe, "0", "1", str(tmp_path / "id"), "8192", "200"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank 0 of 1" in r.stdout and "gathered 8192 final estimates" in r.stdout
