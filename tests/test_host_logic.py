"""Host logic of the host-pointer boundary that needs no GPU (csrc/crx_host.h): the copy-thread pool (random dense / strided jobs from
several submitters, tickets joined in either order) and the contiguous balanced partition of the device set; the device-selection
entry points refusing loudly without a device."""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_copy_pool_and_shard_partition(tmp_path):
    exe = str(tmp_path / "cpt")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-pthread", "-w", "-o", exe, os.path.join(HERE, "tools", "copy_pool_test.cpp")])
    out = subprocess.run([exe, "4", "60"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " 0 failures" in out.stdout, out.stdout + out.stderr


def test_device_entry_points_without_a_gpu(crx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    l = crx.lib()
    assert l.crx_get_device() == -1 and l.crx_set_device(0) == -2            # CRX_ERR_NO_DEVICE
    assert l.crx_set_devices(None, 2, 1) == -2 and l.crx_set_devices(None, 0, 0) == 0 and l.crx_get_devices(None, 0) == 0
    assert l.crx_set_devices(None, -1, 0) == -1
    assert not l.crx_host_alloc(64) and b"no HIP device" in l.crx_last_error()
    assert l.crx_release_workspace() == 0 and l.crx_shutdown() == 0 and l.crx_reserve_workspace(1 << 20, 1 << 20) == -2


def test_host_libm_is_the_one_the_kernels_restate(crx):
    """crx_host_libm_check: this host's libm (the oracle's sinf / cosf / expf / atan2f / tanf / acosf / sin / cos / atan2) against the
    engine's restatements on 200,000 pseudo-random arguments per family — a host-side regression test of every restated libm
    function at once, and the check a maintainer runs on a new host before trusting bit parity there."""
    assert crx.lib().crx_host_libm_check() == 0
