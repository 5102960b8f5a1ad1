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


def test_swarm_and_comm_entry_points_check_their_arguments_without_a_gpu(crx):
    """crx_swarm_* / crx_comm_* (include/crx.h, csrc/api_swarm.inl): defaults, argument checks and the no-device answer need no GPU."""
    import subprocess
    import sys
    import torch
    from cpprobotics_amd import _lib as L
    l = crx.lib()
    cfg = L.SwarmConfig()
    l.crx_swarm_default_config(C.byref(cfg))
    assert (cfg.T, cfg.Tm, cfg.plan_every, cfg.depth, cfg.nsearch) == (100, 21, 8, 6, 10) and abs(cfg.v_cmd - 2.5) < 1e-7 and cfg.dt_ref == 0.2
    assert cfg.mpc.max_iter == 50 and cfg.ekf.dt == 0.1 and not cfg.planner_streams
    h = C.c_void_p()
    assert l.crx_swarm_create(C.byref(h), None, None, None, None, None, None) == -1 and b"null argument" in l.crx_last_error()
    assert l.crx_swarm_round_dev(None, None, None, None, None, None) == -1
    assert l.crx_swarm_plans(None, 0, None, None, None, None, None, None) == -1 and l.crx_swarm_wait(None, None) == -1
    assert l.crx_swarm_destroy(None) == 0 and l.crx_swarm_state(None) is None
    assert l.crx_comm_unique_id(None) == -1 and l.crx_comm_destroy(None) == 0 and l.crx_comm_rank(None) == -1 and l.crx_comm_world(None) == 0
    assert l.crx_comm_init_rank(C.byref(h), b"\0" * 128, 2, 2) == -1            # rank outside [0, world)
    assert l.crx_allgather_dev(None, None, None, 16, None) == -1
    if not torch.cuda.is_available():
        # a well-formed configuration without a device: CRX_ERR_NO_DEVICE, no CPU fallback
        cfg.n = 64
        course = L.Course(n=4, cx=1, cy=1, cyaw=1, ck=1, sp=1)
        q = (C.c_float * 16)(); r = (C.c_float * 4)()
        assert l.crx_swarm_create(C.byref(h), C.byref(cfg), C.byref(course), C.c_void_p(16), C.c_void_p(16), q, r) == -2
        assert l.crx_comm_init_rank(C.byref(h), b"\0" * 128, 0, 1) == -2
    # crx_hw_queues reports the HIP runtime's variable (default 4; garbage -> 4): in processes of their own
    code = "import sys; sys.path.insert(0, sys.argv[1]); import cpprobotics_amd as c; print(c.lib().crx_hw_queues())"
    root = os.path.dirname(HERE)
    for val, want in ((None, 4), ("16", 16), ("abc", 4), ("0", 4)):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if val is not None:
            env["GPU_MAX_HW_QUEUES"] = val
        out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == str(want), (val, out.stdout, out.stderr[-500:])
