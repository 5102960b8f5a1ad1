"""The C-ABI shared library loads and exports every symbol include/crx.h and include/crx_experimental.h declare (CPU-only checks)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header="crx.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(crx_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_and_loader_agree(crx):
    declared = _declared_functions()
    assert len(declared) >= 20
    assert sorted(crx.EXPORTED_SYMBOLS) == declared


def test_library_exports_every_declared_symbol(crx):
    raw = C.CDLL(crx.lib_path())
    for name in _declared_functions() + _declared_functions("crx_experimental.h"):
        assert hasattr(raw, name), f"libcrx.so does not export {name}"


def test_experimental_entry_points_are_separate(crx):
    """Measurement-only entry points: prefix crx_x_, declared in include/crx_experimental.h only, bound in
    cpprobotics_amd/experimental.py only; the product header and loader name none of them and no experiment is steered by an
    environment variable any more."""
    from cpprobotics_amd import experimental as X
    xs = _declared_functions("crx_experimental.h")
    assert xs and all(n.startswith("crx_x_") for n in xs) and sorted(X.EXPERIMENTAL_SYMBOLS) == xs
    assert not [n for n in _declared_functions() if n.startswith("crx_x_")]
    X.xlib()
    import glob
    csrc = os.path.join(ROOT, "cpprobotics_amd", "csrc")
    parts = [os.path.join(csrc, "crx_api.hip"), os.path.join(csrc, "crx_host.h")] + sorted(glob.glob(os.path.join(csrc, "api_*.inl")))
    assert len(parts) >= 10
    for f in parts:      # the one environment variable the library reads is the HIP runtime's own queue count, to REPORT it (crx_hw_queues)
        assert "getenv" not in open(f).read().replace('getenv("GPU_MAX_HW_QUEUES")', ""), f


def test_product_library_does_not_carry_the_rejected_variants(crx):
    """the measured-and-rejected kernel variants (two-lane EKF, four-lane MPC, lane-refilling MPC) are compiled into the A/B
    build libcrx_x.so only; the product libcrx.so neither contains their code objects nor runs them."""
    from cpprobotics_amd import experimental as X
    prod = open(crx.lib_path(), "rb").read()
    for k in (b"ekf_run_pair_kernel", b"mpc_quad_kernel", b"mpc_refill_kernel"):
        assert k not in prod, k
    ab = X.ab_lib_path()
    assert os.path.exists(ab), "libcrx_x.so missing: make -C cpprobotics_amd/csrc all"
    abb = open(ab, "rb").read()
    for k in (b"ekf_run_pair_kernel", b"mpc_quad_kernel", b"mpc_refill_kernel", b"ekf_run_kernel", b"mpc_kernel"):
        assert k in abb, k
    raw = C.CDLL(ab)
    for name in _declared_functions() + _declared_functions("crx_experimental.h"):
        assert hasattr(raw, name), f"libcrx_x.so does not export {name}"
    l = X.xlib()                                     # the product library answers the rejected variants with CRX_ERR_INVALID
    assert l.crx_x_ekf_run_pair_batch_dev(0, 0, None, None, None, None, None, None, None, None, None, None) == -1
    assert b"without the experimental kernels" in l.crx_last_error()


def test_version_and_defaults(crx):
    from cpprobotics_amd import _lib as L
    l = crx.lib()
    assert l.crx_version() >= 400
    e = L.EkfParams(); l.crx_ekf_default_params(C.byref(e)); assert e.dt == 0.1
    q = L.LqrParams(); l.crx_lqr_default_params(C.byref(q))
    assert (q.dt, q.L, q.maxiter) == (0.1, 0.5, 150) and abs(q.eps - 0.01) < 1e-9
    m = L.MpcParams(); l.crx_mpc_default_params(C.byref(m))
    assert (m.dt, m.wb, m.max_accel, m.max_iter, m.shared_gpu) == (0.2, 2.5, 1.0, 50, 0)
    import ctypes
    assert ctypes.sizeof(m) == 128                      # shared_gpu sits in what was padding behind max_iter
    assert abs(m.max_steer - np.pi / 4) < 1e-15 and abs(m.max_speed - 55 / 3.6) < 1e-12


def test_jacobH_constant(crx, oracle_mod):
    assert np.array_equal(crx.jacobH(), oracle_mod.jacobH())
    assert crx.jacobH().reshape(4, 2).T.tolist() == [[1, 0, 0, 0], [0, 1, 0, 0]]


def test_no_cpu_fallback(crx):
    """Without a GPU every compute entry point must fail loudly (status CRX_ERR_NO_DEVICE / CrxError)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    l = crx.lib()
    x = np.zeros((4, 4), np.float32); u = np.zeros((4, 2), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = l.crx_motion_model_batch(4, vp(x), vp(u), vp(x), None)
    assert rc == -2 and b"no HIP device" in l.crx_last_error()
    v = np.ones(4, np.float32); X = np.zeros((4, 25), np.float32)
    assert l.crx_dare_from_v_batch(4, 5, vp(v), None, vp(X), None, None) == -2
    with pytest.raises(crx.CrxError):
        crx.ekf_estimation(torch.zeros(4, 4), torch.zeros(4, 16), torch.zeros(4, 2), torch.zeros(4, 2), np.eye(4), np.eye(2))


def test_argument_validation(crx):
    l = crx.lib()
    assert l.crx_dare_from_v_batch(4, 3, None, None, None, None, None) == -1      # dim must be 4 or 5
    assert l.crx_ekf_step_batch(-1, None, None, None, None, None, None, None) == -1
    assert l.crx_jacobH(None) == -1
    assert len(l.crx_last_error()) > 0


def test_argument_validation_tracking_pf_dwa(crx):
    """The newer entry points validate before touching the device (status CRX_ERR_INVALID = -1, message set)."""
    from cpprobotics_amd import _lib as L
    l = crx.lib()
    x = np.zeros((4, 4), np.float32); f = np.zeros(4, np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    course = L.Course(4, vp(f), vp(f), vp(f), vp(f), vp(f))
    assert l.crx_lqr_steering_control_batch(4, 3, vp(x), C.byref(course), None, vp(f), vp(f), None, vp(x)) == -1       # dim
    bad = L.Course(0, vp(f), vp(f), vp(f), vp(f), vp(f))
    assert l.crx_lqr_steering_control_batch(4, 5, vp(x), C.byref(bad), None, vp(f), vp(f), None, vp(x)) == -1        # empty course
    assert l.crx_calc_nearest_index_batch(4, None, C.byref(course), None, None) == -1
    assert l.crx_update_batch(4, vp(x), None, vp(f), None) == -1
    lp = L.LoopParams(0, 0, 0.3, 1.0, 0.05, -1)
    assert l.crx_lqr_closed_loop_batch(4, 5, vp(x), C.byref(course), None, None, None, None, None, C.byref(lp), None, None) == -1   # max_ticks < 0
    assert l.crx_calc_ref_trajectory_batch(4, 0, vp(x), C.byref(course), 1.0, 0.2, 10, vp(f), vp(x)) == -1              # T < 1
    assert l.crx_pf_run_batch_dev(4, 77, 1, 4, vp(x), vp(x), vp(x), vp(x), None, None, None, None, None, None, None, None, None) == -1   # NP
    assert l.crx_dwa_run_batch_dev(4, 1, vp(x), vp(x), vp(x), vp(x), 1000, None, None, None, None, None, None, None) == -1   # too many obstacles
    assert b"bad argument" in l.crx_last_error()
    v = L.VehicleParams(); l.crx_vehicle_default_params(C.byref(v), 1)
    assert (v.dt, v.wheelbase, v.clamp_speed) == (0.2, 2.5, 1) and abs(v.max_steer - np.pi / 4) < 1e-15
    d = L.DwaConfig(); l.crx_dwa_default_config(C.byref(d))
    assert abs(d.max_yawrate - 40.0 * 3.141592653 / 180.0) < 1e-7 and abs(d.yawrate_reso - 0.1 * 3.141592653 / 180.0) < 1e-9
    q = L.PfParams(); l.crx_pf_default_params(C.byref(q))
    assert abs(q.Q - 0.01) < 1e-9 and q.dt == 0.1 and q.rsim0 == 1.0


def test_product_does_not_reference_the_oracle():
    """Nothing under cpprobotics_amd/ may import, include or link the oracle (crx_trig.h is the one
    file the oracle borrows FROM the product, not the other way round)."""
    for dp, _, files in os.walk(os.path.join(ROOT, "cpprobotics_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "oracle/" not in txt.replace(
                    "oracle/eigen_order.h", "").replace("oracle/mpc_ref.cpp", "").replace("oracle/eigen_qr.h", ""), f"{f} references the oracle"
                # comments may name those three oracle files; no line may include or import anything of the oracle
                assert not [l for l in txt.splitlines() if ("#include" in l or l.strip().startswith(("import ", "from "))) and "oracle" in l], f


@pytest.mark.parametrize("n", [-1, 1])
def test_every_batch_entry_point_refuses_null_arguments(n):
    """No GPU needed: argument validation comes before anything touches a device.  Every entry point whose first argument is the batch
    size, product and measurement-only alike, called with that size and NULL / zero for everything else, reports an error
    (CRX_ERR_INVALID, or CRX_ERR_NO_DEVICE on a box without a GPU) — it neither accepts the call nor dereferences a NULL."""
    from cpprobotics_amd import _lib as L
    from cpprobotics_amd import experimental as X
    lib = C.CDLL(L.lib_path())
    sigs = dict(L._SIGNATURES); sigs.update(X._X_SIGNATURES)
    called = 0
    for name, (res, args) in sorted(sigs.items()):
        if not args or args[0] is not C.c_int or res is not C.c_int or name == "crx_set_device":
            continue
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
        vals = [(n if i == 0 else 0) if a is C.c_int else (0.0 if a in (C.c_float, C.c_double) else (None if a is C.c_void_p or hasattr(a, "contents") else 0))
                for i, a in enumerate(args)]
        rc = fn(*vals)
        assert rc in (-1, -2), (name, rc)
        called += 1
    assert called >= 45



def test_libraries_bind_their_own_definitions(crx):
    """libcrx.so and libcrx_x.so define the same inline launch functions and kernel stubs with different bodies; both are loaded into
    one process by the A/B scripts.  Linked with -Wl,-Bsymbolic neither has a dynamic relocation against a crx symbol of its own, so
    the second library cannot end up calling the first one's definitions (round 5: an A/B that measured one kernel three times)."""
    import shutil
    import subprocess
    from cpprobotics_amd import experimental as X
    if not shutil.which("readelf"):
        pytest.skip("readelf not available")
    for path in (crx.lib_path(), X.ab_lib_path()):
        rel = subprocess.run(["readelf", "-r", "--wide", path], capture_output=True, text=True, check=True).stdout
        bad = [l for l in rel.splitlines() if ("JUMP_SLO" in l or "GLOB_DAT" in l) and ("crx" in l.split()[-1] or "_ZN3crx" in l)]
        assert not bad, (path, bad[:5])
