"""The C-ABI shared library loads and exports every symbol include/crx.h declares (CPU-only checks)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "crx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(crx_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_and_loader_agree(crx):
    declared = _declared_functions()
    assert len(declared) >= 20
    assert sorted(crx.EXPORTED_SYMBOLS) == declared


def test_library_exports_every_declared_symbol(crx):
    raw = C.CDLL(crx.lib_path())
    for name in _declared_functions():
        assert hasattr(raw, name), f"libcrx.so does not export {name}"


def test_version_and_defaults(crx):
    from cpprobotics_amd import _lib as L
    l = crx.lib()
    assert l.crx_version() >= 100
    e = L.EkfParams(); l.crx_ekf_default_params(C.byref(e)); assert e.dt == 0.1
    q = L.LqrParams(); l.crx_lqr_default_params(C.byref(q))
    assert (q.dt, q.L, q.maxiter) == (0.1, 0.5, 150) and abs(q.eps - 0.01) < 1e-9
    m = L.MpcParams(); l.crx_mpc_default_params(C.byref(m))
    assert (m.dt, m.wb, m.max_accel, m.max_iter) == (0.2, 2.5, 1.0, 50)
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


def test_product_does_not_reference_the_oracle():
    """Nothing under cpprobotics_amd/ may import, include or link the oracle (crx_trig.h is the one
    file the oracle borrows FROM the product, not the other way round)."""
    for dp, _, files in os.walk(os.path.join(ROOT, "cpprobotics_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "oracle/" not in txt.replace(
                    "oracle/eigen_order.h", "").replace("oracle/mpc_ref.cpp", ""), f"{f} references the oracle"
