"""Host logic of the fused EKF kernel: the packed fast step (csrc/ekf_math.h: ekf_step_packed, the
fast sincos, the fast-domain flags and the general-step fallback) built for the CPU and compared
bit-for-bit with the oracle.  No GPU needed; the same cases run through the HIP kernel in
tests/test_ekf_gpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import bit_equal, ekf_QR, ekf_agents, ekf_noise

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "tools", "ekf_packed_host.cpp")


@pytest.fixture(scope="module")
def packed(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("packed") / "ekf_packed_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, SRC])
    lib = C.CDLL(so)

    def run(x0, P0, z, ud, Q, R, dt=0.1):
        T, n = z.shape[0], x0.shape[0]
        x, P = x0.copy(), P0.copy()
        xh = np.empty((T, n, 4), dtype=np.float32)
        slow = C.c_longlong(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib.ekf_packed_run(C.c_int(n), C.c_int(T), vp(x), vp(P), vp(np.ascontiguousarray(z)),
                                vp(np.ascontiguousarray(ud)), vp(xh), vp(Q), vp(R), C.c_double(dt), C.byref(slow))
        assert rc == 0
        return x, P, xh, slow.value
    return run


def _inputs(oracle, n, T, seed, x0_override=None, P_scale=None):
    u, x0, P0 = ekf_agents(n, seed)
    if x0_override is not None:
        x0 = x0_override(x0)
    if P_scale is not None:
        P0 = (P0 * P_scale).astype(np.float32)
    w = ekf_noise(T, n, seed + 1000)
    z, ud, _, _, _, _ = oracle.ekf_simulate_inputs(u, x0, x0, w)
    return x0, P0, z, ud


def test_packed_step_matches_oracle_common_path(packed, oracle_mod):
    Q, R = ekf_QR()
    x0, P0, z, ud = _inputs(oracle_mod, 257, 300, seed=11)
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    x, P, xh, slow = packed(x0, P0, z, ud, Q, R)
    assert slow == 0                      # the benchmark's regime never leaves the fast domain
    assert bit_equal(xh, xho) and bit_equal(x, xo) and bit_equal(P, Po)


def test_packed_step_tiny_and_zero_yaw(packed, oracle_mod):
    """|yaw| < 2^-12 (sinf returns its argument, cosf returns 1: reproduced by the fast polynomials) and
    yaw = +-0 (the reference's own start; outside the fast domain because of sinf(-0) = -0)."""
    Q, R = ekf_QR()
    def tiny(x0):
        x0 = x0.copy()
        x0[:, 2] = np.float32(1e-5) * np.linspace(-1, 1, x0.shape[0], dtype=np.float32)
        x0[0, 2] = 0.0
        x0[1, 2] = -0.0
        return x0
    x0, P0, z, ud = _inputs(oracle_mod, 64, 20, seed=12, x0_override=tiny)
    ud[:, :, 1] *= np.float32(1e-4)       # keep the yaw tiny for a few steps
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    x, P, xh, slow = packed(x0, P0, z, ud, Q, R)
    assert slow == 2                      # exactly the two vehicles that start at yaw = +0 / -0, in their first step
    assert bit_equal(xh, xho) and bit_equal(x, xo) and bit_equal(P, Po)
    assert np.array_equal(np.signbit(xh), np.signbit(xho))    # signed zeros too (sinf(-0) = -0)


@pytest.mark.parametrize("yaw", [119.99999, 120.0, 121.0, -500.0, 1.0e6, -3.0e9, 1.0e30])
def test_packed_step_large_yaw_falls_back(packed, oracle_mod, yaw):
    """|yaw| >= 120 leaves the fast sincos domain: the step must be redone by the general code."""
    Q, R = ekf_QR()
    def big(x0):
        x0 = x0.copy()
        x0[::3, 2] = np.float32(yaw)
        return x0
    x0, P0, z, ud = _inputs(oracle_mod, 96, 12, seed=13, x0_override=big)
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    x, P, xh, slow = packed(x0, P0, z, ud, Q, R)
    if abs(yaw) >= 121.0:
        assert slow > 0
    assert bit_equal(xh, xho) and bit_equal(x, xo) and bit_equal(P, Po)


@pytest.mark.parametrize("scale", [1e-12, 1e12])
def test_packed_step_extreme_determinant_falls_back(packed, oracle_mod, scale):
    """det(S) outside [2^-60, 2^60] leaves the fast reciprocal's domain."""
    Q, R = ekf_QR()
    Qs, Rs = (Q * np.float32(scale)).astype(np.float32), (R * np.float32(scale)).astype(np.float32)
    x0, P0, z, ud = _inputs(oracle_mod, 64, 10, seed=14, P_scale=scale)
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Qs, Rs)
    x, P, xh, slow = packed(x0, P0, z, ud, Qs, Rs)
    assert slow > 0 and np.isfinite(xho).all()
    assert bit_equal(xh, xho) and bit_equal(x, xo) and bit_equal(P, Po)


def test_packed_step_nonfinite_state_falls_back(packed, oracle_mod):
    Q, R = ekf_QR()
    def bad(x0):
        x0 = x0.copy()
        x0[1, 2] = np.inf
        x0[2, 2] = np.nan
        return x0
    x0, P0, z, ud = _inputs(oracle_mod, 8, 5, seed=15)
    x0 = bad(x0)
    xo, Po, xho, _ = oracle_mod.ekf_run(x0, P0, z, ud, Q, R)
    x, P, xh, slow = packed(x0, P0, z, ud, Q, R)
    assert slow > 0
    ok = [0, 3, 4, 5, 6, 7]
    assert bit_equal(xh[:, ok], xho[:, ok]) and bit_equal(P[ok], Po[ok])
    assert np.isnan(xh[:, 1:3]).all() and np.isnan(xho[:, 1:3]).all()


def test_dt_split_is_the_double_product_on_every_float(tmp_path):
    """csrc/ekf_math.h: dt_mul_split — fma(t, dt_hi, t * dt_lo) in fp32 instead of (float)(0.1 * (double)t) — walked over ALL 2^32 floats:
    the same bits for every finite |t| >= 2^-120, and every sine / cosine of a fast-domain angle (2^-100 <= |yaw| < 120) lies there
    (tests/tools/dt_split_exhaustive.cpp; ~10 s on 8 cores).  Reference lines: /root/reference/src/extended_kalman_filter.cpp:30-31,43,45."""
    import subprocess
    exe = str(tmp_path / "dts")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-pthread"]
    try:
        cpuflags = open("/proc/cpuinfo").read()
    except OSError:
        cpuflags = ""
    if " fma " in cpuflags or " fma\n" in cpuflags:
        flags.append("-mfma")               # hardware fmaf; without it libm's (correctly rounded, slow): then a stride keeps the test short
    subprocess.check_call(["g++", *flags, os.path.join(HERE, "tools", "dt_split_exhaustive.cpp"), "-o", exe])
    r = subprocess.run([exe, "1" if "-mfma" in flags else "257"], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 mismatching;" in r.stdout and " 0 with |sin| or |cos|" in r.stdout, r.stdout


def test_fast_sincos_gives_glibc_bits_on_every_float_of_its_domain(tmp_path):
    """csrc/ekf_math.h: sincos_fast2 (round 5: range reduction by one fma against 1.5 * 2^52, Horner polynomials, v_bitop3 quadrant logic —
    NOT glibc's operations) walked over every float with |y| < 128: the same sine and cosine bits as sincosf_ (= glibc's sinf / cosf on
    all 2^32 inputs, tests/test_trig.py) wherever it reports "inside the fast domain", and "outside" exactly for |y| >= 120 and
    |y| < 2^-100 (tests/tools/trig_fast_exhaustive.cpp; ~7 s on 8 cores).  Reference lines: /root/reference/src/extended_kalman_filter.cpp:30-31,43-45."""
    import subprocess
    exe = str(tmp_path / "tfe")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-pthread"]
    try:
        cpuflags = open("/proc/cpuinfo").read()
    except OSError:
        cpuflags = ""
    fma = " fma " in cpuflags or " fma\n" in cpuflags
    if fma:
        flags.append("-mfma")
    subprocess.check_call(["g++", *flags, os.path.join(HERE, "tools", "trig_fast_exhaustive.cpp"), "-o", exe])
    # without hardware fma the sweep goes through libm's fma(): a band of 2^24 patterns around the quadrant boundaries of |y| ~ 1..8 instead
    r = subprocess.run([exe] if fma else [exe, "0x40800000"], capture_output=True, text=True)
    print(r.stdout.strip())
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout
