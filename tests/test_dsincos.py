"""csrc/crx_dsincos.h — the double sin / cos of the Frenet planner's normal (reference
src/frenet_optimal_trajectory.cpp:111-112: std::cos(iyaw + M_PI/2.0), std::sin(iyaw + M_PI/2.0) with a float iyaw) — against the
host libm's sin() / cos(), bit for bit.

CPU: the header compiled for the host, a prime-strided sweep of every argument the planner can form ((double)f + M_PI/2 for the
floats f in [-pi, pi]) plus 20 M random doubles of every branch up to 1e8 (tests/tools/dsincos_exhaustive.cpp; stride 1 — all
2.16e9 arguments, 0 mismatches — takes 40 s on 8 cores: `for p in 0..7: dsincos_exhaustive $p 8`).
GPU: the same header on the device through the probe entry crx_x_dsincos_dev against Python's math.sin / math.cos (libm calls)."""
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_dsincos_matches_host_libm(tmp_path):
    exe = str(tmp_path / "dsx")
    # the builtins stay separate libm calls (an optimising GCC would merge them into sincos(), which rounds differently)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fno-builtin-sin", "-fno-builtin-cos", "-o", exe,
                           os.path.join(HERE, "tools", "dsincos_exhaustive.cpp"), "-lm"])
    out = subprocess.run([exe, "0", "1", "127", "random"], capture_output=True, text=True)   # 17 M grid points x 2 signs + 20 M random
    assert out.returncode == 0, out.stdout
    assert " 0 mismatches" in out.stdout


def _planner_arguments(count, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 0x40490fdb + 64, count, dtype=np.uint64).astype(np.uint32)
    f |= (rng.integers(0, 2, count, dtype=np.uint64).astype(np.uint32) << np.uint32(31))
    return f.view(np.float32).astype(np.float64) + math.pi / 2.0


@pytest.mark.gpu
def test_device_dsincos_is_libm_bit_for_bit():
    import torch
    from cpprobotics_amd.experimental import dsincos
    rng = np.random.default_rng(5)
    general = rng.uniform(-1.0, 1.0, 200000) * np.exp2(rng.integers(-30, 26, 200000))   # every branch, |x| < 6.8e7
    edges = np.array([0.0, -0.0, 2.0 ** -27, 2.0 ** -26, 0.126, 0.855468, 0.855469, 0.8554688, 2.426264, 2.426265, 2.4262657, math.pi / 2,
                      math.pi, -math.pi / 2, 3 * math.pi / 2, 105414335.0, 1e9, float("inf"), float("nan")])
    x = np.concatenate([_planner_arguments(600000, 3), general, edges])
    s, c = dsincos(torch.from_numpy(x).cuda())
    s, c = s.cpu().numpy(), c.cpu().numpy()
    want_s = np.array([math.sin(v) if abs(v) < 105414336.0 else float("nan") for v in x])
    want_c = np.array([math.cos(v) if abs(v) < 105414336.0 else float("nan") for v in x])
    bad_s = (s.view(np.uint64) != want_s.view(np.uint64)) & ~(np.isnan(s) & np.isnan(want_s))
    bad_c = (c.view(np.uint64) != want_c.view(np.uint64)) & ~(np.isnan(c) & np.isnan(want_c))
    assert not bad_s.any(), (x[bad_s][:5], s[bad_s][:5], want_s[bad_s][:5])
    assert not bad_c.any(), (x[bad_c][:5], c[bad_c][:5], want_c[bad_c][:5])
