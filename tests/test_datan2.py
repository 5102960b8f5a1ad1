"""csrc/crx_datan2.h — the double atan2(L*k, 1.0) of the tracking controllers' feed-forward term (reference
src/lqr_speed_steer_control.cpp:143, src/lqr_steer_control.cpp:126) — against the host libm's atan2(), bit for bit.

CPU: the header compiled for the host: a prime-strided sweep of the float curvatures k at L = 0.5, 160 M random doubles of every
branch, every table row and branch boundary (tests/tools/datan2_exhaustive.cpp; stride 1 — ALL 2^32 curvatures, 0 mismatches for
six wheelbases — takes 20 s per wheelbase on 8 cores and is recorded in profiles/r04/datan2.txt).
GPU: the same header on the device: samples through the probe crx_x_datan2_dev against Python's math.atan2 (a libm call), and
ALL 2^32 curvatures through 4096 block checksums that the host tool forms from the libm (threads = the host's cores)."""
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FLAGS = ["-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fno-builtin-atan2", "-pthread"]


def _tool(tmp_path):
    exe = str(tmp_path / "dax")
    subprocess.check_call(["g++", *FLAGS, "-o", exe, os.path.join(HERE, "tools", "datan2_exhaustive.cpp"), "-lm"])
    return exe


def test_datan2_matches_host_libm(tmp_path):
    exe = _tool(tmp_path)
    for L, stride in (("0.5", "251"), ("2.5", "1009")):
        out = subprocess.run([exe, L, "4", stride, "random"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        assert " 0 mismatches" in out.stdout


def test_generated_table_is_the_host_libms(tmp_path):
    """scripts/gen/gen_datan2.py re-run against this host's libm reproduces the committed table (same glibc build)."""
    import shutil
    inc = os.path.join(HERE, "..", "cpprobotics_amd", "csrc", "crx_datan2_tab.inc")
    keep = open(inc).read()
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if not os.path.exists(libm):
        pytest.skip("no libm.so.6 at the expected path")
    try:
        subprocess.check_call(["python3", os.path.join(HERE, "..", "scripts", "gen", "gen_datan2.py"), libm], stdout=subprocess.DEVNULL)
        assert open(inc).read() == keep
    finally:
        open(inc, "w").write(keep)


@pytest.mark.gpu
def test_device_datan2_is_libm_bit_for_bit():
    import torch
    from cpprobotics_amd.experimental import datan2_one
    rng = np.random.default_rng(11)
    k = rng.integers(0, 2 ** 32, 600000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    general = rng.uniform(-1.0, 1.0, 300000) * np.exp2(rng.integers(-62, 62, 300000))
    edges = np.array([0.0, -0.0, 0.0625, np.nextafter(0.0625, 0), np.nextafter(0.0625, 1), 1.0, np.nextafter(1.0, 0), np.nextafter(1.0, 2),
                      16.0, np.nextafter(16.0, 0), np.nextafter(16.0, 17), 2.0 ** -57, 2.0 ** 57, np.nextafter(2.0 ** 57, 0), 5e-324, 1e308,
                      float("inf"), float("nan")])
    y = np.concatenate([0.5 * k.astype(np.float64), 2.5 * k.astype(np.float64), general, edges, -edges])
    got = datan2_one(torch.from_numpy(y).cuda()).cpu().numpy()
    want = np.array([math.atan2(v, 1.0) for v in y])
    bad = (got.view(np.uint64) != want.view(np.uint64)) & ~(np.isnan(got) & np.isnan(want))
    assert not bad.any(), (y[bad][:5], got[bad][:5], want[bad][:5])


@pytest.mark.gpu
def test_device_datan2_all_float_curvatures(tmp_path):
    """All 2^32 curvatures at the reference's wheelbase: 4096 block checksums of the device's results equal the host libm's."""
    from cpprobotics_amd.experimental import datan2_sweep
    exe = _tool(tmp_path)
    threads = str(max(1, min(64, len(os.sched_getaffinity(0)))))
    host = subprocess.run([exe, "0.5", threads, "sums"], capture_output=True, text=True, check=True).stdout.split()
    host = np.array([int(h, 16) for h in host], dtype=np.uint64)
    sums, diff, ks = datan2_sweep(0.5)
    dev = sums.cpu().numpy().view(np.uint64)
    assert host.shape == (4096,) and np.array_equal(dev, host), np.nonzero(dev != host)[0][:10]
    d = diff.cpu().numpy()
    print(f"OCML atan vs glibc-exact atan2 over 2^32 curvatures at L=0.5: {int(d[0])} differ in double, {int(d[1])} after rounding to float; "
          f"e.g. k = {[float(v).hex() for v in ks.cpu().numpy()[:8]]}")
