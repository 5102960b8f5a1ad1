"""tests/tools/eigen_order_probe.cpp: oracle/eigen_order.h — the restated accumulation order of Eigen's small fixed-size float products,
the one piece of the reference's arithmetic that lives in an un-vendored dependency (find_package(Eigen3 REQUIRED),
/root/reference/CMakeLists.txt:13) — against an EXECUTED Eigen, on whichever box has one.  Needs neither /root/reference nor a GPU; the
same test runs in the CPU session and (marked gpu) on the GPU box, and reports what it found as a warning so that it shows in the
session's summary either way."""
import os
import subprocess
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "tools", "eigen_order_probe.cpp")


def eigen_include_dir():
    """The directory holding Eigen/Eigen on this host, or None.  EIGEN3_INCLUDE_DIR overrides the usual places."""
    cands = [os.environ.get("EIGEN3_INCLUDE_DIR"), "/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/homebrew/include/eigen3",
             "/usr/include", "/usr/local/include"]
    for d in cands:
        if d and os.path.exists(os.path.join(d, "Eigen", "Eigen")) and os.path.exists(os.path.join(d, "Eigen", "src", "Core")):
            return d
    return None


def _run_probe(tmp_path, include_dir, seeds=300):
    exe = str(tmp_path / "eigen_order_probe")
    # no -march / -mfma: the reference's own build sets none (/root/reference/CMakeLists.txt:4-6)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", include_dir, "-I", ROOT, SRC, "-o", exe])
    r = subprocess.run([exe, str(seeds)], capture_output=True, text=True)
    return r.returncode, r.stdout


def _against_standin(tmp_path):
    rc, out = _run_probe(tmp_path, os.path.join(ROOT, "oracle", "ref_shim"))
    assert rc == 0 and "0 mismatching" in out, out[-2000:]


def _against_real_eigen(tmp_path):
    inc = eigen_include_dir()
    if inc is None:
        msg = ("eigen_order_probe: no <Eigen/Eigen> on this host (looked in EIGEN3_INCLUDE_DIR, /usr/include/eigen3, /usr/local/include/eigen3, "
               "/usr/include): oracle/eigen_order.h stays UNPINNED against an executed Eigen here")
        warnings.warn(msg)
        pytest.skip(msg)
    rc, out = _run_probe(tmp_path, inc)
    warnings.warn(" | ".join(out.strip().splitlines()[-2:]) + f"  [{inc}]")
    assert rc == 0, out[-4000:]


def test_probe_flags_agree_with_the_generic_rule(tmp_path):
    """Against the stand-in (which derives the order from the operand TYPES): the per-call-site transposed/row-major flags the oracle
    passes to oracle::mul are the ones the generic rule derives, for every product and chain of the hot path."""
    _against_standin(tmp_path)


def test_eigen_order_against_executed_eigen(tmp_path):
    _against_real_eigen(tmp_path)


@pytest.mark.gpu
def test_eigen_order_against_executed_eigen_on_the_gpu_box(tmp_path):
    """The same probe on the GPU box (it needs no GPU; the box is simply another host that may have Eigen)."""
    _against_standin(tmp_path)
    _against_real_eigen(tmp_path)
