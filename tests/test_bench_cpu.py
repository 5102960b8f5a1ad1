"""bench.py / __graft_entry__ on a box without a GPU: they must refuse loudly (no CPU fallback of the product path), and the pieces of
bench.py that are plain host logic behave."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_bench_refuses_without_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env={**os.environ, "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
    assert "{" not in r.stdout                      # no JSON line, no number


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_smoke_refuses_without_gpu():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and ("no HIP device" in r.stderr or "GPU" in r.stderr)


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present")
def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus N` invoked plainly must not die on an assert: without WORLD_SIZE it starts the ranks
    itself.  On a GPU-less box that shows as (a) a clear refusal when fewer than N devices are visible, (b) with --oversubscribe, N
    ranks started under torch.distributed.run, each refusing for lack of a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "only 0 GPU(s) visible" in (r.stderr + r.stdout) and "--oversubscribe" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--oversubscribe"],
                       capture_output=True, text=True, env=env, timeout=600)
    out = r.stderr + r.stdout
    # both ranks ran bench.py's main(): each says so — unless the launcher, seeing the first rank fail, has already terminated the
    # other (SIGTERM), which its report then lists as a second local rank
    assert r.returncode != 0 and (out.count("needs a GPU") >= 2 or ("needs a GPU" in out and "local_rank: 1" in out)), out[-2000:]
    assert "{\"metric\"" not in r.stdout


def test_bench_host_helpers():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    c = b.usable_cores()
    assert 1 <= c <= (os.cpu_count() or 1)
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = b.parse()
        assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0 and a.vehicles == 65536 and a.T == 1000     # BASELINE configs[1] by default
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"]
        a = b.parse()
        assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)
    finally:
        sys.argv = old


def test_bench_builds_the_reference_matrices_itself(oracle_mod):
    """bench.py's timed DARE legs build lqr_steering_control's A, B, Q, R without the oracle (only its cpu_baseline leg may touch
    oracle/): the helper equals the oracle's builder bit for bit."""
    import numpy as np
    from common import lqr_speeds
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    v = lqr_speeds(2000, 3)
    for got, want in zip(b.lqr_pattern_mats(v), oracle_mod.lqr_build(v, 5)):
        assert np.array_equal(got.view(np.uint32), np.asarray(want, dtype=np.float32).reshape(got.shape).view(np.uint32))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02", "bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] == "GB/s"
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1 and "workload" in d["config"]
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["higher_is_better"] is True


def test_committed_counters_are_bound_to_the_kernel_code(crx):
    """profiles/traffic.json, side_counters.json and mpc_traffic.json carry the hash of the kernels' instruction streams they were taken
    from (cpprobotics_amd/_lib.py: kernel_code_hash); bench.py prints them only when the loaded libcrx.so has the same code.  Here: the
    hash is readable on the build box, stable, sensitive to the kernel family — and a stale file is reported (a warning, not a failure:
    the counters can only be re-taken on a GPU box)."""
    import json
    import warnings
    from cpprobotics_amd._lib import kernel_code_hash
    h = {f: kernel_code_hash(f) for f in ("ekf", "side", "mpc")}
    if not any(h.values()):
        pytest.skip("the code object cannot be disassembled on this host (no ROCm LLVM tools): kernel_code_hash is None, bench.py prints "
                    "the committed counters as null")
    assert all(v and len(v) == 16 for v in h.values()) and len(set(h.values())) == 3
    assert h == {f: kernel_code_hash(f) for f in h}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, fam in (("traffic.json", "ekf"), ("side_counters.json", "side"), ("mpc_traffic.json", "mpc")):
        d = json.load(open(os.path.join(root, "profiles", name)))
        assert "kernel_code_hash" in d, name
        if d["kernel_code_hash"] != h[fam]:
            warnings.warn(f"profiles/{name} was taken from kernel code {d['kernel_code_hash']}, this build has {h[fam]}: bench.py will print "
                          "its counters as null until scripts/gpu_prof.sh is re-run")


def test_code_hash_ignores_layout_not_code():
    """The hash behind the committed counters must survive a change ELSEWHERE in the library (addresses, encodings, the pc-relative
    distance to a constant table all move) and must not survive a change to the kernel's own instructions."""
    from cpprobotics_amd._lib import _functions_of_disassembly
    a = ("0000000000001000 <_Z3ekfv>:\n"
         "\ts_getpc_b64 s[8:9]                       // 000000001000: BE881C00\n"
         "\ts_add_u32 s8, s8, 0x153470                // 000000001004: 8008FF08 00153470\n"
         "\ts_addc_u32 s9, s9, 0                      // 00000000100C: 82098009\n"
         "\tv_fma_f32 v0, v1, v2, v3                  // 000000001010: D1CB0000 040E0501\n")
    moved = a.replace("0x153470", "0x14f99c").replace("000000001", "000000007").replace("0000000000001000", "0000000000007000")
    other = a.replace("v_fma_f32 v0, v1, v2, v3", "v_fma_f32 v0, v1, v2, v4")
    assert _functions_of_disassembly(a) == _functions_of_disassembly(moved)
    assert _functions_of_disassembly(a) != _functions_of_disassembly(other)
    assert _functions_of_disassembly(a)["_Z3ekfv"][1] == "s_add_u32 s8, s8, <pcrel>"
