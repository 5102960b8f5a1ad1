"""ctypes loader for libcrx.so (the C ABI declared in include/crx.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class CrxError(RuntimeError):
    pass


class EkfParams(C.Structure):
    _fields_ = [("dt", C.c_double)]


class LqrParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("L", C.c_double), ("eps", C.c_float), ("maxiter", C.c_int)]


class MpcParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "dt", "wb", "max_steer", "max_accel", "max_speed", "min_speed", "r_a", "r_delta", "rd_a",
        "rd_delta", "q_x", "q_y", "q_yaw", "q_v", "tol")] + [("max_iter", C.c_int), ("shared_gpu", C.c_int)]


class Course(C.Structure):
    _fields_ = [("n", C.c_int), ("cx", C.c_void_p), ("cy", C.c_void_p), ("cyaw", C.c_void_p), ("ck", C.c_void_p),
                ("sp", C.c_void_p)]


class VehicleParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("wheelbase", C.c_double), ("max_steer", C.c_double), ("clamp_speed", C.c_int),
                ("max_speed", C.c_double), ("min_speed", C.c_double)]


class LoopParams(C.Structure):
    _fields_ = [("goal_x", C.c_float), ("goal_y", C.c_float), ("goal_dis", C.c_float), ("kp", C.c_double),
                ("stop_speed", C.c_float), ("max_ticks", C.c_int)]


class DwaConfig(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("max_speed", "min_speed", "max_yawrate", "max_accel", "robot_radius", "max_dyawrate",
                                          "v_reso", "yawrate_reso", "dt", "predict_time", "to_goal_cost_gain", "speed_cost_gain")]


class FrenetConfig(C.Structure):
    _fields_ = ([(k, C.c_double) for k in ("max_speed", "max_accel", "max_curvature", "max_road_width", "d_road_w", "dt", "maxt",
                                           "mint", "target_speed", "d_t_s")] + [("n_s_sample", C.c_int)] +
                [(k, C.c_double) for k in ("robot_radius", "kj", "kt", "kd", "klat", "klon")])


class SwarmConfig(C.Structure):
    _fields_ = [("n", C.c_int), ("T", C.c_int), ("Tm", C.c_int), ("plan_every", C.c_int), ("depth", C.c_int), ("v_cmd", C.c_float),
                ("dl", C.c_float), ("dt_ref", C.c_double), ("nsearch", C.c_int), ("allow_shared_queues", C.c_int), ("planner_streams", C.POINTER(C.c_void_p)), ("ekf", EkfParams),
                ("mpc", MpcParams)]


class PfParams(C.Structure):
    _fields_ = [("rsim0", C.c_float), ("rsim1", C.c_float), ("Q", C.c_float), ("dt", C.c_double), ("nth", C.c_float)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_D = C.c_double
_CP = C.POINTER(Course)

# name -> (restype, argtypes); every symbol include/crx.h declares.
_SIGNATURES = {
    "crx_version": (_I, []),
    "crx_init": (_I, []),
    "crx_shutdown": (_I, []),
    "crx_device_count": (_I, []),
    "crx_host_libm_check": (_I, []),
    "crx_set_device": (_I, [_I]),
    "crx_get_device": (_I, []),
    "crx_set_devices": (_I, [_P, _I, _I]),
    "crx_get_devices": (_I, [_P, _I]),
    "crx_host_alloc": (_P, [C.c_size_t]),
    "crx_host_free": (None, [_P]),
    "crx_release_workspace": (_I, []),
    "crx_reserve_workspace": (_I, [C.c_size_t, C.c_size_t]),
    "crx_last_error": (C.c_char_p, []),
    "crx_ekf_default_params": (None, [C.POINTER(EkfParams)]),
    "crx_lqr_default_params": (None, [C.POINTER(LqrParams)]),
    "crx_mpc_default_params": (None, [C.POINTER(MpcParams)]),
    "crx_motion_model_batch": (_I, [_I, _P, _P, _P, C.POINTER(EkfParams)]),
    "crx_motion_model_batch_dev": (_I, [_I, _P, _P, _P, C.POINTER(EkfParams), _P]),
    "crx_jacobF_batch": (_I, [_I, _P, _P, _P, C.POINTER(EkfParams)]),
    "crx_jacobF_batch_dev": (_I, [_I, _P, _P, _P, C.POINTER(EkfParams), _P]),
    "crx_observation_model_batch": (_I, [_I, _P, _P]),
    "crx_observation_model_batch_dev": (_I, [_I, _P, _P, _P]),
    "crx_jacobH": (_I, [_P]),
    "crx_ekf_step_batch": (_I, [_I, _P, _P, _P, _P, _P, _P, C.POINTER(EkfParams)]),
    "crx_ekf_step_batch_dev": (_I, [_I, _P, _P, _P, _P, _P, _P, C.POINTER(EkfParams), _P]),
    "crx_ekf_run_batch": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(EkfParams)]),
    "crx_ekf_run_batch_dev": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(EkfParams), _P]),
    "crx_ekf_simulate_inputs_dev": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                         C.POINTER(EkfParams), _P]),
    "crx_normal_draws_dev": (_I, [_I, _I, C.c_longlong, C.c_ulonglong, C.c_uint, _P, _P]),
    "crx_dare_batch": (_I, [_I, _I, _P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "crx_dare_batch_dev": (_I, [_I, _I, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P]),
    "crx_dare_from_v_batch": (_I, [_I, _I, _P, C.POINTER(LqrParams), _P, _P, _P]),
    "crx_dare_from_v_batch_dev": (_I, [_I, _I, _P, C.POINTER(LqrParams), _P, _P, _P, _P]),
    "crx_mpc_solve_batch": (_I, [_I, _I, _P, _P, C.POINTER(MpcParams), _P, _P, _P]),
    "crx_mpc_solve_portfolio_batch": (_I, [_I, _I, _P, _P, C.POINTER(MpcParams), _P, _P, _P]),
    "crx_mpc_solve_batch_dev": (_I, [_I, _I, _P, _P, C.POINTER(MpcParams), _P, _P, _P, _P]),
    "crx_mpc_solve_portfolio_batch_dev": (_I, [_I, _I, _P, _P, C.POINTER(MpcParams), _P, _P, _P, _P]),
    "crx_vehicle_default_params": (None, [C.POINTER(VehicleParams), _I]),
    "crx_calc_nearest_index_batch": (_I, [_I, _P, _CP, _P, _P]),
    "crx_calc_nearest_index_batch_dev": (_I, [_I, _P, _CP, _P, _P, _P]),
    "crx_lqr_steering_control_batch": (_I, [_I, _I, _P, _CP, _P, _P, _P, C.POINTER(LqrParams), _P]),
    "crx_lqr_steering_control_batch_dev": (_I, [_I, _I, _P, _CP, _P, _P, _P, C.POINTER(LqrParams), _P, _P]),
    "crx_update_batch": (_I, [_I, _P, _P, _P, C.POINTER(VehicleParams)]),
    "crx_update_batch_dev": (_I, [_I, _P, _P, _P, C.POINTER(VehicleParams), _P]),
    "crx_lqr_closed_loop_batch": (_I, [_I, _I, _P, _CP, _P, _P, _P, C.POINTER(LqrParams), C.POINTER(VehicleParams),
                                       C.POINTER(LoopParams), _P, _P]),
    "crx_lqr_closed_loop_batch_dev": (_I, [_I, _I, _P, _CP, _P, _P, _P, C.POINTER(LqrParams), C.POINTER(VehicleParams),
                                           C.POINTER(LoopParams), _P, _P, _P]),
    "crx_calc_nearest_index_window_batch": (_I, [_I, _P, _CP, _P, _I, _P]),
    "crx_calc_nearest_index_window_batch_dev": (_I, [_I, _P, _CP, _P, _I, _P, _P]),
    "crx_calc_ref_trajectory_batch": (_I, [_I, _I, _P, _CP, _F, _D, _I, _P, _P]),
    "crx_calc_ref_trajectory_batch_dev": (_I, [_I, _I, _P, _CP, _F, _D, _I, _P, _P, _P]),
    "crx_dwa_default_config": (None, [C.POINTER(DwaConfig)]),
    "crx_dwa_run_batch_dev": (_I, [_I, _I, _P, _P, _P, _P, _I, C.POINTER(DwaConfig), _P, _P, _P, _P, _P, _P]),
    "crx_frenet_default_config": (None, [C.POINTER(FrenetConfig)]),
    "crx_frenet_num_paths": (_I, [C.POINTER(FrenetConfig)]),
    "crx_frenet_spline_build": (_I, [_P, _P, _I, _P]),
    "crx_frenet_course_samples": (_I, [_P, _I, _P, _P, _I]),
    "crx_course_from_waypoints": (_I, [_P, _P, _I, _D, _P, _P, _P, _P, _I]),
    "crx_calc_speed_profile": (_I, [_I, _P, _P, _P, _I, _F, _P]),
    "crx_smooth_yaw": (_I, [_P, _I]),
    "crx_frenet_run_batch_dev": (_I, [_I, _I, _P, _P, _I, _P, _P, _I, C.POINTER(FrenetConfig), _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "crx_pf_default_params": (None, [C.POINTER(PfParams)]),
    "crx_pf_run_batch_dev": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(PfParams), _P, _P, _P]),
    "crx_mpc_closed_loop_work_bytes": (C.c_size_t, [_I, _I]),
    "crx_mpc_closed_loop_batch_dev": (_I, [_I, _I, _P, _CP, _F, _I, C.POINTER(MpcParams), C.POINTER(LoopParams), _P, _P, _P,
                                           _P, _P]),
    "crx_mpc_closed_loop_flags_batch_dev": (_I, [_I, _I, _P, _CP, _F, _I, C.POINTER(MpcParams), C.POINTER(LoopParams), _P, _P, _P,
                                                 _P, _P]),
    "crx_mpc_closed_loop_batch": (_I, [_I, _I, _P, _CP, _F, _I, C.POINTER(MpcParams), C.POINTER(LoopParams), _P, _P, _P, _P]),
    "crx_hw_queues": (_I, []),
    "crx_swarm_default_config": (None, [C.POINTER(SwarmConfig)]),
    "crx_swarm_create": (_I, [C.POINTER(_P), C.POINTER(SwarmConfig), _CP, _P, _P, _P, _P]),
    "crx_swarm_round_dev": (_I, [_P, _P, _P, _P, _P, C.POINTER(C.c_longlong)]),
    "crx_swarm_plans": (_I, [_P, C.c_longlong, C.POINTER(_I), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "crx_swarm_state": (_P, [_P]),
    "crx_swarm_wait": (_I, [_P, _P]),
    "crx_swarm_destroy": (_I, [_P]),
    "crx_comm_unique_id": (_I, [_P]),
    "crx_comm_init_rank": (_I, [C.POINTER(_P), _P, _I, _I]),
    "crx_comm_rank": (_I, [_P]),
    "crx_comm_world": (_I, [_P]),
    "crx_allgather_dev": (_I, [_P, _P, _P, C.c_size_t, _P]),
    "crx_comm_destroy": (_I, [_P]),
}
EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


def lib_path():
    # CRX_LIB_PATH: developer override used for A/B-ing alternative builds of the same ABI
    return os.environ.get("CRX_LIB_PATH") or os.path.join(_HERE, "libcrx.so")


_lib = None


def lib():
    """Load libcrx.so once.  torch is imported first so that the library binds to the HIP
    runtime torch already mapped (same soname, libamdhip64.so.7) — one runtime per process."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise CrxError(
            f"{path} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C cpprobotics_amd/csrc`). "
            "crx has no CPU fallback.")
    import torch  # noqa: F401  (maps torch's libamdhip64 first)
    l = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise CrxError(f"libcrx.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def check(rc, what):
    if rc != 0:
        msg = lib().crx_last_error().decode("utf-8", "replace")
        raise CrxError(f"{what} failed with status {rc}: {msg}")


def require_cuda(*tensors, dtypes=None):
    """Every tensor handed to a *_dev entry point: on the device, contiguous, float32 or int32 (the kernels reinterpret the
    bytes as float4/float2/int — any other dtype would be silently misread) and 16-byte aligned (vector loads)."""
    import torch
    if not torch.cuda.is_available():
        raise CrxError("no HIP device visible: crx has no CPU fallback")
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise CrxError("crx *_dev entry points need device tensors")
        if not t.is_contiguous():
            raise CrxError("crx needs contiguous tensors")
        if t.dtype not in (dtypes or (torch.float32, torch.int32)):
            raise CrxError(f"crx needs float32 (or int32 index) tensors, got {t.dtype}")
        if t.numel() and t.data_ptr() % 16:
            raise CrxError("crx needs 16-byte aligned tensors (got an offset view; .clone() it)")


def expect(name, t, kind, *shape, optional=False):
    """Shape/dtype contract of one argument: kind 'f' = float32, 'd' = float64, 'i' = int32; shape entries may be None (any)."""
    import torch
    if t is None:
        if optional:
            return
        raise CrxError(f"{name}: missing")
    want = {"f": torch.float32, "d": torch.float64, "i": torch.int32}[kind]
    if t.dtype != want:
        raise CrxError(f"{name}: expected {want}, got {t.dtype}")
    if len(shape) != t.dim() or any(s is not None and int(s) != int(d) for s, d in zip(shape, t.shape)):
        raise CrxError(f"{name}: expected shape {tuple('*' if s is None else s for s in shape)}, got {tuple(t.shape)}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def host_floats(values, n):
    import numpy as np
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
    if a.size != n:
        raise CrxError(f"expected {n} floats, got {a.size}")
    return a


# The committed hardware counters (profiles/traffic.json, side_counters.json, mpc_traffic.json — rocprofv3 PMC passes cannot run inside
# bench.py) carry a hash of the CODE of the kernels they were taken from: the instruction stream of those functions in the gfx950 code
# object of the libcrx.so that was loaded (mnemonics and operands; addresses, encodings and pc-relative distances stripped, so that a change elsewhere in the
# library — or a comment — does not invalidate them, and any change to the kernels' code does).  scripts/summarize_prof.py and
# scripts/gpu_mpc_traffic.sh write it, bench.py prints the counters only when the library it runs has the same hash.
KERNEL_FAMILIES = {
    "ekf": ("ekf_run_kernel",),
    "side": ("dare_", "mpc_kernel", "mpc_portfolio_kernel", "ekf_step_kernel", "lqr_closed_loop"),
    "mpc": ("mpc_kernel", "mpc_tile_kernel", "mpc_tile_refill_kernel", "mpc_tile_lite_kernel"),
}


def llvm_bin_dir():
    """Where clang-offload-bundler / llvm-objdump of the ROCm that built the library live: next to $HIPCC's ROCm, under $ROCM_PATH,
    under `hipconfig --rocmpath`, else /opt/rocm — the same precedence csrc/Makefile gives HIPCC (this was hard-coded)."""
    import shutil
    import subprocess
    cands = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    hc = shutil.which("hipconfig")
    if hc:
        try:
            cands.append(os.path.join(subprocess.run([hc, "--rocmpath"], capture_output=True, text=True, timeout=20).stdout.strip(), "lib", "llvm", "bin"))
        except Exception:
            pass
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")) and os.path.exists(os.path.join(c, "clang-offload-bundler")):
            return c
    raise FileNotFoundError("no ROCm LLVM tools (llvm-objdump, clang-offload-bundler) under any of: " + ", ".join(cands))


_code_hashes = {}


def disassemble_code_object(lib):
    """llvm-objdump -d of the gfx950 code object embedded in a libcrx build (also what scripts/check_isa.py reads)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        import shutil
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
        llvm = llvm_bin_dir()
        # binutils' objcopy where the host has it, ROCm's llvm-objcopy otherwise
        objcopy = shutil.which("objcopy") or os.path.join(llvm, "llvm-objcopy")
        subprocess.check_call([objcopy, "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
        text = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
        # the MPC tile kernels are a second code object, embedded raw in the section .crx_tile_hsaco (csrc/Makefile)
        tile = os.path.join(d, "tile.hsaco")
        subprocess.check_call([objcopy, "-O", "binary", "--only-section=.crx_tile_hsaco", lib, tile])
        if os.path.exists(tile) and os.path.getsize(tile) > 0:
            text += subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", tile], capture_output=True, text=True, check=True).stdout
        return text


def _functions_of_disassembly(text):
    """{function: [instruction, ...]} of an llvm-objdump -d listing: mnemonics and operands only."""
    import re
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
        elif cur is not None and line.startswith("\t"):
            ins = line.split("//")[0].strip()
            # the literal of the s_add_u32 behind an s_getpc_b64 is the distance to a global (a constant table, a symbol's GOT
            # slot): it moves whenever ANY function of the library changes size — layout, not this kernel's code
            if ins.startswith("s_add_u32") and cur and cur[-1].startswith("s_getpc_b64"):
                ins = re.sub(r"0x[0-9a-f]+$|\d+$", "<pcrel>", ins)
            cur.append(ins)
    return funcs


def kernel_code_hash(family, lib=None):
    """sha256 (16 hex digits) over the instruction streams of a kernel family in the library's gfx950 code object, or None when the
    code object cannot be read on this host (no ROCm LLVM tools)."""
    import hashlib
    lib = lib or lib_path()
    key = (lib, os.path.getmtime(lib))
    if key not in _code_hashes:
        try:
            _code_hashes[key] = _functions_of_disassembly(disassemble_code_object(lib))
        except Exception:
            _code_hashes[key] = None
    funcs = _code_hashes[key]
    if not funcs:
        return None
    h = hashlib.sha256()
    for name in sorted(funcs):
        if any(p in name for p in KERNEL_FAMILIES[family]):
            h.update(name.encode() + b"\0" + "\n".join(funcs[name]).encode() + b"\0")
    return h.hexdigest()[:16]
