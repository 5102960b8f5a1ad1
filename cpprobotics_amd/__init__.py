"""cpprobotics_amd — MI355X-native batched EKF / DARE-LQR / MPC engine ("crx").

The product is the C-ABI shared library ``libcrx.so`` (include/crx.h) built from the HIP
sources in ``cpprobotics_amd/csrc``.  This package is the thin Python plumbing used by the
tests and by bench.py: it loads the library with ctypes and passes raw device pointers of
torch tensors (torch is used for device memory, streams and torch.distributed only).

There is no CPU fallback: importing works without a GPU (so the library's symbols can be
checked), but every compute call raises ``CrxError`` when no HIP device is present.
"""
from ._lib import CrxError, lib, lib_path, EXPORTED_SYMBOLS  # noqa: F401
from .ekf import (  # noqa: F401
    ekf_default_QR, ekf_estimation, ekf_run, ekf_simulate_inputs, jacobF, jacobH, motion_model,
    normal_draws, observation_model,
)
from .lqr import dlqr, dlqr_from_v, solve_DARE, solve_DARE_from_v  # noqa: F401
from .mpc import mpc_n_vars, mpc_solve  # noqa: F401
from .dwa import dwa_control, dwa_default_config, dwa_run  # noqa: F401
from .frenet import FrenetCourse, frenet_default_config, frenet_num_paths, frenet_optimal_planning, frenet_run  # noqa: F401
from .pf import pf_default_params, pf_run  # noqa: F401
from . import host  # noqa: F401  (host-pointer entry points on numpy arrays, device selection / device set)
from .track import (  # noqa: F401
    Course, course_from_waypoints, calc_speed_profile, calc_nearest_index, calc_nearest_index_window, calc_ref_trajectory, closed_loop_prediction,
    lqr_steering_control, mpc_simulation, smooth_yaw, update, vehicle_params,
)

__all__ = [
    "CrxError", "lib", "lib_path", "EXPORTED_SYMBOLS",
    "motion_model", "jacobF", "jacobH", "observation_model", "ekf_estimation", "ekf_run",
    "ekf_simulate_inputs", "ekf_default_QR", "normal_draws",
    "solve_DARE", "dlqr", "solve_DARE_from_v", "dlqr_from_v",
    "mpc_solve", "mpc_n_vars",
    "pf_run", "pf_default_params", "dwa_run", "dwa_control", "dwa_default_config",
    "FrenetCourse", "frenet_default_config", "frenet_num_paths", "frenet_optimal_planning", "frenet_run",
    "Course", "course_from_waypoints", "calc_speed_profile", "calc_nearest_index", "lqr_steering_control", "update", "closed_loop_prediction",
    "calc_nearest_index_window", "calc_ref_trajectory", "mpc_simulation", "smooth_yaw", "vehicle_params",
]
