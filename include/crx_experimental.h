/* crx_experimental.h — measurement-only entry points of libcrx.so (prefix crx_x_).
 *
 * NOT part of the drop-in boundary (include/crx.h): these force a kernel variant or a launch geometry that the product entry
 * points choose by themselves, so that scripts/ and bench.py can A/B them on the same inputs.  Nothing here has a user in
 * cpprobotics_amd/ outside `cpprobotics_amd/experimental.py`; results are the same bits as the product path's. */
#ifndef CRX_EXPERIMENTAL_H
#define CRX_EXPERIMENTAL_H
#include <stddef.h>
#include "crx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* crx_dare_from_v_batch_dev with the register layout forced: lanes_per_agent = 1 (one agent per lane), 4 (one agent per DPP
 * quad) or 0 (what the product entry point would pick for this n). */
int crx_x_dare_from_v_lanes_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                                int* iters, void* stream, int lanes_per_agent);

/* crx_mpc_solve_batch_dev with the register layout forced: lanes_per_agent = 1 (one agent per lane), 4 (one agent per DPP quad,
 * the line search's step lengths rolled out side by side; T <= 24) or 0 (what the product entry point would pick). */
int crx_x_mpc_solve_lanes_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                              double* cost, void* stream, int lanes_per_agent);

/* The MPC solve with the source of the backward sweep's trig forced: recompute_trig = 0 the rollout stores sin / cos / tan for it, 1 the
 * sweep recomputes them (less memory traffic, more arithmetic).  Both give the same bits in every output; crx_mpc_solve_batch_dev picks by
 * batch size (csrc/mpc_kernels.hip.h: kMpcLeanFrom). */
int crx_x_mpc_solve_trig_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                             double* cost, void* stream, int recompute_trig);

/* The fused EKF run (crx_ekf_run_batch_dev without a covariance history) with the packed step's multiply-then-add pairs fused
 * (v_pk_fma_f32): the same operations in the same order, ~20 % fewer matrix instructions, +9 % throughput — and NOT the reference's
 * bits: 4.2e-7 floored relative error on configs[0]'s single vehicle, but up to 5.8e-6 over the 65,536 vehicles x 1000 steps of the
 * headline workload (measured against the oracle in every bench.py run, `extra.ekf_contracted`).  That is outside BASELINE's 1e-6, so
 * this is an experiment, not a mode of crx_ekf_params (VERDICT r5 item 3's own condition).  dt != 0.1 computes in the exact arithmetic. */
int crx_x_ekf_run_contracted_dev(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist,
                                 const float* Q, const float* R, const crx_ekf_params* prm, void* stream);

/* The MPC solve with the lane's working-set layout forced: store = 0 private memory (crx::mpc_kernel), 1 the tile layout of round 6
 * (crx::mpc_tile_kernel: controls in LDS, feedback gains in accumulator registers; T <= 21), 2 the same with every second knot kept in
 * one buffer (the odd knots and the accepted trajectory of a rollout are re-rolled).  Both give the same bits in every
 * output; crx_mpc_solve_batch_dev picks by horizon and batch size (csrc/api_mpc.inl). */
int crx_x_mpc_solve_store_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                              double* cost, void* stream, int store);

/* The tile layout with refilled lanes (crx::mpc_tile_refill_kernel): a wave owns agents_per_wave consecutive agents (64 .. 2^20),
 * finished lanes hand their agents back hold_lanes (1 .. 64) at a time and take the next ones.  Bit-identical per agent. */
int crx_x_mpc_solve_tile_refill_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                    double* cost, void* stream, int agents_per_wave, int hold_lanes);

/* ... with the layout chosen: store = 1 (crx_x_mpc_solve_tile_refill_dev) or 2 (the checkpointed tile layout: every second knot, one
 * buffer — csrc/mpc_kernels.hip.h).  Bit-identical per agent. */
int crx_x_mpc_solve_store_refill_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                     double* cost, void* stream, int store, int agents_per_wave, int hold_lanes);

/* crx_mpc_solve_batch_dev in TWO PHASES (round 6, measured and not selected): the ordinary launch with the sweep cap lowered to first_sweeps
 * on `stream`, then — behind an event, on tail_stream (may be the same stream) — the few agents that ran into that cap, solved from
 * scratch with prm's cap.  The solver is deterministic: every output is bit for bit what crx_mpc_solve_batch_dev writes, complete once
 * BOTH streams have passed the call.  status is required; work: (n + 64) ints of device memory owned by the call until then.
 * The idea: a launch lasts as long as its slowest agent, so a pipelined host gets its stream back sooner.  The measurement
 * (profiles/r06/swarm_two_phase_ab*.jsonl, mpc_store_ab_two_phase.jsonl): configs[4]'s round is bound by the solver's THROUGHPUT under
 * load, not by launch latency / depth — 0.48-0.53 ms with 7-9 straggler streams against 0.42-0.45 single-phase; and a lone batch LOSES,
 * because the single launch runs its straggler chains from t = 0 beside everything else, the second phase starts them after the first. */
int crx_x_mpc_solve_two_phase_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                  double* cost, int first_sweeps, int* work, void* stream, void* tail_stream);

/* crx_mpc_solve_batch_dev in PHASES (round 6): every agent still unconverged is swept up to sweep index caps[0]; the ones that get there
 * are suspended (their solver state saved), compacted into full waves and resumed up to caps[1], and so on; after the last cap to the
 * solver's own.  Per agent the same sweeps on the same doubles: bit for bit crx_mpc_solve_batch_dev's outputs.  A lockstep wave runs
 * as many sweeps as the slowest of its 64 agents (11.9 on average against a mean of 6.8); compaction takes most of that back.
 * status required; work: crx_x_mpc_phased_work_bytes(n, T) bytes of device memory; T <= 24. */
size_t crx_x_mpc_phased_work_bytes(int n, int T);
int crx_x_mpc_solve_phased_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                               double* cost, const int* caps, int ncaps, void* work, void* stream);

/* ... with the layout of the lane's working set chosen: store 0 private memory, 1 the tile layout (T <= 21). */
int crx_x_mpc_solve_phased_store_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                     double* cost, const int* caps, int ncaps, void* work, void* stream, int store);

/* crx_mpc_solve_batch_dev with the launch geometry forced: agents_per_wave in 1..64 (the low lanes of every wave), 1..4 waves per
 * workgroup.  The product entry point uses 64 and 1. */
int crx_x_mpc_solve_geometry_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                 double* cost, void* stream, int agents_per_wave, int waves_per_workgroup);

/* crx_lqr_closed_loop_batch_dev with the register layout forced: lanes_per_agent = 1 (one agent per lane), 4 (one agent per DPP
 * quad: the Riccati iteration one row per lane, the course scan split four ways; needs a course that fits in LDS) or 0 (what the
 * product entry point would pick for this n). */
int crx_x_lqr_closed_loop_lanes_dev(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                    const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                    float* traj_hist, int* ticks_done, void* stream, int lanes_per_agent);

/* crx_dare_batch_dev with the structure detection switched off: every agent through a DENSE kernel, whatever its matrices look
 * like (the product entry point serves agents whose arguments carry lqr_steering_control's pattern by the structured kernels) —
 * and with the dense kernel's register layout forced: lanes_per_agent = 1 (dare_dense_kernel: one agent per lane), 4
 * (dare_dense_quad_kernel: one row of X per lane of a quad) or 0 (what the product picks for this n: 4 up to 32,768 agents).
 * For the A/B of the paths and for tests that want the dense kernels on the reference's own matrices. */
int crx_x_dare_batch_dense_dev(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                               float eps, int maxiter, float* X, float* K, int* iters, void* stream, int lanes_per_agent);

/* Probe of the device's double-precision sin / cos (csrc/crx_dsincos.h: glibc 2.35's sin() / cos() restated for the Frenet
 * planner's frenet_optimal_trajectory.cpp:111-112): s[i] = sin(x[i]), c[i] = cos(x[i]) for |x[i]| < 105414336 (NaN beyond). */
int crx_x_dsincos_dev(int n, const double* x, double* s, double* c, void* stream);

/* Probes of the device's double-precision atan2(y, 1.0) (csrc/crx_datan2.h: glibc 2.35's atan2() restated for the feed-forward
 * term of lqr_steering_control, src/lqr_speed_steer_control.cpp:143, src/lqr_steer_control.cpp:126).
 * crx_x_datan2_dev: out[i] = atan2(y[i], 1.0).
 * crx_x_datan2_sweep_dev: ALL 2^32 float bit patterns k, y = L * (double)k.  sums[j], j = 0..4095 (device, 4096 x u64) = the sum
 * mod 2^64 of the result's bit patterns over the patterns [j << 20, (j + 1) << 20) — an order-independent checksum that
 * tests/tools/datan2_exhaustive.cpp forms from the host libm's atan2 (mode `sums`); ocml_diff (device, 3 x u64, may be NULL):
 * [0] = curvatures whose OCML device atan(y) differs from it in double, [1] = [2] = those whose rounded FLOAT differs (what rounds
 * 1-3 shipped); diff_k (device, 64 x u32, may be NULL; needs ocml_diff): the bit patterns of up to 64 such curvatures — the
 * adversarial inputs of tests/test_track_gpu.py. */
int crx_x_datan2_dev(int n, const double* y, double* out, void* stream);
int crx_x_datan2_sweep_dev(double L, unsigned long long* sums, unsigned long long* ocml_diff, unsigned* diff_k, void* stream);

/* crx_mpc_solve_batch_dev through the lane-refilling kernel (mpc_refill_kernel: a wave owns `agents_per_wave` consecutive agents;
 * once `hold_lanes` of its lanes hold a finished solve they write their solutions and take the next agents of the range; the line
 * search scheduled asynchronously across the lanes).  Measured and rejected twice (rounds 4 and 5): libcrx_x.so only.
 * scripts/gpu_mpc_refill_ab.py, scripts/gpu_mpc_variants_ab.py. */
int crx_x_mpc_solve_refill_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                               double* cost, void* stream, int agents_per_wave, int hold_lanes);


/* crx_dare_from_v_batch_dev with one agent per lane and the lane-refilling kernel forced (dare_from_v_refill_kernel: a wave owns
 * `agents_per_wave` consecutive agents; once `hold_lanes` of its lanes hold a finished agent they hand them back in one pass and
 * take the next agents of the range; what the product uses above 262,144 agents), or with agents_per_wave = -1 the masked kernel
 * of rounds 2-3 (lanes idle behind their wave's slowest agent).  scripts/gpu_dare_refill_ab.py. */
int crx_x_dare_from_v_refill_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                                 int* iters, void* stream, int agents_per_wave, int hold_lanes);

/* HBM calibration: a plain streaming kernel over `bytes` (a multiple of 16) with `workgroups` workgroups of 256 lanes, 16 bytes per
 * lane per access.  mode 0: dst = src, 1: read src only, 2: write dst only, 3: dst += 1 in place (the single-step EKF's traffic
 * shape: every line read, then written); mode + 8: the same with an XCD-contiguous workgroup order (every XCD one contiguous eighth
 * of the buffer; `workgroups` a multiple of 8); mode 1 / 2 + 16 (+ 8): the read-only / write-only kernel with plain instead of nontemporal accesses.
 * scripts/gpu_hbm_calib.py and bench.py (`extra.hbm_calibration`) time it next to the
 * HBM-bound EKF launches. */
/* What the FETCH_SIZE / WRITE_SIZE counters count for narrow accesses: mode 0 / 1 read `bytes` of src once, 4 / 8 bytes per lane; mode 2
 * writes and reads back a private array of 4 KiB per lane `passes` times (dst: one double per launched lane).  csrc/api_probes.inl. */
int crx_x_fetch_units_dev(int mode, const void* src, size_t bytes, double* dst, int workgroups, int passes, void* stream);
int crx_x_hbm_stream_dev(int mode, void* dst, const void* src, size_t bytes, int workgroups, void* stream);

/* The fused EKF step's reciprocal (v_rcp_f32 + one Newton step, csrc/ekf_math.h: recip_fast) against the IEEE quotient 1.0f / d on every
 * float 2^-60 <= |d| <= 2^60.  counts (device, 3 x u64): inputs walked, inputs where recip_fast differs, inputs where the six-fma form of
 * rounds 2-4 differs.  Both must be 0 on the device the engine runs on. */
int crx_x_recip_sweep_dev(unsigned long long* counts, void* stream);

/* crx_ekf_run_batch_dev through the 64-bit-address instantiations of the fused kernel whatever n is (the product entry point
 * switches to them above 4 M vehicles). */
int crx_x_ekf_run_addr64_dev(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist, float* P_hist,
                             const float* Q, const float* R, const crx_ekf_params* prm, void* stream);

/* Round 2's measured-and-rejected A/B variant of the fused EKF launch: two lanes per vehicle with DPP moves (0.61-0.73x of the
 * production kernel, profiles/r02/ekf_wave_ab.txt).  No general-step fallback: left_domain[0] != 0 afterwards means a vehicle left
 * the fast domain (|yaw| >= 120, extreme determinant) and xEst / PEst / x_hist of this call are NOT valid. */
int crx_x_ekf_run_pair_batch_dev(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist, const float Q[16],
                                 const float R[4], const crx_ekf_params* prm, int* left_domain, void* stream);

#ifdef __cplusplus
}
#endif
#endif
