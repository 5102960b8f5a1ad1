/* crx.h — C ABI of the crx engine: batched EKF / DARE-LQR / MPC for MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of onlytailei/CppRobotics.  The reference
 * exposes no library and no FFI; its "API" for this path is a set of free functions living in
 * the same translation unit as main().  Every entry point below names the reference function
 * it replaces (paths relative to the reference checkout).  include/crx_dropin.hpp re-creates
 * the reference's exact C++ signatures on top of this header.
 *
 * Conventions
 *   - All matrices are column-major and densely packed, exactly what Eigen's fixed-size
 *     `.data()` returns: M(i,j) = m[i + rows*j].
 *   - "batch" = n independent agents.  Agent k's 4-vector starts at x + 4*k, its 4x4 at
 *     P + 16*k, and so on (array-of-structures, i.e. a std::vector<Eigen::Matrix4f>).
 *   - Functions ending in `_dev` take DEVICE pointers for all batched arrays and a HIP stream
 *     (`void* stream` is a hipStream_t; NULL = the null stream); they only enqueue work.
 *     Functions without the suffix take HOST pointers, copy in/out, and synchronise.
 *   - Small per-launch constants (Q, R, params structs) are always HOST pointers.
 *   - Return value: 0 on success, a negative crx_status otherwise (the reference has no error
 *     channel at all; singular matrices silently produce inf/NaN there and here alike).
 *   - There is no CPU fallback.  Without a HIP device every compute entry point returns
 *     CRX_ERR_NO_DEVICE.
 */
#ifndef CRX_H_
#define CRX_H_

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum crx_status {
  CRX_OK = 0,
  CRX_ERR_INVALID = -1,    /* bad argument (n < 0, NULL pointer, unsupported dim, ...) */
  CRX_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime error at init               */
  CRX_ERR_HIP = -3,        /* a HIP call failed; see crx_last_error()                 */
  CRX_ERR_ALLOC = -4
} crx_status;

/* ---- runtime ------------------------------------------------------------------------- */
int crx_version(void);                 /* 10000*major + 100*minor + patch                 */
int crx_init(void);                    /* optional: device check + runtime warm-up         */
int crx_shutdown(void);                /* optional: drain the devices, release workspaces  */
int crx_device_count(void);            /* number of HIP devices visible (0 if none)       */
const char* crx_last_error(void);      /* thread-local, never NULL                        */

/* Bit parity with a reference built on a given host needs that host's libm to be the one the kernels restate (glibc 2.35's
 * x86-64 FMA build: sinf, cosf, expf, atanf, atan2f, tanf, acosf, double sin / cos / atan2).  crx_host_libm_check() compares the
 * two on 200,000 pseudo-random arguments per family and returns 0 when all agree, else a bit mask (bit 0 sinf / cosf, 1 expf,
 * 2 atanf / atan2f / tanf / acosf, 3 double sin / cos, 4 double atan2(y, 1)).  Needs no device; a few milliseconds. */
int crx_host_libm_check(void);

/* Devices (since 0.4).  The `_dev` entry points launch on the calling thread's current device (crx_set_device = hipSetDevice);
 * their pointers and stream must belong to it.  The host-pointer entry points use the calling thread's current device too —
 * unless a device set is installed: then every host-pointer BATCH entry point (crx_ekf_run_batch, crx_ekf_step_batch,
 * crx_dare_batch, crx_dare_from_v_batch, crx_mpc_solve_batch, the tracking functions and both closed loops) splits its agents
 * [0, n) contiguously over devices[0 .. ndev) — shard r on devices[r], the first n % G shards one agent longer, as few shards as
 * keep min_agents_per_device agents each —, one host thread and one set of streams per shard, results landing directly in the
 * caller's arrays.  No collective: agents never read one another (src/extended_kalman_filter.cpp:64-78,
 * src/lqr_speed_steer_control.cpp:85-151, src/model_predictive_control.cpp:255-346).  The result does not depend on the
 * partition (tests/test_host_boundary_gpu.py: a forced 3-way split on one GPU equals the unsplit call bit for bit).  Shared inputs (Q, R, params, the
 * course) are replicated.  ndev = 0 removes the set; devices = NULL means 0 .. ndev-1; a device may be listed more than once.
 * Serves the reference's fleet-sized callers: the EKF main loop (src/extended_kalman_filter.cpp:171-183) and the tracking loops
 * (src/lqr_speed_steer_control.cpp:194-205, src/model_predictive_control.cpp:371-385) for n vehicles on 1 .. N GPUs. */
int crx_set_device(int device);
int crx_get_device(void);              /* -1 without a device */
int crx_set_devices(const int* devices, int ndev, int min_agents_per_device);
int crx_get_devices(int* devices, int cap);   /* returns the size of the set (0: none installed) */

/* Host-pointer calls keep per-device workspaces (device memory + pinned staging) that grow to the largest call seen and are
 * reused — no allocation in steady state.  Small calls (arguments <= 256 KB in all: the literal one-vehicle drop-in calls) are
 * zero-copy: one pinned block the kernel reads and writes across PCIe.  Large pageable arrays are staged through pinned rings by
 * copy threads; arrays from crx_host_alloc (pinned) are DMA'd in place — the fastest way across the bus.
 * crx_release_workspace gives the workspaces back (so does crx_shutdown). */
void* crx_host_alloc(size_t bytes);    /* pinned host memory (NULL on failure) */
void crx_host_free(void* p);
int crx_release_workspace(void);
/* Grow the CURRENT device's workspaces now (sizes of the largest call to come), so that no later call pays for the growth. */
int crx_reserve_workspace(size_t device_bytes, size_t pinned_bytes);

/* ---- EKF localisation (src/extended_kalman_filter.cpp) --------------------------------- */
typedef struct crx_ekf_params {
  double dt;   /* `#define DT 0.1` (src/extended_kalman_filter.cpp:17); double on purpose:   */
               /* the reference forms DT*cos(yaw) etc. in double before rounding to float.   */
} crx_ekf_params;
void crx_ekf_default_params(crx_ekf_params* p);

/* motion_model(x,u)  src/extended_kalman_filter.cpp:22-36.  x_out may alias x. */
int crx_motion_model_batch(int n, const float* x, const float* u, float* x_out,
                           const crx_ekf_params* prm);
int crx_motion_model_batch_dev(int n, const float* x, const float* u, float* x_out,
                               const crx_ekf_params* prm, void* stream);
/* jacobF(x,u)  src/extended_kalman_filter.cpp:38-47.  jF: n x 16 col-major. */
int crx_jacobF_batch(int n, const float* x, const float* u, float* jF,
                     const crx_ekf_params* prm);
int crx_jacobF_batch_dev(int n, const float* x, const float* u, float* jF,
                         const crx_ekf_params* prm, void* stream);
/* observation_model(x)  src/extended_kalman_filter.cpp:50-55.  z_out: n x 2. */
int crx_observation_model_batch(int n, const float* x, float* z_out);
int crx_observation_model_batch_dev(int n, const float* x, float* z_out, void* stream);
/* jacobH()  src/extended_kalman_filter.cpp:57-62.  Writes the constant 2x4 (col-major, 8 floats). */
int crx_jacobH(float* jH_out);

/* ekf_estimation(xEst,PEst,z,u,Q,R)  src/extended_kalman_filter.cpp:64-78, n agents, one step.
 * x (n x 4) and P (n x 16) are updated in place; z, u are n x 2; Q is 4x4, R is 2x2 (shared). */
int crx_ekf_step_batch(int n, float* x, float* P, const float* z, const float* u,
                       const float* Q, const float* R, const crx_ekf_params* prm);
int crx_ekf_step_batch_dev(int n, float* x, float* P, const float* z, const float* u,
                           const float* Q, const float* R, const crx_ekf_params* prm,
                           void* stream);

/* T consecutive ekf_estimation() calls per agent in ONE launch — the body of the reference's
 * `while(time <= SIM_TIME)` loop (src/extended_kalman_filter.cpp:171-188) minus RNG/drawing.
 * z, u are time-major [T][n][2]; x_hist (may be NULL) receives xEst after every step,
 * time-major [T][n][4] (the reference's `hxEst.push_back(xEst)`, :187).
 * P_hist (may be NULL) receives PEst after every step, [T][n][16]. */
int crx_ekf_run_batch(int n, int T, float* x, float* P, const float* z, const float* u,
                      float* x_hist, float* P_hist, const float* Q, const float* R,
                      const crx_ekf_params* prm);
int crx_ekf_run_batch_dev(int n, int T, float* x, float* P, const float* z, const float* u,
                          float* x_hist, float* P_hist, const float* Q, const float* R,
                          const crx_ekf_params* prm, void* stream);

/* The input side of the reference's simulation loop (src/extended_kalman_filter.cpp:174-181):
 *   ud = u + w[0:2]*diag(Qsim) ; xTrue = motion_model(xTrue,u) ; xDR = motion_model(xDR,ud) ;
 *   z  = xTrue[0:2] + w[2:4]*diag(Rsim)
 * with the caller supplying the standard-normal draws w [T][n][4] (the reference seeds from
 * std::random_device, so no draw sequence of its own can be reproduced).
 * u_true: n x 2.  xTrue, xDR: n x 4, updated in place.  Outputs z, ud: [T][n][2].
 * xTrue_hist / xDR_hist ([T][n][4]) may be NULL.  qsim[2], rsim[2] are the diagonal entries
 * Qsim(0,0),Qsim(1,1),Rsim(0,0),Rsim(1,1) (:154-161) as floats. */
/* Standard-normal draws for synthetic inputs, keyed by (seed, stream_id, GLOBAL agent id, step) with Philox4x32-10 + Box-Muller
 * (csrc/crx_philox.h): w[t][a][0..3] = the four draws pass t of the reference's loop consumes for agent agent0 + a
 * (src/extended_kalman_filter.cpp:174-181; the reference's own generator is random_device-seeded, :162-164).  The bytes do not
 * depend on n or agent0's alignment to a shard: a swarm sharded over G GPUs sees what one GPU would generate. */
int crx_normal_draws_dev(int n, int T, long long agent0, unsigned long long seed, unsigned stream_id,
                         float* w /* [T][n][4] */, void* stream);
int crx_ekf_simulate_inputs_dev(int n, int T, const float* u_true, float* xTrue, float* xDR,
                                const float* w, float* z, float* ud, float* xTrue_hist,
                                float* xDR_hist, const float qsim[2], const float rsim[2],
                                const crx_ekf_params* prm, void* stream);

/* ---- LQR: DARE fixed point + gain ------------------------------------------------------- */
/* solve_DARE / dlqr, 5x5 state, 2 inputs: src/lqr_speed_steer_control.cpp:85-100, 102-106.
 * solve_DARE / dlqr, 4x4 state, 1 input : src/lqr_steer_control.cpp:75-90, 92-96.
 * dim selects the variant (5 -> B is 5x2, R is 2x2, K is 2x5; 4 -> B is 4x1, R is 1x1, K is 1x4).
 * A: n x dim*dim, B: n x dim*m, Q: n x dim*dim, R: n x m*m  (per-agent matrices).
 * Outputs (any may be NULL): X n x dim*dim, K n x m*dim, iters n (number of fixed-point
 * evaluations performed; == maxiter when the cap was hit — the reference returns the most
 * recent iterate in that case too, silently).
 * eps / maxiter are the reference's locals `eps = 0.01`, `maxiter = 150`. */
int crx_dare_batch(int n, int dim, const float* A, const float* B, const float* Q,
                   const float* R, float eps, int maxiter, float* X, float* K, int* iters);
int crx_dare_batch_dev(int n, int dim, const float* A, const float* B, const float* Q,
                       const float* R, float eps, int maxiter, float* X, float* K, int* iters,
                       void* stream);

typedef struct crx_lqr_params {
  double dt;     /* `#define DT 0.1`  src/lqr_speed_steer_control.cpp:20 */
  double L;      /* `#define L 0.5`   src/lqr_speed_steer_control.cpp:21 */
  float eps;     /* 0.01              src/lqr_speed_steer_control.cpp:88 */
  int maxiter;   /* 150               src/lqr_speed_steer_control.cpp:87 */
} crx_lqr_params;
void crx_lqr_default_params(crx_lqr_params* p);

/* Same solve, but A and B are built on the fly from the vehicle speed exactly as
 * lqr_steering_control() builds them (src/lqr_speed_steer_control.cpp:116-129 for dim 5,
 * src/lqr_steer_control.cpp:104-115 for dim 4), with Q = I, R = I.  v: n floats. */
int crx_dare_from_v_batch(int n, int dim, const float* v, const crx_lqr_params* prm, float* X,
                          float* K, int* iters);
int crx_dare_from_v_batch_dev(int n, int dim, const float* v, const crx_lqr_params* prm,
                              float* X, float* K, int* iters, void* stream);

/* ---- MPC speed + steer (src/model_predictive_control.cpp) ------------------------------- */
typedef struct crx_mpc_params {
  double dt;         /* `#define DT 0.2`            :26 */
  double wb;         /* `#define WB 2.5`            :36 */
  double max_steer;  /* 45 deg                      :27 */
  double max_accel;  /* 1.0                         :39 */
  double max_speed;  /* 55/3.6                      :37 */
  double min_speed;  /* -20/3.6                     :38 */
  double r_a, r_delta;    /* 0.01, 0.01   input cost      :203-204 */
  double rd_a, rd_delta;  /* 0.01, 1.0    input-rate cost :208-209 */
  double q_x, q_y, q_yaw, q_v; /* 1, 1, 0.5, 0.5  tracking cost :247-250 */
  double tol;        /* stop when the projected-gradient norm and the step are below tol */
  int max_iter;      /* outer iteration cap (the reference: IPOPT max_iter 50, :326)     */
  int shared_gpu;    /* performance hint, never changes an answer (every form of the solve computes the same bits): 0 = this launch
                        has the GPU to itself, the kernel form is picked by its size; nonzero = other launches run next to it (a
                        pipelined host): from 16,384 agents on the form with the least memory traffic is used
                        (configs[4]'s round: 0.481 -> 0.466 ms).  Sits in what was padding: sizeof and every offset are unchanged. */
} crx_mpc_params;
void crx_mpc_default_params(crx_mpc_params* p);

/* mpc_solve(State x0, M_XREF traj_ref)  src/model_predictive_control.cpp:255-346 for n agents.
 * T = number of knot points (the reference's `#define T 6`; BASELINE's "N=20" is T = 21).
 * x0: n x 4 (x,y,yaw,v).  xref: n x (4*T), each agent's block a column-major 4 x T matrix
 * (Eigen::Matrix<float,NX,T>::data()).
 * sol: n x (4T + 2(T-1)) floats in the reference's variable layout
 *      [x(T) | y(T) | yaw(T) | v(T) | delta(T-1) | a(T-1)]   (:54-60, :341-345).
 * status (may be NULL): per agent, bit0 = converged to tol, bit1 = a speed knot of the returned trajectory lies outside
 * [min_speed, max_speed] — which happens only when the START speed x0.v does (every rollout clamps the acceleration so that
 * later knots stay inside; with an infeasible start the acceleration limits win) —, bits 8.. = iterations used.  cost (may be NULL): final objective (double).
 * An agent's answer (sol, status, cost: every bit) depends on its own problem only — not on the batch it travels in, its size, the launch
 * geometry or which form of the kernel the batch size selects (tests/test_mpc_gpu.py). */
int crx_mpc_solve_batch(int n, int T, const float* x0, const float* xref,
                        const crx_mpc_params* prm, float* sol, int* status, double* cost);
int crx_mpc_solve_batch_dev(int n, int T, const float* x0, const float* xref,
                            const crx_mpc_params* prm, float* sol, int* status, double* cost,
                            void* stream);
/* The same solve with a PORTFOLIO of four variants of the solver's globalisation (leading Gauss-Newton sweeps 2 / 3 / 2 / 1, trust
 * box of a Newton step x1 / x1 / x2 / x2; variant 0 is crx_mpc_solve_batch_dev's solver): an agent is a quad of lanes, the variants
 * run in lockstep and the first to converge — fewest sweeps, ties to the lowest variant — is the agent's answer.  Same NLP, same
 * tolerance, same layout of sol; never more sweeps than crx_mpc_solve_batch_dev and a far shorter tail (slowest agent of 8,192: 13-14
 * sweeps on six seeds against 16-28), at four times the lanes and a dearer sweep: for batches that leave the GPU idle (<= 8 k agents),
 * where the launch lasts as long as its slowest agent — measured 1.01-1.53x at 8,192 agents x T = 21, < 1 at T = 6 and at 16,384
 * agents without a straggler (profiles/r04/mpc_portfolio_ab.jsonl): an insurance against the tail, opt-in.  In rare cases the winning variant settles in a different local optimum of the
 * (non-convex) NLP than variant 0 would.  status: as crx_mpc_solve_batch, plus the winning variant in bits 2-3.  Reproduced by the CPU
 * twin (oracle_mpc_solve_portfolio).  Enqueue only, like every _dev entry point. */
int crx_mpc_solve_portfolio_batch_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm,
                                      float* sol, int* status, double* cost, void* stream);
int crx_mpc_solve_portfolio_batch(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol,
                                  int* status, double* cost);   /* host pointers */


/* ---- course tracking front-end, vehicle update, closed loops ------------------------------------
 * The callers on either side of the solves.
 * State: n x 4 floats (x, y, yaw, v) = the reference's `struct State` (include/motion_model.h:31-42).
 * The course is shared by all agents: five arrays of `n` floats (cx, cy, cyaw, ck, sp), as the
 * reference's main() builds them with Spline2D + calc_speed_profile (course generation itself is out
 * of scope).  In the _dev entry points the arrays are device pointers. */
typedef struct crx_course {
  int n;
  const float* cx;
  const float* cy;
  const float* cyaw;
  const float* ck;   /* curvature; only lqr_steering_control reads it */
  const float* sp;   /* speed profile */
} crx_course;

typedef struct crx_vehicle_params {  /* update(State&, a, delta) */
  double dt;          /* DT:  0.1 in the LQR files, 0.2 in the MPC file                                  */
  double wheelbase;   /* L 0.5 (src/lqr_speed_steer_control.cpp:21) / WB 2.5 (model_predictive_control.cpp:36) */
  double max_steer;   /* 45.0/180*M_PI                                                                  */
  int clamp_speed;    /* 0: LQR update (:154-164); 1: MPC update (:69-81) clamps v to [min_speed, max_speed] */
  double max_speed, min_speed;   /* 55.0/3.6, -20.0/3.6 (:37-38)                                        */
} crx_vehicle_params;
void crx_vehicle_default_params(crx_vehicle_params* p, int mpc /* 0: LQR files, 1: MPC file */);

/* calc_nearest_index(state, cx, cy, cyaw, ind)  src/lqr_speed_steer_control.cpp:65-83 (full scan; returns the
 * signed SQUARED distance in e, the index through ind — in/out: an agent whose position compares smaller to no
 * course point (NaN) keeps its incoming ind, as the reference's reference parameter does). */
int crx_calc_nearest_index_batch(int n, const float* state, const crx_course* course, int* ind, float* e);
int crx_calc_nearest_index_batch_dev(int n, const float* state, const crx_course* course, int* ind, float* e, void* stream);

/* lqr_steering_control.
 * dim 5: src/lqr_speed_steer_control.cpp:108-151 — control: n x 2 {ai, delta}; ind (may be NULL): out only.
 * dim 4: src/lqr_steer_control.cpp:98-133        — control: n x 1 {delta};     ind: in/out (the caller's persistent index).
 * pe, pth_e: n floats, in/out (previous lateral / heading error). */
int crx_lqr_steering_control_batch(int n, int dim, const float* state, const crx_course* course, int* ind, float* pe,
                                   float* pth_e, const crx_lqr_params* prm, float* control);
int crx_lqr_steering_control_batch_dev(int n, int dim, const float* state, const crx_course* course, int* ind, float* pe,
                                       float* pth_e, const crx_lqr_params* prm, float* control, void* stream);

/* update(state, a, delta) for n agents, in place. */
int crx_update_batch(int n, float* state, const float* a, const float* delta, const crx_vehicle_params* prm);
int crx_update_batch_dev(int n, float* state, const float* a, const float* delta, const crx_vehicle_params* prm, void* stream);

/* closed_loop_prediction, maths only (src/lqr_speed_steer_control.cpp:194-205 for dim 5, src/lqr_steer_control.cpp:186-196
 * for dim 4): per tick lqr_steering_control -> update -> goal test, as ONE kernel with the agent state in registers.
 * The reference's loops never advance their clock and only end at the goal; max_ticks bounds them here.
 * state: in/out.  pe, pth_e, ind: in/out per-agent loop variables (NULL = start from 0 as the reference does).
 * kp, stop_speed: the 4-state loop's `KP` and `stop_speed` (ignored for dim 5).
 * ticks_done (may be NULL): ticks executed per agent (== max_ticks if the goal was not reached).
 * traj_hist (may be NULL): [max_ticks][n][4], the state after each executed tick. */
typedef struct crx_loop_params {
  float goal_x, goal_y, goal_dis;   /* goal_dis: 0.3 (5-state file :168) / 0.5 (4-state file :148, MPC :353) */
  double kp;                        /* KP 1.0 (src/lqr_steer_control.cpp:22) */
  float stop_speed;                 /* 0.05 */
  int max_ticks;
} crx_loop_params;
int crx_lqr_closed_loop_batch(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                              const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                              float* traj_hist, int* ticks_done);
int crx_lqr_closed_loop_batch_dev(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                  const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                  float* traj_hist, int* ticks_done, void* stream);

/* MPC front-end: calc_nearest_index(state, cx, cy, cyaw, pind) src/model_predictive_control.cpp:107-127 (window of
 * nsearch = N_IND_SEARCH points from pind; the reference's unchecked read past the course end is clipped) and
 * calc_ref_trajectory :130-170 (xref: n x 4T, column-major 4 x T per agent; target_ind in/out). */
int crx_calc_nearest_index_window_batch(int n, const float* state, const crx_course* course, const int* pind, int nsearch,
                                        int* ind_out);
int crx_calc_nearest_index_window_batch_dev(int n, const float* state, const crx_course* course, const int* pind, int nsearch,
                                            int* ind_out, void* stream);
int crx_calc_ref_trajectory_batch(int n, int T, const float* state, const crx_course* course, float dl, double dt, int nsearch,
                                  int* target_ind, float* xref);
int crx_calc_ref_trajectory_batch_dev(int n, int T, const float* state, const crx_course* course, float dl, double dt,
                                      int nsearch, int* target_ind, float* xref, void* stream);

/* mpc_simulation's loop (src/model_predictive_control.cpp:371-385), maths only: per tick calc_ref_trajectory -> mpc_solve -> update
 * (first control of the solution) -> goal test, for n agents, the WHOLE episode in one persistent kernel enqueued on `stream`
 * (state, target_ind and the reference trajectory of an agent never leave the lane between ticks).  Agents that reached the
 * goal stop being updated.  target_ind: in/out.
 * crx_mpc_closed_loop_batch_dev keeps its 0.2 signature: `work` is ignored (crx_mpc_closed_loop_work_bytes() is 0) and never
 * written, so a caller built against 0.2 that passes a dummy pointer stays safe.  (0.3.0 had put an `int* solve_flags` the kernel
 * writes n ints into in that slot under the same symbol; 0.4 moved it to a symbol of its own.)
 * crx_mpc_closed_loop_flags_batch_dev (since 0.4) — the same launch plus solve_flags (may be NULL; n ints): per agent, bit 0 = at
 * least one tick's solve did not converge within max_iter (its first control was applied all the same — the reference applies
 * whatever IPOPT returns, :338-339), bit 1 = at least one tick's solve reported crx_mpc_solve status bit 1 (it started from a
 * speed outside [min_speed, max_speed]). */
size_t crx_mpc_closed_loop_work_bytes(int n, int T);   /* 0 since 0.2: no work buffer */
int crx_mpc_closed_loop_batch_dev(int n, int T, float* state, const crx_course* course, float dl, int nsearch,
                                  const crx_mpc_params* prm, const crx_loop_params* loop, int* target_ind, float* traj_hist,
                                  int* ticks_done, void* work /* ignored */, void* stream);
int crx_mpc_closed_loop_flags_batch_dev(int n, int T, float* state, const crx_course* course, float dl, int nsearch,
                                        const crx_mpc_params* prm, const crx_loop_params* loop, int* target_ind, float* traj_hist,
                                        int* ticks_done, int* solve_flags, void* stream);
/* host pointers (target_ind may be NULL = start from 0 as mpc_simulation does, :357) */
int crx_mpc_closed_loop_batch(int n, int T, float* state, const crx_course* course, float dl, int nsearch, const crx_mpc_params* prm,
                              const crx_loop_params* loop, int* target_ind, float* traj_hist, int* ticks_done, int* solve_flags);


/* ---- particle-filter localisation (src/particle_filter.cpp) ---------------------------------
 * pf_localization (:73-109) + resampling (:120-148) for n vehicles, T fused ticks, NP particles each (NP <= 128; the
 * reference's `#define NP 100`).  One vehicle per wavefront.  The random numbers the reference draws from std::mt19937
 * inside these functions are inputs here.  Device pointers:
 *   px [n][NP][4] in/out (Eigen::Matrix<float,4,NP> column-major), pw [n][NP] in/out, xEst [n][4] out, PEst [n][16] out,
 *   obs [T][n][L][3] = (noisy range, landmark x, landmark y) with nobs [T][n] <= L valid rows (:258-267),
 *   u [T][n][2], nrm [T][n][NP][2] standard normals (:87-88), uni [T][n][NP] uniforms in [1,2) (:133, uni_d{1.0,2.0} :242),
 *   x_hist [T][n][4] (may be NULL), n_resampled [n] (may be NULL; incremented by the number of resampling ticks).
 * Parity: bit-exact against the CPU oracle evaluated with the engine's summation order, statistical against index-order sums
 * (the reference's own order is Eigen's vectorised redux / gemv): see pf_kernels.hip.h. */
typedef struct crx_pf_params {
  float rsim0, rsim1;   /* Rsim(0,0) = 1.0, Rsim(1,1) = (30 deg)^2 with PI 3.141592653   :229-230 */
  float Q;              /* 0.01   :219 */
  double dt;            /* DT 0.1 :18  */
  float nth;            /* NTh = NP/2  :22 ; <= 0 selects NP/2 */
} crx_pf_params;
void crx_pf_default_params(crx_pf_params* p);
int crx_pf_run_batch_dev(int n, int np, int T, int L, float* px, float* pw, float* xEst, float* PEst, const float* obs,
                         const int* nobs, const float* u, const float* nrm, const float* uni, const crx_pf_params* prm,
                         float* x_hist, int* n_resampled, void* stream);


/* ---- dynamic-window planner (src/dynamic_window_approach.cpp) -----------------------------------
 * The reference's main loop (:192-194, goal test :221) for n agents, max_ticks control steps at most, ONE AGENT PER
 * WAVEFRONT: per tick dwa_control (calc_dynamic_window, every (v, yawrate) sample rolled out and scored, the reference's
 * winner picked) -> motion -> goal test.  max_ticks = 1 is a single dwa_control + motion.
 * state [n][5] = (x, y, yaw, v, yawrate) in/out; u [n][2] in/out; goal [n][2]; ob [nob][2] shared (nob <= 256);
 * traj_hist (may be NULL) [max_ticks][n][5]; ticks_done, status (bit 0: the window held more samples than the kernel's
 * grids, 64 speeds x 1024 yaw rates, and was clipped), best_idx / n_samples of the LAST executed tick (may be NULL). */
typedef struct crx_dwa_config {   /* Config :25-41 */
  float max_speed, min_speed, max_yawrate, max_accel, robot_radius, max_dyawrate, v_reso, yawrate_reso, dt, predict_time,
      to_goal_cost_gain, speed_cost_gain;
} crx_dwa_config;
void crx_dwa_default_config(crx_dwa_config* c);
int crx_dwa_run_batch_dev(int n, int max_ticks, float* state, float* u, const float* goal, const float* ob, int nob,
                          const crx_dwa_config* cfg, float* traj_hist, int* ticks_done, int* status, int* best_idx,
                          int* n_samples, void* stream);

/* ---- Frenet optimal-trajectory planner (src/frenet_optimal_trajectory.cpp) --------------------------
 * The reference's main loop (:224-236) for n agents sharing one course and one obstacle set, ONE AGENT PER WAVEFRONT: per
 * tick frenet_optimal_planning (calc_frenet_paths :51-100 -> calc_global_paths :102-136 -> check_paths :150-158 -> the
 * cheapest survivor :167-174), the winner's second sample handed over as the new state, the goal test (:232).
 * max_ticks = 1 is a single planning call.  Tolerance parity (1e-5), see DESIGN.md 5e.
 *   coef  [9][nx]   the course's Spline2D (include/cubic_spline.h:130-178) as a coefficient table: rows s, then a,b,c,d of
 *                   sx, then a,b,c,d of sy (b and d have nx-1 entries, the last column is padding); device pointer for
 *                   the _dev call; built on the host by crx_frenet_spline_build (2 <= nx <= 64)
 *   state [n][5]    (s0, c_speed, c_d, c_d_d, c_d_dd) in/out          ob [nob][2] shared, nob <= 128
 *   goal            r_x.back(), r_y.back() of main :205-213, see crx_frenet_course_samples
 *   hist            (may be NULL) [max_ticks][n][8] = (s0, c_speed, c_d, c_d_d, c_d_dd, x, y, cf) after each tick
 *   status          bit 0: no candidate survived check_paths (the reference would index an empty path) — the episode ends
 *                   with the state unchanged; bit 2: a candidate started before the course (Spline::calc would throw)
 *   best_idx / n_valid  winner (generation order: di outer, Ti, tv inner) and survivor count of the LAST executed tick
 *   path_cf / path_ok   (may be NULL) [n][path_cap] every candidate's cost and check_paths verdict in the last tick */
typedef struct crx_frenet_config {   /* the #defines :20-38, as the double expressions they expand to */
  double max_speed, max_accel, max_curvature, max_road_width, d_road_w, dt, maxt, mint, target_speed, d_t_s;
  int n_s_sample;
  double robot_radius, kj, kt, kd, klat, klon;
} crx_frenet_config;
void crx_frenet_default_config(crx_frenet_config* c);
/* number of candidate paths the configuration generates, or a negative crx error if it exceeds the kernel's grids
 * (<= 64 lateral offsets, horizons x target speeds <= 64, <= 64 time steps, horizons x speeds x time steps <= 2048) */
int crx_frenet_num_paths(const crx_frenet_config* cfg);
/* host: Spline2D(wx, wy) -> coef[9][nx] (include/cubic_spline.h:53-65, :95-116, :172-186).  The nx-by-nx float system is
 * solved as the reference solves it, A.colPivHouseholderQr().solve(B) in float: csrc/crx_qr.h restates Eigen 3.3.9's
 * ColPivHouseholderQR with ascending loops.  That is Eigen's own order for the 2x2 / 3x3 polynomial systems (every reduction is
 * shorter than a SIMD packet) and bit-identical to the Eigen STAND-IN of the reference-line build for any nx; against real Eigen,
 * whose GEMV kernels may associate the nx x nx products differently, last-ulp differences are possible for nx >= 4 (unpinned). */
int crx_frenet_spline_build(const float* wx, const float* wy, int nx, float* coef);
/* host: the course sampled as main :205-213 does (float i += 0.1); returns the sample count, fills up to cap of them */
int crx_frenet_course_samples(const float* coef, int nx, float* rx, float* ry, int cap);
/* host: the course the reference's tracking mains build from their way-points — Spline2D(wx, wy) (include/cubic_spline.h:130-178,
 * the nx x nx system by colPivHouseholderQr in float) sampled every ds: src/lqr_speed_steer_control.cpp:252-265 (ds = 0.1),
 * src/model_predictive_control.cpp:473-486 (ds = 1.0).  Returns the sample count; fills up to cap of each non-NULL array. */
int crx_course_from_waypoints(const float* wx, const float* wy, int nx, double ds, float* cx, float* cy, float* cyaw, float* ck, int cap);
/* host: calc_speed_profile, variant 5 = src/lqr_speed_steer_control.cpp:40-62, variant 4 = src/lqr_steer_control.cpp:35-52,
 * variant 0 = src/model_predictive_control.cpp:83-105; the two out-of-bounds writes of the reference (5-state file :55-56 with
 * k = 0, MPC file :102 `speed_profile[-1]`) are not made. */
int crx_calc_speed_profile(int variant, const float* rx, const float* ry, const float* ryaw, int n, float target_speed, float* sp);
/* host, in place: smooth_yaw, src/model_predictive_control.cpp:172-185 — what mpc_simulation (:360) does to the course headings
 * before its loop: a step of more than pi/2 between consecutive samples is unwound by 2 pi (float -= double, as the reference).
 * n < 2 is a no-op (the reference's unsigned size()-1 would wrap).  A heading the float walk cannot unwind (non-finite, or so
 * large that -2 pi does not change it; the reference would not return) is refused with CRX_ERR_INVALID. */
int crx_smooth_yaw(float* cyaw, int n);
int crx_frenet_run_batch_dev(int n, int max_ticks, float* state, const float* coef, int nx, const float* goal_xy,
                             const float* ob, int nob, const crx_frenet_config* cfg, float* hist, int* ticks_done,
                             int* status, int* best_idx, int* n_valid, float* path_cf, int* path_ok, int path_cap,
                             void* stream);


/* ---- a mixed EKF + MPC swarm round and the multi-GPU gather, from C (BASELINE.json configs[4]) -------
 * What cpprobotics_amd/swarm.py does from Python, behind the C boundary, so that a C++ host of the reference's loops
 * (src/extended_kalman_filter.cpp:171-188, src/model_predictive_control.cpp:371-385) can run the swarm and concatenate the
 * per-GPU results without Python or torch.
 *
 * crx_hw_queues(): the hardware queues the HIP runtime will multiplex this process's streams onto (GPU_MAX_HW_QUEUES, default 4;
 * read by the runtime when it first touches the device).  Streams that share a queue run their kernels one after the other: a round
 * pipelined `depth` deep wants depth + 2 (measured: 0.74 ms per round on 4 queues, 0.49 on 8+, profiles/r05/swarm_hw_queues.txt).
 * Export the variable before the first HIP call of the process. */
int crx_hw_queues(void);

/* One rank's shard of the swarm: n vehicles run T fused EKF steps per round from (x0, P0); every plan_every-th vehicle then plans
 * from its final estimate at the commanded speed (full-scan calc_nearest_index, calc_ref_trajectory, mpc_solve over Tm knots).
 * The planners of `depth` consecutive rounds are in flight together, each on its own slot (stream, events, buffers), overlapping the
 * EKF launches of the following rounds.  All pointers are device pointers of the current device; the object copies what it needs at
 * creation (x0, P0, Q, R, the course struct — NOT the course arrays, which must outlive it). */
typedef struct crx_swarm crx_swarm;
typedef struct crx_swarm_config {
  int n;                  /* vehicles of this shard */
  int T;                  /* EKF steps per round */
  int Tm;                 /* MPC knots (crx_mpc_solve_batch_dev's T; 21 in configs[4]) */
  int plan_every;         /* every plan_every-th vehicle plans (8) */
  int depth;              /* planner slots in flight, each with a stream, events and buffers of its own (1 .. 12: every queue the
                             solver has run on keeps a private-memory reservation) */
  float v_cmd;            /* the commanded speed the planners start from (the filter's 4th state is a random walk) */
  float dl;               /* calc_ref_trajectory: course tick (1.0) */
  double dt_ref;          /* calc_ref_trajectory: DT (0.2) */
  int nsearch;            /* calc_ref_trajectory: N_IND_SEARCH (10) */
  int allow_shared_queues;/* 0: creation fails if depth + 1 streams exceed crx_hw_queues(); 1: accept the serialisation */
  void* const* planner_streams; /* NULL: the object creates (and owns) its `depth` slot streams; else `depth` streams of the caller's —
                                   a process that already runs the solver on streams of its own hands those in: every distinct stream
                                   the private-memory solver has run on counts against the 12 the library admits */
  crx_ekf_params ekf;
  crx_mpc_params mpc;     /* shared_gpu is set by the library when depth > 1 */
} crx_swarm_config;
void crx_swarm_default_config(crx_swarm_config* c);
int crx_swarm_create(crx_swarm** out, const crx_swarm_config* cfg, const crx_course* course_dev, const float* x0_dev,
                     const float* P0_dev, const float* Q, const float* R /* host: 16 + 4 floats, column-major */);
/* Issue one round on `stream` (nothing is waited for): reset (x, P) to (x0, P0), T EKF steps over z, u ([T][n][2]), xEst history
 * into x_hist ([T][n][4], may be NULL), then — on the round's slot stream — the planners.  Returns the round's index through
 * round_out (may be NULL).  A slot's buffers are reused `depth` rounds later; the launch stream waits for that round's planners
 * by event, the host never blocks. */
int crx_swarm_round_dev(crx_swarm* s, const float* z_dev, const float* u_dev, float* x_hist_dev, void* stream, long long* round_out);
/* The buffers of round `round` (one of the last `depth` rounds): device pointers, valid to READ after crx_swarm_wait.  Any of the
 * out-pointers may be NULL.  sol [n_plan][4 Tm + 2 (Tm - 1)], status [n_plan], cost [n_plan], xref [n_plan][4 Tm], est [n_plan][4]. */
int crx_swarm_plans(crx_swarm* s, long long round, int* n_plan, const float** sol, const int** status, const double** cost,
                    const float** xref, const float** est);
const float* crx_swarm_state(crx_swarm* s);            /* [n][4]: the filter state after the most recent round's EKF launch */
int crx_swarm_wait(crx_swarm* s, void* stream);         /* make `stream` wait for every planner in flight (events; the host does not block) */
int crx_swarm_destroy(crx_swarm* s);                    /* synchronises the slot streams first */

/* The trajectory / final-state concat over GPUs (the one collective the path has): RCCL's all-gather over xGMI,
 * one process (or thread) per GPU.  librccl is loaded on the first of these calls (no link-time dependency).
 *   rank 0: crx_comm_unique_id(id) -> hand the CRX_COMM_ID_BYTES bytes to every rank (MPI, a socket, a file: examples/ekf_fleet_mgpu.cpp)
 *   every rank: crx_comm_init_rank(&comm, id, rank, world) on its device, then crx_allgather_dev(comm, send, recv, bytes, stream):
 *   recv[r * bytes .. (r + 1) * bytes) = rank r's send buffer — for contiguous equal agent shards, the swarm in global agent order. */
#define CRX_COMM_ID_BYTES 128
typedef struct crx_comm crx_comm;
int crx_comm_unique_id(void* id_out /* CRX_COMM_ID_BYTES */);
int crx_comm_init_rank(crx_comm** out, const void* id, int rank, int world);
int crx_comm_rank(const crx_comm* c);
int crx_comm_world(const crx_comm* c);
int crx_allgather_dev(crx_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int crx_comm_destroy(crx_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* CRX_H_ */
