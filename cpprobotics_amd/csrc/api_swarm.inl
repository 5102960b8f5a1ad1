// api_swarm.inl — part of the single translation unit crx_api.hip (#included last): one round of a mixed EKF + MPC swarm issued from
// C (crx_swarm_*: BASELINE.json configs[4]; the bodies of /root/reference/src/extended_kalman_filter.cpp:171-188 and
// /root/reference/src/model_predictive_control.cpp:371-385 for a shard of vehicles) and the multi-GPU concat (crx_comm_*,
// crx_allgather_dev: RCCL's all-gather).  cpprobotics_amd/swarm.py is the Python twin of the round
// (MixedSwarmRound + SwarmShard); tests/test_swarm_gpu.py demands the same bytes from both.
namespace crx {
// est[j] = (x, y, yaw of vehicle j * every | v_cmd): the planners start from the estimated pose at the commanded speed
__global__ void __launch_bounds__(256) swarm_pick_kernel(int n_plan, int every, float v_cmd, const float4* __restrict__ x, float4* __restrict__ est) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_plan) return;
  float4 s = x[(size_t)j * every];
  s.w = v_cmd;
  est[j] = s;
}
}  // namespace crx

struct crx_swarm {
  int device = 0;
  crx_swarm_config cfg{};
  crx_course course{};
  float Q[16], R[4];
  int n_plan = 0;
  long long round = 0;
  float *x = nullptr, *P = nullptr, *x0 = nullptr, *P0 = nullptr;
  struct Slot {
    float *est = nullptr, *e = nullptr, *xref = nullptr, *sol = nullptr;
    int *tind = nullptr, *status = nullptr;
    double* cost = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ekf_done = nullptr, plan_done = nullptr;
    bool used = false, own_stream = false;
  };
  std::vector<Slot> slots;
};

namespace {
int hw_queues_env() {
  const char* v = getenv("GPU_MAX_HW_QUEUES");
  if (!v || !*v) return 4;
  char* end = nullptr;
  const long q = strtol(v, &end, 10);
  return (end == v || q < 1) ? 4 : (int)q;
}
void swarm_free(crx_swarm* s) {
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  for (auto& sl : s->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    fr(sl.est); fr(sl.e); fr(sl.xref); fr(sl.sol); fr(sl.tind); fr(sl.status); fr(sl.cost);
    if (sl.ekf_done) (void)hipEventDestroy(sl.ekf_done);
    if (sl.plan_done) (void)hipEventDestroy(sl.plan_done);
    if (sl.stream && sl.own_stream) (void)hipStreamDestroy(sl.stream);
  }
  fr(s->x); fr(s->P); fr(s->x0); fr(s->P0);
  delete s;
}
}  // namespace

extern "C" {

int crx_hw_queues(void) { return hw_queues_env(); }

void crx_swarm_default_config(crx_swarm_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->n = 0; c->T = 100; c->Tm = 21; c->plan_every = 8; c->depth = 6;
  c->v_cmd = 2.5f; c->dl = 1.0f; c->dt_ref = 0.2; c->nsearch = 10; c->allow_shared_queues = 0;
  crx_ekf_default_params(&c->ekf);
  crx_mpc_default_params(&c->mpc);
}

int crx_swarm_create(crx_swarm** out, const crx_swarm_config* cfg, const crx_course* course, const float* x0, const float* P0,
                     const float* Q, const float* R) {
  CRX_TRACE();
  if (!out || !cfg || !course || !x0 || !P0 || !Q || !R) return fail(CRX_ERR_INVALID, "swarm_create: null argument");
  *out = nullptr;
  if (cfg->n < 1 || cfg->T < 1 || cfg->Tm < 2 || cfg->Tm > CRX_MPC_MAX_T || cfg->plan_every < 1 || cfg->depth < 1 || cfg->depth > kMaxPrivateMemoryStreams)
    return fail(CRX_ERR_INVALID, "swarm_create: n >= 1, T >= 1, 2 <= Tm <= 64, plan_every >= 1, 1 <= depth <= 12 (every queue the solver has "
                                 "run on keeps a private-memory reservation: INTEGRATION.md 7)");
  if (!course_ok(course, false)) return fail(CRX_ERR_INVALID, "swarm_create: bad course");
  if (int rc = check_device()) return rc;
  if (!cfg->allow_shared_queues && cfg->depth + 1 > hw_queues_env()) {
    g_err = "swarm_create: " + std::to_string(cfg->depth) + " planner streams + the launch stream on " + std::to_string(hw_queues_env()) +
            " hardware queues (GPU_MAX_HW_QUEUES; streams that share a queue run one after the other: 0.74 instead of 0.49 ms per round at "
            "depth 6) — export GPU_MAX_HW_QUEUES >= " + std::to_string(cfg->depth + 2) + " before the process first touches the GPU, or set "
            "allow_shared_queues";
    return CRX_ERR_INVALID;
  }
  crx_swarm* s = new crx_swarm;
  s->cfg = *cfg;
  s->cfg.planner_streams = nullptr;            // (the slots keep the handles; the caller's array need not outlive this call)
  s->cfg.mpc.shared_gpu = cfg->depth > 1 ? 1 : cfg->mpc.shared_gpu;
  s->course = *course;
  memcpy(s->Q, Q, sizeof(s->Q)); memcpy(s->R, R, sizeof(s->R));
  s->n_plan = (cfg->n + cfg->plan_every - 1) / cfg->plan_every;
  hipError_t e = hipGetDevice(&s->device);
  const size_t n = (size_t)cfg->n, np = (size_t)s->n_plan, nv = 4 * (size_t)cfg->Tm + 2 * ((size_t)cfg->Tm - 1);
  auto al = [&](auto** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc((void**)p, bytes); };
  al(&s->x, 16 * n); al(&s->P, 64 * n); al(&s->x0, 16 * n); al(&s->P0, 64 * n);
  if (e == hipSuccess) e = hipMemcpy(s->x0, x0, 16 * n, hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipMemcpy(s->P0, P0, 64 * n, hipMemcpyDeviceToDevice);
  if (e == hipSuccess) e = hipMemcpy(s->x, x0, 16 * n, hipMemcpyDeviceToDevice);
  s->slots.resize(cfg->depth);
  for (auto& sl : s->slots) {
    al(&sl.est, 16 * np); al(&sl.e, 4 * np); al(&sl.xref, 16 * (size_t)cfg->Tm * np); al(&sl.sol, 4 * nv * np);
    al(&sl.tind, 4 * np); al(&sl.status, 4 * np); al(&sl.cost, 8 * np);
    if (e == hipSuccess) e = hipMemset(sl.tind, 0, 4 * np);
    const size_t k = (size_t)(&sl - s->slots.data());
    if (cfg->planner_streams) sl.stream = (hipStream_t)cfg->planner_streams[k];
    else if (e == hipSuccess) { e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking); sl.own_stream = e == hipSuccess; }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ekf_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.plan_done, hipEventDisableTiming);
  }
  if (e != hipSuccess) { swarm_free(s); hip_fail(e, "swarm_create"); return CRX_ERR_ALLOC; }
  *out = s;
  return CRX_OK;
}

int crx_swarm_round_dev(crx_swarm* s, const float* z, const float* u, float* x_hist, void* stream, long long* round_out) {
  CRX_TRACE();
  if (!s || !z || !u) return fail(CRX_ERR_INVALID, "swarm_round: null argument");
  const crx_swarm_config& c = s->cfg;
  hipStream_t main = (hipStream_t)stream;
  const size_t n = (size_t)c.n;
  crx_swarm::Slot& sl = s->slots[(size_t)(s->round % c.depth)];
  // 1. the shard's vehicles: T fused EKF steps from the start state
  CRX_HIP(hipMemcpyAsync(s->x, s->x0, 16 * n, hipMemcpyDeviceToDevice, main));
  CRX_HIP(hipMemcpyAsync(s->P, s->P0, 64 * n, hipMemcpyDeviceToDevice, main));
  CRX_TRY(crx_ekf_run_batch_dev(c.n, c.T, s->x, s->P, z, u, x_hist, nullptr, s->Q, s->R, &c.ekf, main));
  // 2. every plan_every-th vehicle plans from its final estimate, on the slot's stream; the slot's buffers are free once the planners
  //    of round - depth are done
  if (sl.used) CRX_HIP(hipStreamWaitEvent(main, sl.plan_done, 0));
  hipLaunchKernelGGL(crx::swarm_pick_kernel, dim3(blocks_for(s->n_plan, 256)), dim3(256), 0, main, s->n_plan, c.plan_every, c.v_cmd,
                     reinterpret_cast<const float4*>(s->x), reinterpret_cast<float4*>(sl.est));
  CRX_HIP(hipGetLastError());
  CRX_HIP(hipEventRecord(sl.ekf_done, main));
  CRX_HIP(hipStreamWaitEvent(sl.stream, sl.ekf_done, 0));
  CRX_TRY(crx_calc_nearest_index_batch_dev(s->n_plan, sl.est, &s->course, sl.tind, sl.e, sl.stream));
  CRX_TRY(crx_calc_ref_trajectory_batch_dev(s->n_plan, c.Tm, sl.est, &s->course, c.dl, c.dt_ref, c.nsearch, sl.tind, sl.xref, sl.stream));
  CRX_TRY(crx_mpc_solve_batch_dev(s->n_plan, c.Tm, sl.est, sl.xref, &c.mpc, sl.sol, sl.status, sl.cost, sl.stream));
  CRX_HIP(hipEventRecord(sl.plan_done, sl.stream));
  sl.used = true;
  if (round_out) *round_out = s->round;
  s->round++;
  return CRX_OK;
}

int crx_swarm_plans(crx_swarm* s, long long round, int* n_plan, const float** sol, const int** status, const double** cost,
                    const float** xref, const float** est) {
  if (!s) return fail(CRX_ERR_INVALID, "swarm_plans: null argument");
  if (round < 0 || round >= s->round || round < s->round - s->cfg.depth) return fail(CRX_ERR_INVALID, "swarm_plans: that round's slot has been reused (or the round has not been issued)");
  const crx_swarm::Slot& sl = s->slots[(size_t)(round % s->cfg.depth)];
  if (n_plan) *n_plan = s->n_plan;
  if (sol) *sol = sl.sol;
  if (status) *status = sl.status;
  if (cost) *cost = sl.cost;
  if (xref) *xref = sl.xref;
  if (est) *est = sl.est;
  return CRX_OK;
}

const float* crx_swarm_state(crx_swarm* s) { return s ? s->x : nullptr; }

int crx_swarm_wait(crx_swarm* s, void* stream) {
  if (!s) return fail(CRX_ERR_INVALID, "swarm_wait: null argument");
  for (auto& sl : s->slots)
    if (sl.used) CRX_HIP(hipStreamWaitEvent((hipStream_t)stream, sl.plan_done, 0));
  return CRX_OK;
}

int crx_swarm_destroy(crx_swarm* s) {
  if (!s) return CRX_OK;
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  if (have && cur != s->device) (void)hipSetDevice(s->device);
  swarm_free(s);
  if (have && cur != s->device) (void)hipSetDevice(cur);
  return CRX_OK;
}

}  // extern "C"

// ---- the multi-GPU gather: RCCL, looked up at run time ------------------------------------------------------------------------
namespace {
struct RcclId { char internal[CRX_COMM_ID_BYTES]; };          // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
struct Rccl {
  int (*get_unique_id)(RcclId*) = nullptr;
  int (*comm_init_rank)(void**, int, RcclId, int) = nullptr;   // the id travels BY VALUE (rccl.h: ncclCommInitRank)
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  bool ok = false;
  Rccl() {
    void* h = dlopen("librccl.so.1", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) return;
    get_unique_id = reinterpret_cast<decltype(get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    comm_init_rank = reinterpret_cast<decltype(comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    all_gather = reinterpret_cast<decltype(all_gather)>(dlsym(h, "ncclAllGather"));
    comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    error_string = reinterpret_cast<decltype(error_string)>(dlsym(h, "ncclGetErrorString"));
    ok = get_unique_id && comm_init_rank && all_gather && comm_destroy;
  }
};
const Rccl& rccl() { static Rccl r; return r; }
int rccl_fail(int rc, const char* what) {
  g_err = std::string(what) + ": RCCL error " + std::to_string(rc) + (rccl().error_string ? std::string(" (") + rccl().error_string(rc) + ")" : std::string());
  return CRX_ERR_HIP;
}
int rccl_ready() {
  if (!rccl().ok) return fail(CRX_ERR_NO_DEVICE, "crx_comm: librccl.so.1 not found or incomplete (the multi-GPU gather needs RCCL; there is no host fallback)");
  return CRX_OK;
}
}  // namespace

struct crx_comm { void* comm = nullptr; int rank = 0, world = 1, device = 0; };

extern "C" {

int crx_comm_unique_id(void* id_out) {
  CRX_TRACE();
  if (!id_out) return fail(CRX_ERR_INVALID, "comm_unique_id: null argument");
  CRX_TRY(rccl_ready());
  RcclId id;
  if (int rc = rccl().get_unique_id(&id)) return rccl_fail(rc, "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return CRX_OK;
}

int crx_comm_init_rank(crx_comm** out, const void* id, int rank, int world) {
  CRX_TRACE();
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(CRX_ERR_INVALID, "comm_init_rank: bad argument (0 <= rank < world)");
  *out = nullptr;
  if (int rc = check_device()) return rc;
  CRX_TRY(rccl_ready());
  RcclId uid;
  memcpy(&uid, id, sizeof(uid));
  crx_comm* c = new crx_comm;
  c->rank = rank; c->world = world;
  (void)hipGetDevice(&c->device);
  if (int rc = rccl().comm_init_rank(&c->comm, world, uid, rank)) { delete c; return rccl_fail(rc, "ncclCommInitRank"); }
  *out = c;
  return CRX_OK;
}

int crx_comm_rank(const crx_comm* c) { return c ? c->rank : -1; }
int crx_comm_world(const crx_comm* c) { return c ? c->world : 0; }

int crx_allgather_dev(crx_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  CRX_TRACE();
  if (!c || (bytes_per_rank && (!send || !recv))) return fail(CRX_ERR_INVALID, "allgather: null argument");
  if (bytes_per_rank == 0) return CRX_OK;
  // as bytes (ncclInt8 = 0): the payload is whatever the caller's rows are — float4 estimates, [T][n][4] histories
  if (int rc = rccl().all_gather(send, recv, bytes_per_rank, 0, c->comm, (hipStream_t)stream)) return rccl_fail(rc, "ncclAllGather");
  return CRX_OK;
}

int crx_comm_destroy(crx_comm* c) {
  if (!c) return CRX_OK;
  int rc = rccl().ok ? rccl().comm_destroy(c->comm) : 0;
  delete c;
  return rc ? rccl_fail(rc, "ncclCommDestroy") : CRX_OK;
}

}  // extern "C"
