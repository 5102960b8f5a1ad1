// api_core.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); version / init / shutdown, device selection and the device set, workspaces, pinned host memory, the host libm check,
// default parameter structs.
extern "C" {

int crx_version(void) { return 500; }  // 0.5.0 (0.4.0 + the MPC tile layout, crx_swarm_round_dev, crx_comm_* / crx_allgather_dev)

// crx_init only checks that a device is there and forces the HIP runtime + code object to load now rather than in the first
// timed call; crx_shutdown drains the devices and gives the host-pointer workspaces back.  Both are optional.  The only state the
// engine keeps is on the host-pointer side (crx_host.h): per-device grow-only workspaces and the device set.
int crx_init(void) {
  if (int rc = check_device()) return rc;
  CRX_HIP(hipFree(nullptr));
  return CRX_OK;
}
// per-device work of crx_release_workspace / crx_shutdown; the caller restores the thread's current device whatever this returns
static int release_device(int d, bool handles_too) {
  crxh::DeviceCtx& c = crxh::ctx_table()[d];
  std::lock_guard<std::mutex> l(c.mu);
  if (!c.dws.p && !c.pws.p && !(handles_too && c.has_handles())) return CRX_OK;
  CRX_HIP(hipSetDevice(d));
  CRX_HIP(hipDeviceSynchronize());
  c.release_workspace();
  if (handles_too) c.destroy_handles();
  return CRX_OK;
}
static int release_all(bool handles_too) {
  const int nd = crx_device_count();
  int cur = 0;
  if (nd == 0) return CRX_OK;
  CRX_HIP(hipGetDevice(&cur));
  int rc = CRX_OK;
  for (int d = 0; d < nd && d < crxh::kMaxDevices && rc == CRX_OK; ++d) rc = release_device(d, handles_too);
  const hipError_t e = hipSetDevice(cur);          // on the failure paths too: the caller's later _dev calls must not land on another GPU
  if (rc == CRX_OK && e != hipSuccess) return hip_fail(e, "hipSetDevice (restore)");
  return rc;
}
int crx_release_workspace(void) { return release_all(false); }
// Grow the current device's workspaces ahead of time (a latency-sensitive host calls this once at start-up with the sizes of its
// largest call, so that no call pays for the growth: hipMalloc / hipHostMalloc of hundreds of MB take tens of milliseconds).
int crx_reserve_workspace(size_t device_bytes, size_t pinned_bytes) {
  CRX_TRACE();
  crxh::DeviceCtx* c = nullptr;
  std::unique_lock<std::mutex> lock;
  if (int rc = ctx_open(&c, lock)) return rc;
  hipError_t e = c->dws.reserve(device_bytes);
  if (e == hipSuccess) e = c->pws.reserve(pinned_bytes);
  if (e != hipSuccess) { hip_fail(e, "reserve_workspace"); return CRX_ERR_ALLOC; }
  return CRX_OK;
}
int crx_shutdown(void) {
  if (crx_device_count() == 0) return CRX_OK;
  if (int rc = release_all(true)) return rc;       // workspaces, and the contexts' streams and events
  CRX_HIP(hipDeviceSynchronize());
  return CRX_OK;
}

int crx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

// The device of the calling thread: what the `_dev` entry points launch on (their pointers and stream must belong to it) and what
// a host-pointer call uses when no device set is installed.  Thin wrappers over hipSetDevice / hipGetDevice so that a C++ host
// needs no HIP header for device selection.
int crx_set_device(int device) {
  if (int rc = check_device()) return rc;
  if (device < 0 || device >= crx_device_count()) return fail(CRX_ERR_INVALID, "set_device: no such device");
  CRX_HIP(hipSetDevice(device));
  return CRX_OK;
}
int crx_get_device(void) {
  int d = 0;
  if (crx_device_count() == 0 || hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return d;
}
// The device set of the host-pointer BATCH entry points (process-wide): with ndev >= 1 they split [0, n) contiguously over
// devices[0..ndev) — shard r on devices[r], the first n % G shards one agent longer, as few shards as keep min_agents_per_device
// agents each — one host thread per shard, results straight into the caller's arrays.  ndev = 0 restores the default (the calling
// thread's current device).  A device may be named more than once (its shards then run one after the other: a way to exercise
// the sharding on a single GPU).  devices = NULL with ndev > 0 means devices 0 .. ndev-1.
int crx_set_devices(const int* devices, int ndev, int min_agents_per_device) {
  if (ndev < 0 || ndev > 1024) return fail(CRX_ERR_INVALID, "set_devices: bad device count");
  std::vector<int> v;
  if (ndev > 0) {
    if (int rc = check_device()) return rc;
    const int have = crx_device_count();
    for (int i = 0; i < ndev; ++i) {
      const int d = devices ? devices[i] : i;
      if (d < 0 || d >= have || d >= crxh::kMaxDevices) return fail(CRX_ERR_INVALID, "set_devices: no such device");
      v.push_back(d);
    }
  }
  crxh::DeviceSet& s = crxh::device_set();
  std::lock_guard<std::mutex> l(s.m);
  s.devs = v;
  s.min_agents = min_agents_per_device < 0 ? 0 : min_agents_per_device;
  return CRX_OK;
}
int crx_get_devices(int* devices, int cap) {
  crxh::DeviceSet& s = crxh::device_set();
  std::lock_guard<std::mutex> l(s.m);
  for (int i = 0; i < (int)s.devs.size() && i < cap && devices; ++i) devices[i] = s.devs[i];
  return (int)s.devs.size();
}
// Do the libm functions of THIS host return the bits the engine's restatements return (crx_trig.h, crx_fdlibm.h, crx_dsincos.h,
// crx_datan2.h: glibc 2.35, x86-64 FMA build)?  Bit parity with a reference built on this host holds only if they do — the
// reference calls the host's libm, the kernels carry the restatements.  200,000 pseudo-random arguments per family (a few ms):
// 0 = all equal; bit 0: sinf / cosf, bit 1: expf, bit 2: atanf / atan2f / tanf / acosf, bit 3: double sin / cos (arguments of the
// form (double)f + pi/2, the Frenet planner's), bit 4: double atan2(y, 1.0).  Needs no device.
int crx_host_libm_check(void) {
  unsigned long long st = 0x9e3779b97f4a7c15ull;
  auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  auto same32 = [](float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (a != a && b != b); };
  auto same64 = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a != a && b != b); };
  int bad = 0;
  for (int i = 0; i < 200000; ++i) {
    const unsigned long long r = next();
    // a float spread over the magnitudes the path sees: |x| from 2^-20 to 2^12, either sign
    const float mag = std::ldexp(1.0f + (float)((r >> 8) & 0x7fffff) / 8388608.0f, (int)(r % 33) - 20);
    const float x = (r >> 63) ? -mag : mag;
    const float unit = (float)((double)((r >> 20) & 0xffffff) / 8388608.0 - 1.0);             // [-1, 1)
    if (!same32(::sinf(x), crx::sinf_(x)) || !same32(::cosf(x), crx::cosf_(x))) bad |= 1;
    if (!same32(::expf(-0.5f * mag), crx::expf_(-0.5f * mag))) bad |= 2;
    if (!same32(::atanf(x), crx::atanf_(x)) || !same32(::atan2f(x, unit), crx::atan2f_(x, unit)) || !same32(::tanf(x), crx::tanf_(x)) ||
        !same32(::acosf(unit), crx::acosf_(unit))) bad |= 4;
    const float yaw = 3.2f * unit;
    const double xd = (double)yaw + M_PI / 2.0;
    if (!same64(::sin(xd), crx::dsin_(xd)) || !same64(::cos(xd), crx::dcos_(xd))) bad |= 8;
    const double y = 0.5 * (double)x;
    if (!same64(::atan2(y, (double)1.0), crx::datan2_one_(y))) bad |= 16;
  }
  return bad;
}

// Pinned (page-locked, device-visible) host memory: arrays allocated here cross PCIe by DMA straight from / into the caller's
// memory, without the staging copy pageable memory needs.
void* crx_host_alloc(size_t bytes) {
  if (check_device()) return nullptr;
  void* p = nullptr;
  const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped);
  if (e != hipSuccess) { hip_fail(e, "hipHostMalloc"); return nullptr; }
  return p;
}
void crx_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

const char* crx_last_error(void) { return g_err.c_str(); }

void crx_ekf_default_params(crx_ekf_params* p) { if (p) p->dt = 0.1; }

void crx_lqr_default_params(crx_lqr_params* p) {
  if (!p) return;
  p->dt = 0.1; p->L = 0.5; p->eps = 0.01f; p->maxiter = 150;
}

void crx_mpc_default_params(crx_mpc_params* p) {
  if (!p) return;
  p->dt = 0.2; p->wb = 2.5;
  p->max_steer = 45.0 / 180 * 3.14159265358979323846;
  p->max_accel = 1.0;
  p->max_speed = 55.0 / 3.6; p->min_speed = -20.0 / 3.6;
  p->r_a = 0.01; p->r_delta = 0.01; p->rd_a = 0.01; p->rd_delta = 1.0;
  p->q_x = 1.0; p->q_y = 1.0; p->q_yaw = 0.5; p->q_v = 0.5;
  p->tol = 1e-9; p->max_iter = 50; p->shared_gpu = 0;
}
}  // extern "C"
