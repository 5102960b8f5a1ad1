// crx_api.hip — the C ABI (include/crx.h) over the gfx950 kernels.  Host side is thin on purpose:
// argument checks, launch geometry, and (for the host-pointer variants) staging copies.
// There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <dlfcn.h>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "crx_host.h"
#include "../../include/crx.h"
#include "../../include/crx_experimental.h"
#include "dare_kernels.hip.h"
#include "ekf_kernels.hip.h"
#include "mpc_kernels.hip.h"
#include "track_kernels.hip.h"
#include "pf_kernels.hip.h"
#include "dwa_kernels.hip.h"
#include "frenet_kernels.hip.h"
#include "crx_philox.h"
#include "crx_qr.h"
// Measured-and-rejected kernel variants (A/B evidence: profiles/r02/ekf_wave_ab.txt, profiles/r03/mpc_lanes_ab.txt): compiled in only
// with -DCRX_EXPERIMENTAL_KERNELS=1 (the Makefile's default, so that the A/B scripts and their parity tests can run); reachable
// through include/crx_experimental.h only, never selected by a product entry point.
#ifndef CRX_EXPERIMENTAL_KERNELS
#define CRX_EXPERIMENTAL_KERNELS 0
#endif
#if CRX_EXPERIMENTAL_KERNELS
#include "ekf_wave2_kernels.hip.h"
#include "mpc_quad_kernels.hip.h"
#endif

namespace {

thread_local std::string g_err = "";

// roctx ranges around every entry point, so that rocprofv3 --marker-trace output of an application is self-describing (which
// crx call a kernel belongs to).  The marker library is looked up at run time: no link dependency, a no-op when it is absent.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
const Roctx& roctx() { static Roctx r; return r; }
struct TraceRange {
  bool on;
  explicit TraceRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~TraceRange() { if (on) roctx().pop(); }
};
#define CRX_TRACE() TraceRange crx_trace_range__(__func__)

int fail(int code, const char* what) { g_err = what; return code; }
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return CRX_ERR_HIP;
}

#define CRX_HIP(call)                                        \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) return hip_fail(e__, #call);      \
  } while (0)

int check_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(CRX_ERR_NO_DEVICE, "no HIP device available (crx has no CPU fallback)");
  }
  return CRX_OK;
}

inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// Largest batch the four-lanes-per-agent Riccati kernel is selected for (measured crossover: profiles/r03/dare_lanes_ab.txt).
constexpr int kDareQuadMaxAgents = 32768;
constexpr int kDareDenseQuadMaxAgents = 32768;   // dense kernels: a quad per agent up to here (measured crossover: profiles/r04/dare_dense_lanes_ab.jsonl: 1.9-2.2x at 16,384, 1.1-1.6x at 32,768, 0.5-0.9x at 65,536)
constexpr int kDareRefillMinAgents = 262144;  // one lane per agent, lanes refilled (dare_from_v_refill_kernel) above this: 1.04x there, 1.55x at 1 M, 1.70x at 4 M agents (profiles/r04/dare_refill_ab.jsonl)
constexpr int kDareRefillHold = 16;           // finished lanes a wave collects before it hands their agents back in one pass (8-16 measured best)
// agents per wave of the refilling kernel: two waves per SIMD, 256 .. 1,024 agents (a multiple of 64); 0 = the batch is too small
inline int dare_refill_chunk(int n) {
  if (n <= kDareRefillMinAgents) return 0;
  const int per = ((n / 2048 + 63) / 64) * 64;
  return per < 256 ? 256 : (per > 1024 ? 1024 : per);
}
constexpr int kDareChainMaxAgents = 98304;   // one lane per agent: the unmasked two-evaluations-per-branch loop up to here (profiles/r03/dare_lanes_ab.txt)

// Threads per workgroup of the iterative kernels (dense DARE, tracking): full 64-lane waves in single-wave workgroups.  Narrower
// waves (32..4 active lanes, to shorten the wait for a wave's slowest agent and to occupy idle SIMDs at BASELINE-sized batches)
// were measured in round 1 and are 1.0x-5x SLOWER: the dispatcher stacks the extra waves on a subset of the CUs.
inline unsigned iter_block() { return 64; }

crx::EkfConsts make_consts(const float* Q, const float* R, const crx_ekf_params* prm) {
  crx::EkfConsts k;
  std::memcpy(k.Q, Q, sizeof(k.Q));
  std::memcpy(k.R, R, sizeof(k.R));
  k.dt = prm ? prm->dt : 0.1;
  return k;
}

// ---- host-pointer calls (crx_host.h: contexts, workspaces, copy pool, device set) ---------------------------------------------
// the context of the calling thread's current device, locked for the duration of one host-pointer call
int ctx_open(crxh::DeviceCtx** out, std::unique_lock<std::mutex>& lock) {
  if (int rc = check_device()) return rc;
  int dev = 0;
  CRX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= crxh::kMaxDevices) return fail(CRX_ERR_INVALID, "device ordinal out of range");
  crxh::DeviceCtx& c = crxh::ctx_table()[dev];
  lock = std::unique_lock<std::mutex>(c.mu);
  const hipError_t e = c.init(dev);
  if (e != hipSuccess) return hip_fail(e, "crx host context (streams / events)");
  *out = &c;
  return CRX_OK;
}

// One host-pointer call on the current device: register the arguments, commit() (places them — zero-copy pinned block or device
// workspace — and moves the inputs), run the `_dev` entry point on stream(), finish() (moves the outputs, synchronises).
class HostCall {
 public:
  // an argument: `rows` rows of `row_bytes`; on the host they lie `pitch` bytes apart (a shard's columns of a time-major array),
  // in the workspace densely.  src: copied in (NULL: not); dst: copied back (NULL: not); zero: cleared when there is no src.
  struct Arg { const char* src; char* dst; size_t row_bytes, rows, pitch; bool zero; char* d; };

  int open() { return ctx_open(&c_, lock_); }
  // kernels that come back to their inputs tick after tick (the closed loops read the course from global memory every tick) must
  // not run out of host memory across PCIe
  void forbid_zero_copy() { allow_zc_ = false; }
  int add(const void* src, void* dst, size_t bytes, bool zero = false) { return add2d(src, dst, bytes, 1, bytes, zero); }
  int add2d(const void* src, void* dst, size_t row_bytes, size_t rows, size_t pitch, bool zero = false) {
    args_.push_back(Arg{static_cast<const char*>(src), static_cast<char*>(dst), row_bytes, rows, pitch, zero, nullptr});
    return (int)args_.size() - 1;
  }
  template <class T> T* p(int i) { return reinterpret_cast<T*>(args_[i].d); }
  hipStream_t stream() const { return c_->s_cmp; }
  bool zero_copy() const { return zc_; }

  int commit() {
    size_t total = 0, biggest_row = 0;
    for (auto& a : args_) { total += crxh::align_up(a.row_bytes * a.rows); if (a.rows > 1) biggest_row = std::max(biggest_row, a.row_bytes); }
    zc_ = allow_zc_ && total <= crxh::kZeroCopyBytes;
    slot_ = std::max(crxh::kStageChunk, crxh::align_up(biggest_row));      // a staging slot holds at least one row of every strided argument
    hipError_t e = zc_ ? c_->pws.reserve(total) : c_->dws.reserve(total);
    if (e == hipSuccess && !zc_) e = c_->pws.reserve(2 * slot_);
    if (e != hipSuccess) { hip_fail(e, zc_ ? "hipHostMalloc (pinned workspace)" : "hipMalloc (device workspace)"); return CRX_ERR_ALLOC; }
    char* base = static_cast<char*>(zc_ ? c_->pws.p : c_->dws.p);
    size_t off = 0;
    for (auto& a : args_) { a.d = base + off; off += crxh::align_up(a.row_bytes * a.rows); }
    for (auto& a : args_) {
      const size_t bytes = a.row_bytes * a.rows;
      if (!bytes) continue;
      if (a.src) { if (int rc = copy_in(a)) return rc; }
      else if (a.zero) {
        if (zc_) std::memset(a.d, 0, bytes);
        else CRX_HIP(hipMemsetAsync(a.d, 0, bytes, c_->s_cmp));
      }
    }
    return CRX_OK;
  }
  int finish() {
    if (zc_) {
      CRX_HIP(hipStreamSynchronize(c_->s_cmp));
      for (auto& a : args_)
        if (a.dst) crxh::CopyPool::get().submit(crxh::CopyPool::Job{a.dst, a.d, a.row_bytes, a.rows, a.pitch, a.row_bytes}, nullptr);
      return CRX_OK;
    }
    for (auto& a : args_)
      if (a.dst && a.row_bytes * a.rows) { if (int rc = copy_out(a)) return rc; }
    CRX_HIP(hipStreamSynchronize(c_->s_cmp));
    return CRX_OK;
  }

 private:
  crxh::DeviceCtx* c_ = nullptr;
  std::unique_lock<std::mutex> lock_;
  std::vector<Arg> args_;
  bool zc_ = false, allow_zc_ = true;
  size_t slot_ = crxh::kStageChunk;

  // how a staged copy is cut: `rows == 1` along the bytes, otherwise along the rows
  struct Cut { size_t chunks, unit_rows, unit_bytes; };
  Cut cut(const Arg& a) const {
    if (a.rows == 1) return Cut{(a.row_bytes + slot_ - 1) / slot_, 1, slot_};
    const size_t rpc = std::max<size_t>(1, slot_ / a.row_bytes);
    return Cut{(a.rows + rpc - 1) / rpc, rpc, a.row_bytes};
  }
  int copy_in(const Arg& a) {
    const size_t bytes = a.row_bytes * a.rows;
    if (zc_) {
      crxh::CopyPool::Ticket t;
      crxh::CopyPool::get().submit(crxh::CopyPool::Job{a.d, a.src, a.row_bytes, a.rows, a.row_bytes, a.pitch}, bytes >= (1u << 20) ? &t : nullptr);
      crxh::CopyPool::get().wait(&t);
      return CRX_OK;
    }
    const bool dense = a.rows == 1 || a.pitch == a.row_bytes;
    if (dense && (bytes < (1u << 20) || crxh::is_pinned(a.src))) {      // small, or DMA straight from the caller's pinned memory
      CRX_HIP(hipMemcpyAsync(a.d, a.src, bytes, hipMemcpyHostToDevice, c_->s_cmp));
      return CRX_OK;
    }
    if (!dense && crxh::is_pinned(a.src)) {
      CRX_HIP(hipMemcpy2DAsync(a.d, a.row_bytes, a.src, a.pitch, a.row_bytes, a.rows, hipMemcpyHostToDevice, c_->s_cmp));
      return CRX_OK;
    }
    // pageable: through the pinned ring; the copy threads fill slot k+1 while the DMA of slot k is in flight
    char* pin = static_cast<char*>(c_->pws.p);
    const Cut ct = cut(a);
    for (size_t k = 0; k < ct.chunks; ++k) {
      char* slot = pin + (k & 1) * slot_;
      if (k >= 2) CRX_HIP(hipEventSynchronize(c_->ev_tmp[k & 1]));
      size_t len, doff;
      crxh::CopyPool::Ticket t;
      if (a.rows == 1) {
        doff = k * ct.unit_bytes; len = std::min(ct.unit_bytes, a.row_bytes - doff);
        crxh::CopyPool::get().submit(crxh::CopyPool::Job{slot, a.src + doff, len, 1, len, len}, &t);
      } else {
        const size_t r0 = k * ct.unit_rows, nr = std::min(ct.unit_rows, a.rows - r0);
        doff = r0 * a.row_bytes; len = nr * a.row_bytes;
        crxh::CopyPool::get().submit(crxh::CopyPool::Job{slot, a.src + r0 * a.pitch, a.row_bytes, nr, a.row_bytes, a.pitch}, &t);
      }
      crxh::CopyPool::get().wait(&t);
      CRX_HIP(hipMemcpyAsync(a.d + doff, slot, len, hipMemcpyHostToDevice, c_->s_cmp));
      CRX_HIP(hipEventRecord(c_->ev_tmp[k & 1], c_->s_cmp));
    }
    return CRX_OK;
  }
  int copy_out(const Arg& a) {
    const size_t bytes = a.row_bytes * a.rows;
    const bool dense = a.rows == 1 || a.pitch == a.row_bytes;
    if (dense && (bytes < (1u << 20) || crxh::is_pinned(a.dst))) {
      CRX_HIP(hipMemcpyAsync(a.dst, a.d, bytes, hipMemcpyDeviceToHost, c_->s_cmp));
      return CRX_OK;
    }
    if (!dense && crxh::is_pinned(a.dst)) {
      CRX_HIP(hipMemcpy2DAsync(a.dst, a.pitch, a.d, a.row_bytes, a.row_bytes, a.rows, hipMemcpyDeviceToHost, c_->s_cmp));
      return CRX_OK;
    }
    // pageable: the DMA of slot k+1 runs while the copy threads drain slot k into the caller's array
    char* pin = static_cast<char*>(c_->pws.p);
    const Cut ct = cut(a);
    auto piece = [&](size_t k, size_t& doff, size_t& len, size_t& r0, size_t& nr) {
      if (a.rows == 1) { doff = k * ct.unit_bytes; len = std::min(ct.unit_bytes, a.row_bytes - doff); r0 = 0; nr = 1; }
      else { r0 = k * ct.unit_rows; nr = std::min(ct.unit_rows, a.rows - r0); doff = r0 * a.row_bytes; len = nr * a.row_bytes; }
    };
    auto issue = [&](size_t k) -> hipError_t {
      size_t doff, len, r0, nr; piece(k, doff, len, r0, nr);
      hipError_t e = hipMemcpyAsync(pin + (k & 1) * slot_, a.d + doff, len, hipMemcpyDeviceToHost, c_->s_cmp);
      if (e != hipSuccess) return e;
      return hipEventRecord(c_->ev_tmp[k & 1], c_->s_cmp);
    };
    CRX_HIP(issue(0));
    for (size_t k = 0; k < ct.chunks; ++k) {
      if (k + 1 < ct.chunks) CRX_HIP(issue(k + 1));
      CRX_HIP(hipEventSynchronize(c_->ev_tmp[k & 1]));
      size_t doff, len, r0, nr; piece(k, doff, len, r0, nr);
      crxh::CopyPool::Ticket t;
      if (a.rows == 1) crxh::CopyPool::get().submit(crxh::CopyPool::Job{a.dst + doff, pin + (k & 1) * slot_, len, 1, len, len}, &t);
      else crxh::CopyPool::get().submit(crxh::CopyPool::Job{a.dst + r0 * a.pitch, pin + (k & 1) * slot_, a.row_bytes, nr, a.pitch, a.row_bytes}, &t);
      crxh::CopyPool::get().wait(&t);
    }
    return CRX_OK;
  }
};

#define CRX_TRY(call) do { if (int rc__ = (call)) return rc__; } while (0)

// Run fn(shard) for every shard of [0, n) over the device set (crx_set_devices): one host thread per shard, each on its device;
// the first failure is reported.  With no device set: one shard on the calling thread's current device, no thread.
template <class F>
int run_sharded(int n, F&& fn) {
  if (int rc = check_device()) return rc;
  int cur = 0;
  CRX_HIP(hipGetDevice(&cur));
  const std::vector<crxh::Shard> sh = crxh::shards_for(n, cur);
  if (sh.size() == 1 && sh[0].dev == cur) return fn(sh[0]);
  std::vector<int> rcs(sh.size(), CRX_OK);
  std::vector<std::string> errs(sh.size());
  std::vector<std::thread> th;
  for (size_t i = 0; i < sh.size(); ++i)
    th.emplace_back([&, i] {
      const hipError_t e = hipSetDevice(sh[i].dev);
      if (e != hipSuccess) { rcs[i] = hip_fail(e, "hipSetDevice (shard)"); errs[i] = g_err; return; }
      rcs[i] = fn(sh[i]);
      if (rcs[i]) errs[i] = g_err;
    });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < sh.size(); ++i)
    if (rcs[i]) { g_err = "shard " + std::to_string(i) + " (device " + std::to_string(sh[i].dev) + "): " + errs[i]; return rcs[i]; }
  return CRX_OK;
}

}  // namespace

extern "C" {

int crx_version(void) { return 400; }  // 0.4.0 (0.3.0 + device selection and sharded host-pointer entries, grow-only workspace, crx_mpc_closed_loop_flags_batch_dev)

// crx_init only checks that a device is there and forces the HIP runtime + code object to load now rather than in the first
// timed call; crx_shutdown drains the devices and gives the host-pointer workspaces back.  Both are optional.  The only state the
// engine keeps is on the host-pointer side (crx_host.h): per-device grow-only workspaces and the device set.
int crx_init(void) {
  if (int rc = check_device()) return rc;
  CRX_HIP(hipFree(nullptr));
  return CRX_OK;
}
int crx_release_workspace(void) {
  const int nd = crx_device_count();
  int cur = 0;
  if (nd == 0) return CRX_OK;
  CRX_HIP(hipGetDevice(&cur));
  for (int d = 0; d < nd && d < crxh::kMaxDevices; ++d) {
    crxh::DeviceCtx& c = crxh::ctx_table()[d];
    std::lock_guard<std::mutex> l(c.mu);
    if (!c.dws.p && !c.pws.p) continue;
    CRX_HIP(hipSetDevice(d));
    CRX_HIP(hipDeviceSynchronize());
    c.release_workspace();
  }
  CRX_HIP(hipSetDevice(cur));
  return CRX_OK;
}
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
  if (int rc = crx_release_workspace()) return rc;
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
  p->tol = 1e-9; p->max_iter = 50;
}

// ---------------------------------------------------------------------------------------------
// EKF
// ---------------------------------------------------------------------------------------------
int crx_motion_model_batch_dev(int n, const float* x, const float* u, float* x_out,
                               const crx_ekf_params* prm, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n && (!x || !u || !x_out))) return fail(CRX_ERR_INVALID, "motion_model: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::motion_model_kernel, dim3(blocks_for(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, x, u, x_out, prm ? prm->dt : 0.1);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_jacobF_batch_dev(int n, const float* x, const float* u, float* jF, const crx_ekf_params* prm,
                         void* stream) {
  CRX_TRACE();
  if (n < 0 || (n && (!x || !u || !jF))) return fail(CRX_ERR_INVALID, "jacobF: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::jacobF_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     n, x, u, jF, prm ? prm->dt : 0.1);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_observation_model_batch_dev(int n, const float* x, float* z_out, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n && (!x || !z_out))) return fail(CRX_ERR_INVALID, "observation_model: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::observation_model_kernel, dim3(blocks_for(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, x, z_out);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_jacobH(float* jH_out) {
  if (!jH_out) return fail(CRX_ERR_INVALID, "jacobH: NULL output");
  // column-major 2x4: [[1,0,0,0],[0,1,0,0]]  — a constant; no arithmetic involved
  const float h[8] = {1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f};
  std::memcpy(jH_out, h, sizeof(h));
  return CRX_OK;
}

int crx_ekf_step_batch_dev(int n, float* x, float* P, const float* z, const float* u, const float* Q,
                           const float* R, const crx_ekf_params* prm, void* stream) {
  CRX_TRACE();
  if (n < 0 || !Q || !R || (n && (!x || !P || !z || !u)))
    return fail(CRX_ERR_INVALID, "ekf_step: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  const crx::EkfConsts k = make_consts(Q, R, prm);
  const dim3 grid(blocks_for(n, CRX_EKF_STEP_BLOCK)), block(CRX_EKF_STEP_BLOCK);
#ifdef CRX_EKF_STEP_NT_FORCE     // A/B builds only (scripts/experiments/gpu_ekf_step_ab.sh)
  hipLaunchKernelGGL(crx::ekf_step_kernel<(CRX_EKF_STEP_NT_FORCE != 0)>, grid, block, 0, (hipStream_t)stream, n, x, P, z, u, k);
#else
  if (n >= crx::kEkfStepNtMinN) hipLaunchKernelGGL(crx::ekf_step_kernel<true>, grid, block, 0, (hipStream_t)stream, n, x, P, z, u, k);
  else hipLaunchKernelGGL(crx::ekf_step_kernel<false>, grid, block, 0, (hipStream_t)stream, n, x, P, z, u, k);
#endif
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

static int ekf_run_launch(int n, int T, float* x, float* P, const float* z, const float* u,
                          float* x_hist, float* P_hist, const float* Q, const float* R,
                          const crx_ekf_params* prm, void* stream, bool force_addr64) {
  if (n < 0 || T < 0 || !Q || !R || (n && (!x || !P)) || (n && T && (!z || !u)))
    return fail(CRX_ERR_INVALID, "ekf_run: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0 || T == 0) return CRX_OK;
  const crx::EkfConsts k = make_consts(Q, R, prm);
  const dim3 grid(blocks_for(n, CRX_EKF_RUN_BLOCK)), block(CRX_EKF_RUN_BLOCK);
  hipStream_t s = (hipStream_t)stream;
#ifndef CRX_EKF_PREFETCH
#define CRX_EKF_PREFETCH 4
#endif
  constexpr int D = CRX_EKF_PREFETCH;
#ifndef CRX_EKF_BUFFER_ADDRESSING
#define CRX_EKF_BUFFER_ADDRESSING 1
#endif
  // 32-bit buffer offsets (ekf_kernels.hip.h) up to kEkfBufMaxN vehicles, the 64-bit-address kernels above (tests force the
  // latter on small inputs through crx_x_ekf_run_addr64_dev)
  const bool buf = CRX_EKF_BUFFER_ADDRESSING && n <= crx::kEkfBufMaxN && !force_addr64;
#define CRX_LAUNCH_RUN(XH, PH)                                                                          \
  do {                                                                                                  \
    if (buf) hipLaunchKernelGGL((crx::ekf_run_kernel<D, XH, PH, true>), grid, block, 0, s, n, T, x, P, z, u, x_hist, P_hist, k);  \
    else hipLaunchKernelGGL((crx::ekf_run_kernel<D, XH, PH, false>), grid, block, 0, s, n, T, x, P, z, u, x_hist, P_hist, k);     \
  } while (0)
  if (x_hist && P_hist) CRX_LAUNCH_RUN(true, true);
  else if (x_hist) CRX_LAUNCH_RUN(true, false);
  else if (P_hist) CRX_LAUNCH_RUN(false, true);
  else CRX_LAUNCH_RUN(false, false);
#undef CRX_LAUNCH_RUN
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_ekf_run_batch_dev(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist, float* P_hist,
                          const float* Q, const float* R, const crx_ekf_params* prm, void* stream) {
  CRX_TRACE();
  return ekf_run_launch(n, T, x, P, z, u, x_hist, P_hist, Q, R, prm, stream, false);
}
int crx_x_ekf_run_addr64_dev(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist, float* P_hist,
                             const float* Q, const float* R, const crx_ekf_params* prm, void* stream) {
  CRX_TRACE();
  return ekf_run_launch(n, T, x, P, z, u, x_hist, P_hist, Q, R, prm, stream, true);
}

#ifdef CRX_EKF_TIMING
// debug builds only: copies out the per-workgroup {shader-clock ticks, 100 MHz real-time ticks} of the last fused launch
int crx_debug_ekf_timing(long long* out, int nblocks) {
  CRX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(crx::g_ekf_timing), sizeof(long long) * 2 * (size_t)nblocks));
  return CRX_OK;
}
#endif

int crx_ekf_simulate_inputs_dev(int n, int T, const float* u_true, float* xTrue, float* xDR,
                                const float* w, float* z, float* ud, float* xTrue_hist,
                                float* xDR_hist, const float qsim[2], const float rsim[2],
                                const crx_ekf_params* prm, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 0 || !qsim || !rsim || (n && (!u_true || !xTrue || !xDR)) || (n && T && (!w || !z || !ud)))
    return fail(CRX_ERR_INVALID, "ekf_simulate_inputs: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0 || T == 0) return CRX_OK;
  const dim3 grid(blocks_for(n, 64)), block(64);
  const double dt = prm ? prm->dt : 0.1;
  if (xTrue_hist || xDR_hist)
    hipLaunchKernelGGL((crx::ekf_simulate_inputs_kernel<true>), grid, block, 0, (hipStream_t)stream, n, T,
                       u_true, xTrue, xDR, w, z, ud, xTrue_hist, xDR_hist, qsim[0], qsim[1], rsim[0], rsim[1], dt);
  else
    hipLaunchKernelGGL((crx::ekf_simulate_inputs_kernel<false>), grid, block, 0, (hipStream_t)stream, n, T,
                       u_true, xTrue, xDR, w, z, ud, xTrue_hist, xDR_hist, qsim[0], qsim[1], rsim[0], rsim[1], dt);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// The two-lanes-per-vehicle A/B variant of the fused launch (ekf_wave2_kernels.hip.h): measured 0.61-0.73x of the production
// kernel (profiles/r02/ekf_wave_ab.txt).  Measurement only (include/crx_experimental.h); it has no general-step fallback: when
// *left_domain comes back non-zero, xEst / PEst / x_hist of this call are not valid.
int crx_x_ekf_run_pair_batch_dev(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist, const float Q[16],
                               const float R[4], const crx_ekf_params* prm, int* left_domain, void* stream) {
  CRX_TRACE();
#if !CRX_EXPERIMENTAL_KERNELS
  (void)x_hist; (void)prm; (void)left_domain; (void)stream;
  return fail(CRX_ERR_INVALID, "ekf_run_pair: this libcrx.so was built without the experimental kernels (CRX_EXPERIMENTAL_KERNELS=0)");
#else
  if (n < 0 || T < 0 || !Q || !R || (n && (!x || !P)) || (n && T && (!z || !u)))
    return fail(CRX_ERR_INVALID, "ekf_run_pair: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0 || T == 0) return CRX_OK;
  const crx::EkfConsts k = make_consts(Q, R, prm);
  hipLaunchKernelGGL((crx::ekf_run_pair_kernel<4>), dim3(blocks_for(2 * (size_t)n, 64)), dim3(64), 0, (hipStream_t)stream, n, T, x, P, z, u,
                     x_hist, k, left_domain);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
#endif
}

// w[t][a][0..3] = the four N(0,1) draws of (seed, stream, global agent id agent0 + a, step t): see crx_philox.h
namespace crx {
__global__ void __launch_bounds__(256) normal_draws_kernel(int n, int T, unsigned long long agent0, unsigned long long seed,
                                                           unsigned stream_id, float4* __restrict__ w) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one (step, agent) pair per lane, agent fastest
  if (i >= (size_t)n * T) return;
  const unsigned step = (unsigned)(i / n);
  const unsigned long long a = agent0 + (i % n);
  float o[4];
  philox_normal4(seed, stream_id, a, step, o);
  w[i] = make_float4(o[0], o[1], o[2], o[3]);
}
}  // namespace crx

int crx_normal_draws_dev(int n, int T, long long agent0, unsigned long long seed, unsigned stream_id, float* w, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 0 || agent0 < 0 || ((size_t)n * T && !w)) return fail(CRX_ERR_INVALID, "normal_draws: bad argument");
  if (int rc = check_device()) return rc;
  if ((size_t)n * T == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::normal_draws_kernel, dim3(blocks_for((size_t)n * T, 256)), dim3(256), 0, (hipStream_t)stream, n, T,
                     (unsigned long long)agent0, seed, stream_id, reinterpret_cast<float4*>(w));
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// ---- host-pointer variants --------------------------------------------------------------------------
// Arguments are marshalled by HostCall (zero-copy for small calls, workspace + staged DMA otherwise) and the agents are split
// over the device set (crx_set_devices) — every array here is per agent, so a shard is a pointer offset.
int crx_motion_model_batch(int n, const float* x, const float* u, float* x_out, const crx_ekf_params* prm) {
  CRX_TRACE();
  if (n < 0 || (n && (!x || !u || !x_out))) return fail(CRX_ERR_INVALID, "motion_model: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc;
    CRX_TRY(hc.open());
    const int ix = hc.add(x + 4 * a0, x_out + 4 * a0, 16 * nl), iu = hc.add(u + 2 * a0, nullptr, 8 * nl);
    CRX_TRY(hc.commit());
    CRX_TRY(crx_motion_model_batch_dev((int)nl, hc.p<float>(ix), hc.p<float>(iu), hc.p<float>(ix), prm, hc.stream()));
    return hc.finish();
  });
}

int crx_jacobF_batch(int n, const float* x, const float* u, float* jF, const crx_ekf_params* prm) {
  CRX_TRACE();
  if (n < 0 || (n && (!x || !u || !jF))) return fail(CRX_ERR_INVALID, "jacobF: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc;
    CRX_TRY(hc.open());
    const int ix = hc.add(x + 4 * a0, nullptr, 16 * nl), iu = hc.add(u + 2 * a0, nullptr, 8 * nl), ij = hc.add(nullptr, jF + 16 * a0, 64 * nl);
    CRX_TRY(hc.commit());
    CRX_TRY(crx_jacobF_batch_dev((int)nl, hc.p<float>(ix), hc.p<float>(iu), hc.p<float>(ij), prm, hc.stream()));
    return hc.finish();
  });
}

int crx_observation_model_batch(int n, const float* x, float* z_out) {
  CRX_TRACE();
  if (n < 0 || (n && (!x || !z_out))) return fail(CRX_ERR_INVALID, "observation_model: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc;
    CRX_TRY(hc.open());
    const int ix = hc.add(x + 4 * a0, nullptr, 16 * nl), iz = hc.add(nullptr, z_out + 2 * a0, 8 * nl);
    CRX_TRY(hc.commit());
    CRX_TRY(crx_observation_model_batch_dev((int)nl, hc.p<float>(ix), hc.p<float>(iz), hc.stream()));
    return hc.finish();
  });
}

int crx_ekf_step_batch(int n, float* x, float* P, const float* z, const float* u, const float* Q,
                       const float* R, const crx_ekf_params* prm) {
  return crx_ekf_run_batch(n, 1, x, P, z, u, nullptr, nullptr, Q, R, prm);
}

}  // extern "C"

namespace {

// The fused EKF run for the agents [a0, a1) of an n-agent batch, host pointers, on the current device: a three-stream pipeline
// over time chunks.  z, u are [T][n][2], x_hist [T][n][4], P_hist [T][n][16] (time-major: a shard's columns of a chunk are
// `rows` of nl agents, n agents apart).  Chunk k: its z,u rows are gathered into a pinned slot by the copy threads (or DMA'd
// straight from the caller's memory when that is pinned), go to the device on s_in, the kernel runs its steps on s_cmp — the
// filter state x, P staying in the workspace from chunk to chunk, so the results are those of ONE T-step launch bit for bit —
// and the chunk's history rows return on s_out, scattered into the caller's arrays by the copy threads.  While the device works
// on chunk k the host fills chunk k+1 and drains chunk k-1.  Rings of kRing slots; reuse is fenced by events.
int ekf_run_host_shard(int n, int a0, int a1, int T, float* x, float* P, const float* z, const float* u, float* x_hist,
                       float* P_hist, const float* Q, const float* R, const crx_ekf_params* prm) {
  using crxh::kRing;
  crxh::DeviceCtx* c = nullptr;
  std::unique_lock<std::mutex> lock;
  CRX_TRY(ctx_open(&c, lock));
  const size_t nl = (size_t)(a1 - a0), nn = (size_t)n;
  const size_t per_step_in = 16 * nl, per_step_out = (x_hist ? 16 * nl : 0) + (P_hist ? 64 * nl : 0);
  const bool direct_in = crxh::is_pinned(z) && crxh::is_pinned(u);
  const bool direct_out = (!x_hist || crxh::is_pinned(x_hist)) && (!P_hist || crxh::is_pinned(P_hist));
  // steps per chunk: ~16 MB of the larger direction when the chunk is staged by the copy threads (short enough to overlap, long
  // enough to amortise the trip), ~64 MB when both directions are DMA'd in place; at least 1, at most T
  const size_t per_step = std::max(per_step_in, per_step_out);
  size_t Tc = std::max<size_t>(1, ((direct_in && direct_out) ? (64u << 20) : (16u << 20)) / per_step);
  Tc = std::min<size_t>(Tc, (size_t)T);
  const int C = (int)(((size_t)T + Tc - 1) / Tc);
  // a slot of the input ring is [z rows | u rows], of the output ring [x_hist rows | P_hist rows]; every part 256-byte aligned
  const size_t zb = crxh::align_up(8 * nl * Tc), inb = 2 * zb, hx = crxh::align_up(16 * nl * Tc);
  const size_t outb = (x_hist ? hx : 0) + (P_hist ? crxh::align_up(64 * nl * Tc) : 0);
  const size_t xb = crxh::align_up(16 * nl), Pb = crxh::align_up(64 * nl);
  const bool whole = nl == nn;                       // the shard is the whole batch: a chunk's rows are one contiguous block
  hipError_t e = c->dws.reserve(xb + Pb + kRing * (inb + outb));
  if (e == hipSuccess) e = c->pws.reserve(xb + Pb + (direct_in ? 0 : kRing * inb) + (direct_out ? 0 : kRing * outb));
  if (e != hipSuccess) { hip_fail(e, "workspace (ekf_run)"); return CRX_ERR_ALLOC; }
  char* dbase = static_cast<char*>(c->dws.p);
  float* dx = reinterpret_cast<float*>(dbase);
  float* dP = reinterpret_cast<float*>(dbase + xb);
  char* din = dbase + xb + Pb;                         // ring: [z rows | u rows] of a chunk
  char* dout = din + kRing * inb;                      // ring: [x_hist rows | P_hist rows] of a chunk
  char* pbase = static_cast<char*>(c->pws.p);
  char* pxP = pbase;
  char* pin_in = pbase + xb + Pb;
  char* pin_out = pin_in + (direct_in ? 0 : kRing * inb);
  crxh::CopyPool& pool = crxh::CopyPool::get();

  // initial state: through the pinned block, on the compute stream
  std::memcpy(pxP, x + 4 * (size_t)a0, 16 * nl);
  std::memcpy(pxP + xb, P + 16 * (size_t)a0, 64 * nl);
  CRX_HIP(hipMemcpyAsync(dx, pxP, 16 * nl, hipMemcpyHostToDevice, c->s_cmp));
  CRX_HIP(hipMemcpyAsync(dP, pxP + xb, 64 * nl, hipMemcpyHostToDevice, c->s_cmp));

  auto steps_of = [&](int k) { return std::min(Tc, (size_t)T - (size_t)k * Tc); };
  // rows of chunk k in the caller's time-major arrays
  auto zrow = [&](int k) { return reinterpret_cast<const char*>(z) + ((size_t)k * Tc * nn + (size_t)a0) * 8; };
  auto urow = [&](int k) { return reinterpret_cast<const char*>(u) + ((size_t)k * Tc * nn + (size_t)a0) * 8; };
  auto xhrow = [&](int k) { return reinterpret_cast<char*>(x_hist) + ((size_t)k * Tc * nn + (size_t)a0) * 16; };
  auto Phrow = [&](int k) { return reinterpret_cast<char*>(P_hist) + ((size_t)k * Tc * nn + (size_t)a0) * 64; };

  for (int it = 0; it <= C + 1; ++it) {
    // (a) device work of chunk k = it - 1 (its inputs were staged during the previous trip)
    const int k = it - 1;
    if (k >= 0 && k < C) {
      const int sl = k % kRing;
      const size_t tc = steps_of(k);
      char* dz = din + (size_t)sl * inb;
      char* du = dz + zb;
      if (k >= kRing) CRX_HIP(hipStreamWaitEvent(c->s_in, c->ev_cmp[sl], 0));          // the kernel of chunk k - kRing has read this slot
      if (direct_in && whole) {
        CRX_HIP(hipMemcpyAsync(dz, zrow(k), 8 * nl * tc, hipMemcpyHostToDevice, c->s_in));
        CRX_HIP(hipMemcpyAsync(du, urow(k), 8 * nl * tc, hipMemcpyHostToDevice, c->s_in));
      } else if (direct_in) {
        CRX_HIP(hipMemcpy2DAsync(dz, 8 * nl, zrow(k), 8 * nn, 8 * nl, tc, hipMemcpyHostToDevice, c->s_in));
        CRX_HIP(hipMemcpy2DAsync(du, 8 * nl, urow(k), 8 * nn, 8 * nl, tc, hipMemcpyHostToDevice, c->s_in));
      } else {
        CRX_HIP(hipMemcpyAsync(dz, pin_in + (size_t)sl * inb, zb + 8 * nl * tc, hipMemcpyHostToDevice, c->s_in));   // z rows, padding, u rows
      }
      CRX_HIP(hipEventRecord(c->ev_in[sl], c->s_in));
      CRX_HIP(hipStreamWaitEvent(c->s_cmp, c->ev_in[sl], 0));
      if (k >= kRing) CRX_HIP(hipStreamWaitEvent(c->s_cmp, c->ev_out[sl], 0));         // the D2H of chunk k - kRing has left this slot
      float* dxh = x_hist ? reinterpret_cast<float*>(dout + (size_t)sl * outb) : nullptr;
      float* dPh = P_hist ? reinterpret_cast<float*>(dout + (size_t)sl * outb + (x_hist ? hx : 0)) : nullptr;
      int rc;
      if (T == 1 && !x_hist && !P_hist)
        rc = crx_ekf_step_batch_dev((int)nl, dx, dP, reinterpret_cast<float*>(dz), reinterpret_cast<float*>(du), Q, R, prm, c->s_cmp);
      else
        rc = crx_ekf_run_batch_dev((int)nl, (int)tc, dx, dP, reinterpret_cast<float*>(dz), reinterpret_cast<float*>(du), dxh, dPh, Q, R, prm, c->s_cmp);
      if (rc) return rc;
      CRX_HIP(hipEventRecord(c->ev_cmp[sl], c->s_cmp));
      if (per_step_out) {
        CRX_HIP(hipStreamWaitEvent(c->s_out, c->ev_cmp[sl], 0));
        if (direct_out && whole) {
          if (x_hist) CRX_HIP(hipMemcpyAsync(xhrow(k), dxh, 16 * nl * tc, hipMemcpyDeviceToHost, c->s_out));
          if (P_hist) CRX_HIP(hipMemcpyAsync(Phrow(k), dPh, 64 * nl * tc, hipMemcpyDeviceToHost, c->s_out));
        } else if (direct_out) {
          if (x_hist) CRX_HIP(hipMemcpy2DAsync(xhrow(k), 16 * nn, dxh, 16 * nl, 16 * nl, tc, hipMemcpyDeviceToHost, c->s_out));
          if (P_hist) CRX_HIP(hipMemcpy2DAsync(Phrow(k), 64 * nn, dPh, 64 * nl, 64 * nl, tc, hipMemcpyDeviceToHost, c->s_out));
        } else {
          char* po = pin_out + (size_t)sl * outb;
          if (x_hist) CRX_HIP(hipMemcpyAsync(po, dxh, 16 * nl * tc, hipMemcpyDeviceToHost, c->s_out));
          if (P_hist) CRX_HIP(hipMemcpyAsync(po + (x_hist ? hx : 0), dPh, 64 * nl * tc, hipMemcpyDeviceToHost, c->s_out));
        }
        CRX_HIP(hipEventRecord(c->ev_out[sl], c->s_out));
      }
    }
    // (b) stage the inputs of chunk `it` into its pinned slot (free once the H2D of chunk it - kRing has completed)
    crxh::CopyPool::Joined jin, jout;             // joined at the end of the trip — and on every early return
    crxh::CopyPool::Ticket &tin = jin.t, &tout = jout.t;
    if (it < C && !direct_in) {
      const int sl = it % kRing;
      const size_t tc = steps_of(it);
      if (it >= kRing) CRX_HIP(hipEventSynchronize(c->ev_in[sl]));
      char* pz = pin_in + (size_t)sl * inb;
      pool.submit(crxh::CopyPool::Job{pz, zrow(it), 8 * nl, tc, 8 * nl, 8 * nn}, &tin);
      pool.submit(crxh::CopyPool::Job{pz + zb, urow(it), 8 * nl, tc, 8 * nl, 8 * nn}, &tin);
    }
    // (c) drain the history rows of chunk it - 2 into the caller's arrays
    const int d = it - 2;
    if (d >= 0 && d < C && per_step_out && !direct_out) {
      const int sl = d % kRing;
      const size_t tc = steps_of(d);
      CRX_HIP(hipEventSynchronize(c->ev_out[sl]));
      char* po = pin_out + (size_t)sl * outb;
      if (x_hist) pool.submit(crxh::CopyPool::Job{xhrow(d), po, 16 * nl, tc, 16 * nn, 16 * nl}, &tout);
      if (P_hist) pool.submit(crxh::CopyPool::Job{Phrow(d), po + (x_hist ? hx : 0), 64 * nl, tc, 64 * nn, 64 * nl}, &tout);
    }
  }
  // final state (the compute stream is behind the last kernel); every history row has left the device before we return
  CRX_HIP(hipMemcpyAsync(pxP, dx, 16 * nl, hipMemcpyDeviceToHost, c->s_cmp));
  CRX_HIP(hipMemcpyAsync(pxP + xb, dP, 64 * nl, hipMemcpyDeviceToHost, c->s_cmp));
  CRX_HIP(hipStreamSynchronize(c->s_cmp));
  if (per_step_out) CRX_HIP(hipStreamSynchronize(c->s_out));
  std::memcpy(x + 4 * (size_t)a0, pxP, 16 * nl);
  std::memcpy(P + 16 * (size_t)a0, pxP + xb, 64 * nl);
  return CRX_OK;
}

}  // namespace

extern "C" {

int crx_ekf_run_batch(int n, int T, float* x, float* P, const float* z, const float* u, float* x_hist,
                      float* P_hist, const float* Q, const float* R, const crx_ekf_params* prm) {
  CRX_TRACE();
  if (n < 0 || T < 0 || !Q || !R || (n && (!x || !P)) || (n && T && (!z || !u)))
    return fail(CRX_ERR_INVALID, "ekf_run: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0 || T == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nn = (size_t)n, tt = (size_t)T;
    if (nl * tt * (32 + (P_hist ? 64 : 0)) + 80 * nl > crxh::kZeroCopyBytes)
      return ekf_run_host_shard(n, sh.a0, sh.a1, T, x, P, z, u, x_hist, P_hist, Q, R, prm);
    // a small call — the literal drop-in, ekf_estimation() for one vehicle — is zero-copy: one pinned block, one launch
    HostCall hc;
    CRX_TRY(hc.open());
    const int ix = hc.add(x + 4 * a0, x + 4 * a0, 16 * nl), iP = hc.add(P + 16 * a0, P + 16 * a0, 64 * nl);
    const int iz = hc.add2d(z + 2 * a0, nullptr, 8 * nl, tt, 8 * nn), iu = hc.add2d(u + 2 * a0, nullptr, 8 * nl, tt, 8 * nn);
    const int ih = x_hist ? hc.add2d(nullptr, x_hist + 4 * a0, 16 * nl, tt, 16 * nn) : -1;
    const int iH = P_hist ? hc.add2d(nullptr, P_hist + 16 * a0, 64 * nl, tt, 64 * nn) : -1;
    CRX_TRY(hc.commit());
    if (T == 1 && !x_hist && !P_hist)
      CRX_TRY(crx_ekf_step_batch_dev((int)nl, hc.p<float>(ix), hc.p<float>(iP), hc.p<float>(iz), hc.p<float>(iu), Q, R, prm, hc.stream()));
    else
      CRX_TRY(crx_ekf_run_batch_dev((int)nl, T, hc.p<float>(ix), hc.p<float>(iP), hc.p<float>(iz), hc.p<float>(iu),
                                    ih >= 0 ? hc.p<float>(ih) : nullptr, iH >= 0 ? hc.p<float>(iH) : nullptr, Q, R, prm, hc.stream()));
    return hc.finish();
  });
}

// ---------------------------------------------------------------------------------------------
// DARE / dlqr
// ---------------------------------------------------------------------------------------------
// structured: 1 = detect the pattern lqr_steering_control builds (per agent) and serve those agents by the structured kernels, the
// rest by a dense kernel (two launches, no workspace, no synchronisation); 0 = a dense kernel for everybody.
// dense_lanes: 1 = dare_dense_kernel (one agent per lane), 4 = dare_dense_quad_kernel (one row of X per lane of a quad), 0 = by batch size.
static int dare_batch_launch(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                             float eps, int maxiter, float* X, float* K, int* iters, void* stream, int structured, int dense_lanes) {
  if (n < 0 || (dim != 4 && dim != 5) || (n && (!A || !B || !Q || !R)))
    return fail(CRX_ERR_INVALID, "dare: bad argument (dim must be 4 or 5)");
  if (dense_lanes != 0 && dense_lanes != 1 && dense_lanes != 4) return fail(CRX_ERR_INVALID, "dare: lanes_per_agent must be 0 (auto), 1 or 4");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned bs = iter_block();
  const dim3 grid(blocks_for(n, bs)), block(bs);
  const dim3 qgrid(blocks_for(4 * (size_t)n, 256)), qblock(256);
  // a quad per agent while the batch leaves SIMDs without a wave of their own (the structured kernels' crossover)
  if (dense_lanes == 0) dense_lanes = (n <= kDareDenseQuadMaxAgents) ? 4 : 1;
  if (structured) {
    const crx::DareFromMats src{A, B, Q, R};
    if (n <= kDareQuadMaxAgents) {
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<5, crx::DareFromMats>), qgrid, qblock, 0, s, n, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<4, crx::DareFromMats>), qgrid, qblock, 0, s, n, src, eps, maxiter, X, K, iters);
    } else if (n <= kDareChainMaxAgents) {
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_kernel<5, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_kernel<4, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
    } else if (maxiter > 0 && dare_refill_chunk(n)) {
      const int chunk = dare_refill_chunk(n);
      const dim3 rgrid(blocks_for(n, chunk));
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<5, crx::DareFromMats>), rgrid, block, 0, s, n, chunk, kDareRefillHold, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<4, crx::DareFromMats>), rgrid, block, 0, s, n, chunk, kDareRefillHold, src, eps, maxiter, X, K, iters);
    } else {
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_masked_kernel<5, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_masked_kernel<4, crx::DareFromMats>), grid, block, 0, s, n, src, eps, maxiter, X, K, iters);
    }
    CRX_HIP(hipGetLastError());
  }
#define CRX_LAUNCH_DENSE(DIM, SKIP)                                                                                                     \
  do {                                                                                                                                  \
    if (dense_lanes == 4) hipLaunchKernelGGL((crx::dare_dense_quad_kernel<DIM, SKIP>), qgrid, qblock, 0, s, n, A, B, Q, R, eps, maxiter, X, K, iters); \
    else hipLaunchKernelGGL((crx::dare_dense_kernel<DIM, SKIP>), grid, block, 0, s, n, A, B, Q, R, eps, maxiter, X, K, iters);          \
  } while (0)
  if (structured) { if (dim == 5) CRX_LAUNCH_DENSE(5, true); else CRX_LAUNCH_DENSE(4, true); }
  else { if (dim == 5) CRX_LAUNCH_DENSE(5, false); else CRX_LAUNCH_DENSE(4, false); }
#undef CRX_LAUNCH_DENSE
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_dare_batch_dev(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                       float eps, int maxiter, float* X, float* K, int* iters, void* stream) {
  CRX_TRACE();
  return dare_batch_launch(n, dim, A, B, Q, R, eps, maxiter, X, K, iters, stream, 1, 0);
}
int crx_x_dare_batch_dense_dev(int n, int dim, const float* A, const float* B, const float* Q, const float* R,
                               float eps, int maxiter, float* X, float* K, int* iters, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return dare_batch_launch(n, dim, A, B, Q, R, eps, maxiter, X, K, iters, stream, 0, lanes_per_agent);
}

// lanes_per_agent: 1 = dare_from_v_kernel, 4 = dare_from_v_quad_kernel, 0 = chosen by batch size.
// refill_chunk: agents per wave of the lane-refilling kernel; 0 = the product's choice (dare_refill_chunk: above 262,144 agents,
// no refilling below), -1 = never (the masked kernel: rounds 2-3's throughput-regime kernel, kept for the A/B)
static int dare_from_v_launch(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                              int* iters, void* stream, int lanes_per_agent, int refill_chunk = 0, int refill_hold = kDareRefillHold) {
  if (n < 0 || (dim != 4 && dim != 5) || (n && !v))
    return fail(CRX_ERR_INVALID, "dare_from_v: bad argument (dim must be 4 or 5)");
  if (lanes_per_agent != 0 && lanes_per_agent != 1 && lanes_per_agent != 4)
    return fail(CRX_ERR_INVALID, "dare_from_v: lanes_per_agent must be 0 (auto), 1 or 4");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_lqr_params p;
  if (prm) p = *prm; else crx_lqr_default_params(&p);
  // Four lanes per agent shorten the launch while the batch leaves SIMDs without a wave of their own; in the throughput
  // regime the one-lane kernel executes fewer instructions per agent (profiles/r03/dare_lanes_ab.txt).
  if (lanes_per_agent == 0) lanes_per_agent = (n <= kDareQuadMaxAgents) ? 4 : 1;
  const crx::DareFromV src{v, (float)p.dt, p.L};
  if (lanes_per_agent == 4) {
    const dim3 grid(blocks_for(4 * (size_t)n, 256)), block(256);
    if (dim == 5)
      hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<5, crx::DareFromV>), grid, block, 0, (hipStream_t)stream, n, src, p.eps, p.maxiter, X, K, iters);
    else
      hipLaunchKernelGGL((crx::dare_from_v_quad_kernel<4, crx::DareFromV>), grid, block, 0, (hipStream_t)stream, n, src, p.eps, p.maxiter, X, K, iters);
  } else {
    const dim3 grid(blocks_for(n, 64)), block(64);
    // up to ~1.5 waves per SIMD the launch is a latency chain: nobody masked off, two evaluations per branch (89 VGPRs); beyond,
    // the masked loop at eight waves per SIMD (60 VGPRs)
    const bool chain = n <= kDareChainMaxAgents;
    if (refill_chunk == 0) refill_chunk = dare_refill_chunk(n) ? dare_refill_chunk(n) : -1;
    if (refill_chunk > 0 && p.maxiter > 0) {
      const dim3 rgrid(blocks_for(n, refill_chunk));
      if (dim == 5) hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<5, crx::DareFromV>), rgrid, block, 0, (hipStream_t)stream, n, refill_chunk, refill_hold, src, p.eps, p.maxiter, X, K, iters);
      else hipLaunchKernelGGL((crx::dare_from_v_refill_kernel<4, crx::DareFromV>), rgrid, block, 0, (hipStream_t)stream, n, refill_chunk, refill_hold, src, p.eps, p.maxiter, X, K, iters);
      CRX_HIP(hipGetLastError());
      return CRX_OK;
    }
#define CRX_LAUNCH_DV(KERNEL, DIM) \
    hipLaunchKernelGGL((crx::KERNEL<DIM, crx::DareFromV>), grid, block, 0, (hipStream_t)stream, n, src, p.eps, p.maxiter, X, K, iters)
    if (dim == 5) { if (chain) CRX_LAUNCH_DV(dare_from_v_kernel, 5); else CRX_LAUNCH_DV(dare_from_v_masked_kernel, 5); }
    else { if (chain) CRX_LAUNCH_DV(dare_from_v_kernel, 4); else CRX_LAUNCH_DV(dare_from_v_masked_kernel, 4); }
#undef CRX_LAUNCH_DV
  }
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_dare_from_v_batch_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                              int* iters, void* stream) {
  CRX_TRACE();
  return dare_from_v_launch(n, dim, v, prm, X, K, iters, stream, 0);
}

int crx_x_dare_from_v_lanes_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                                int* iters, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return dare_from_v_launch(n, dim, v, prm, X, K, iters, stream, lanes_per_agent);
}

int crx_x_dare_from_v_refill_dev(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K,
                                 int* iters, void* stream, int agents_per_wave, int hold_lanes) {
  CRX_TRACE();
  if (agents_per_wave != -1 && (agents_per_wave < 64 || agents_per_wave > (1 << 20)))
    return fail(CRX_ERR_INVALID, "dare_from_v_refill: agents_per_wave must be -1 (the masked kernel) or 64 .. 2^20");
  if (hold_lanes < 1 || hold_lanes > 64) return fail(CRX_ERR_INVALID, "dare_from_v_refill: hold_lanes must be 1 .. 64");
  return dare_from_v_launch(n, dim, v, prm, X, K, iters, stream, 1, agents_per_wave, hold_lanes);
}

int crx_dare_batch(int n, int dim, const float* A, const float* B, const float* Q, const float* R, float eps,
                   int maxiter, float* X, float* K, int* iters) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || (n && (!A || !B || !Q || !R)))
    return fail(CRX_ERR_INVALID, "dare: bad argument (dim must be 4 or 5)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, m = (dim == 5) ? 2 : 1, d2 = (size_t)dim * dim, db = (size_t)dim * m;
    HostCall hc;
    CRX_TRY(hc.open());
    const int iA = hc.add(A + d2 * a0, nullptr, 4 * d2 * nl), iB = hc.add(B + db * a0, nullptr, 4 * db * nl);
    const int iQ = hc.add(Q + d2 * a0, nullptr, 4 * d2 * nl), iR = hc.add(R + m * m * a0, nullptr, 4 * m * m * nl);
    const int iX = X ? hc.add(nullptr, X + d2 * a0, 4 * d2 * nl) : -1, iK = K ? hc.add(nullptr, K + db * a0, 4 * db * nl) : -1;
    const int iI = iters ? hc.add(nullptr, iters + a0, 4 * nl) : -1;
    CRX_TRY(hc.commit());
    CRX_TRY(crx_dare_batch_dev((int)nl, dim, hc.p<float>(iA), hc.p<float>(iB), hc.p<float>(iQ), hc.p<float>(iR), eps, maxiter,
                               iX >= 0 ? hc.p<float>(iX) : nullptr, iK >= 0 ? hc.p<float>(iK) : nullptr, iI >= 0 ? hc.p<int>(iI) : nullptr,
                               hc.stream()));
    return hc.finish();
  });
}

int crx_dare_from_v_batch(int n, int dim, const float* v, const crx_lqr_params* prm, float* X, float* K, int* iters) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || (n && !v))
    return fail(CRX_ERR_INVALID, "dare_from_v: bad argument (dim must be 4 or 5)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, m = (dim == 5) ? 2 : 1, d2 = (size_t)dim * dim, db = (size_t)dim * m;
    HostCall hc;
    CRX_TRY(hc.open());
    const int iv = hc.add(v + a0, nullptr, 4 * nl);
    const int iX = X ? hc.add(nullptr, X + d2 * a0, 4 * d2 * nl) : -1, iK = K ? hc.add(nullptr, K + db * a0, 4 * db * nl) : -1;
    const int iI = iters ? hc.add(nullptr, iters + a0, 4 * nl) : -1;
    CRX_TRY(hc.commit());
    CRX_TRY(crx_dare_from_v_batch_dev((int)nl, dim, hc.p<float>(iv), prm, iX >= 0 ? hc.p<float>(iX) : nullptr,
                                      iK >= 0 ? hc.p<float>(iK) : nullptr, iI >= 0 ? hc.p<int>(iI) : nullptr, hc.stream()));
    return hc.finish();
  });
}

// ---------------------------------------------------------------------------------------------
// MPC
// ---------------------------------------------------------------------------------------------
// agents_per_wave (1..64) and waves_per_workgroup (1..4): the launch geometry; the product entry point uses full waves in
// single-wave workgroups (every emptier or stacked geometry measured slower: profiles/r02/mpc_tail.txt).
static int mpc_solve_launch(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm,
                            float* sol, int* status, double* cost, void* stream, int agents_per_wave, int waves_per_workgroup) {
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve: bad argument (2 <= T <= 64)");
  if (agents_per_wave < 1 || agents_per_wave > 64 || waves_per_workgroup < 1 || waves_per_workgroup > 4)
    return fail(CRX_ERR_INVALID, "mpc_solve: launch geometry out of range (1..64 agents per wave, 1..4 waves per workgroup)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const hipError_t e = crx::mpc_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream, agents_per_wave, waves_per_workgroup);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc launch");
}
// lanes_per_agent: 1 = mpc_kernel (one agent per lane), 4 = mpc_quad_kernel (a DPP quad per agent, parallel line search; T <= 24;
// measured 0.95x at BASELINE configs[3] and less beyond, never selected), 0 = what the product entry point uses (= 1).
static int mpc_solve_lanes(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                           double* cost, void* stream, int lanes_per_agent) {
  if (lanes_per_agent != 0 && lanes_per_agent != 1 && lanes_per_agent != 4)
    return fail(CRX_ERR_INVALID, "mpc_solve: lanes_per_agent must be 0 (auto), 1 or 4");
  if (lanes_per_agent == 0) lanes_per_agent = 1;     // the quad variant lost its A/B at every batch size (profiles/r03/mpc_lanes_ab.txt)
  if (lanes_per_agent == 1) return mpc_solve_launch(n, T, x0, xref, prm, sol, status, cost, stream, 64, 1);
#if !CRX_EXPERIMENTAL_KERNELS
  return fail(CRX_ERR_INVALID, "mpc_solve (four lanes per agent): this libcrx.so was built without the experimental kernels");
#else
  if (n < 0 || T < 2 || T > 24 || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve (four lanes per agent): bad argument (2 <= T <= 24)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const hipError_t e = crx::mpc_quad_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc launch");
#endif
}
int crx_mpc_solve_batch_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm,
                            float* sol, int* status, double* cost, void* stream) {
  CRX_TRACE();
  return mpc_solve_lanes(n, T, x0, xref, prm, sol, status, cost, stream, 0);
}
// mpc_solve for n agents with the four-variant portfolio (mpc_kernels.hip.h: mpc_variant): the same NLP, every agent answered by the
// variant of the solver that converges in the fewest sweeps.
int crx_mpc_solve_portfolio_batch_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm,
                                      float* sol, int* status, double* cost, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve (portfolio): bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const hipError_t e = crx::mpc_portfolio_launch(n, T, x0, xref, p, sol, status, cost, (hipStream_t)stream);
  return e == hipSuccess ? CRX_OK : hip_fail(e, "mpc portfolio launch");
}
int crx_x_mpc_solve_lanes_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                              double* cost, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return mpc_solve_lanes(n, T, x0, xref, prm, sol, status, cost, stream, lanes_per_agent);
}
int crx_x_mpc_solve_geometry_dev(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol, int* status,
                                 double* cost, void* stream, int agents_per_wave, int waves_per_workgroup) {
  CRX_TRACE();
  return mpc_solve_launch(n, T, x0, xref, prm, sol, status, cost, stream, agents_per_wave, waves_per_workgroup);
}

static int mpc_solve_host(bool portfolio, int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol,
                          int* status, double* cost) {
  if (n < 0 || T < 2 || T > CRX_MPC_MAX_T || (n && (!x0 || !xref || !sol)))
    return fail(CRX_ERR_INVALID, "mpc_solve: bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
    HostCall hc;
    CRX_TRY(hc.open());
    const int ix = hc.add(x0 + 4 * a0, nullptr, 16 * nl), ir = hc.add(xref + 4 * (size_t)T * a0, nullptr, 16 * (size_t)T * nl);
    const int is = hc.add(nullptr, sol + nv * a0, 4 * nv * nl);
    const int it = hc.add(nullptr, status ? status + a0 : nullptr, 4 * nl), ic = hc.add(nullptr, cost ? cost + a0 : nullptr, 8 * nl);
    CRX_TRY(hc.commit());
    if (portfolio)
      CRX_TRY(crx_mpc_solve_portfolio_batch_dev((int)nl, T, hc.p<float>(ix), hc.p<float>(ir), prm, hc.p<float>(is), hc.p<int>(it),
                                                hc.p<double>(ic), hc.stream()));
    else
      CRX_TRY(crx_mpc_solve_batch_dev((int)nl, T, hc.p<float>(ix), hc.p<float>(ir), prm, hc.p<float>(is), hc.p<int>(it), hc.p<double>(ic),
                                      hc.stream()));
    return hc.finish();
  });
}
int crx_mpc_solve_batch(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol,
                        int* status, double* cost) {
  CRX_TRACE();
  return mpc_solve_host(false, n, T, x0, xref, prm, sol, status, cost);
}
int crx_mpc_solve_portfolio_batch(int n, int T, const float* x0, const float* xref, const crx_mpc_params* prm, float* sol,
                                  int* status, double* cost) {
  CRX_TRACE();
  return mpc_solve_host(true, n, T, x0, xref, prm, sol, status, cost);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// course tracking front-end, vehicle update, closed loops
// ---------------------------------------------------------------------------------------------
namespace {

bool course_ok(const crx_course* c, bool need_ck_sp) {
  return c && c->n > 0 && c->cx && c->cy && c->cyaw && (!need_ck_sp || (c->ck && c->sp));
}
crx::CourseView view(const crx_course* c) { return crx::CourseView{c->cx, c->cy, c->cyaw, c->ck, c->sp, c->n}; }
crx::VehicleParams vparams(const crx_vehicle_params* p, int mpc) {
  crx_vehicle_params d;
  if (p) d = *p; else crx_vehicle_default_params(&d, mpc);
  return crx::VehicleParams{d.dt, d.wheelbase, d.max_steer, d.max_speed, d.min_speed, d.clamp_speed};
}
inline bool use_lds(const crx_course* c) { return c->n <= crx::kCourseLdsMax; }
inline size_t lds_bytes(const crx_course* c) { return use_lds(c) ? sizeof(float4) * (((size_t)c->n + 1) / 2) : 0; }   // two points per word
// the four-lanes-per-agent tracking kernels keep a gain slot per agent in static LDS next to the staged course (64 KB per workgroup in all)
inline bool use_quad(const crx_course* c, int n) { return n <= kDareQuadMaxAgents && use_lds(c) && lds_bytes(c) + 1024 <= 64 * 1024; }

// the course of a host-pointer call: its five arrays travel with the call's other arguments (replicated per shard)
struct CallCourse {
  int idx[5], n;
  void add(HostCall& hc, const crx_course* h) {
    const float* src[5] = {h->cx, h->cy, h->cyaw, h->ck, h->sp};
    n = h->n;
    for (int i = 0; i < 5; ++i) idx[i] = src[i] ? hc.add(src[i], nullptr, sizeof(float) * (size_t)h->n) : -1;
  }
  crx_course dev(HostCall& hc) const {
    const float* d[5];
    for (int i = 0; i < 5; ++i) d[i] = idx[i] >= 0 ? hc.p<float>(idx[i]) : nullptr;
    return crx_course{n, d[0], d[1], d[2], d[3], d[4]};
  }
};

}  // namespace

extern "C" {

void crx_vehicle_default_params(crx_vehicle_params* p, int mpc) {
  if (!p) return;
  p->dt = mpc ? 0.2 : 0.1;
  p->wheelbase = mpc ? 2.5 : 0.5;
  p->max_steer = 45.0 / 180 * 3.14159265358979323846;
  p->clamp_speed = mpc ? 1 : 0;
  p->max_speed = 55.0 / 3.6;
  p->min_speed = -20.0 / 3.6;
}

int crx_calc_nearest_index_batch_dev(int n, const float* state, const crx_course* course, int* ind, float* e, void* stream) {
  CRX_TRACE();
  if (n < 0 || !course_ok(course, false) || (n && (!state || !ind))) return fail(CRX_ERR_INVALID, "calc_nearest_index: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  const dim3 grid(blocks_for(n, crx::kTrackBlock)), block(crx::kTrackBlock);
  if (use_lds(course))
    hipLaunchKernelGGL((crx::calc_nearest_index_kernel<true>), grid, block, lds_bytes(course), (hipStream_t)stream, n, state, view(course), ind, e);
  else
    hipLaunchKernelGGL((crx::calc_nearest_index_kernel<false>), grid, block, 0, (hipStream_t)stream, n, state, view(course), ind, e);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_lqr_steering_control_batch_dev(int n, int dim, const float* state, const crx_course* course, int* ind, float* pe,
                                       float* pth_e, const crx_lqr_params* prm, float* control, void* stream) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || (n && (!state || !pe || !pth_e || !control)))
    return fail(CRX_ERR_INVALID, "lqr_steering_control: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_lqr_params p;
  if (prm) p = *prm; else crx_lqr_default_params(&p);
  const unsigned bs = iter_block();
  const crx::CourseView cv = view(course);
  hipStream_t s = (hipStream_t)stream;
  if (use_quad(course, n)) {      // a DPP quad per agent while one agent per lane would leave SIMDs idle
    const dim3 qgrid(blocks_for((size_t)n * 4, crx::kTrackBlock)), qblock(crx::kTrackBlock);
    if (dim == 5)
      hipLaunchKernelGGL((crx::lqr_steering_control_quad_kernel<5>), qgrid, qblock, lds_bytes(course), s, n, state, cv, ind, pe, pth_e, p.dt, p.L, p.eps, p.maxiter, control);
    else
      hipLaunchKernelGGL((crx::lqr_steering_control_quad_kernel<4>), qgrid, qblock, lds_bytes(course), s, n, state, cv, ind, pe, pth_e, p.dt, p.L, p.eps, p.maxiter, control);
    CRX_HIP(hipGetLastError());
    return CRX_OK;
  }
  const dim3 grid(blocks_for(n, bs)), block(bs);
#define CRX_LAUNCH_CTL(DIM, LDS) \
  hipLaunchKernelGGL((crx::lqr_steering_control_kernel<DIM, LDS>), grid, block, (LDS) ? lds_bytes(course) : 0, s, n, state, cv, ind, pe, pth_e, p.dt, p.L, p.eps, p.maxiter, control)
  if (dim == 5) { if (use_lds(course)) CRX_LAUNCH_CTL(5, true); else CRX_LAUNCH_CTL(5, false); }
  else { if (use_lds(course)) CRX_LAUNCH_CTL(4, true); else CRX_LAUNCH_CTL(4, false); }
#undef CRX_LAUNCH_CTL
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_update_batch_dev(int n, float* state, const float* a, const float* delta, const crx_vehicle_params* prm, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n && (!state || !a || !delta))) return fail(CRX_ERR_INVALID, "update: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::update_kernel, dim3(blocks_for(n, crx::kTrackBlock)), dim3(crx::kTrackBlock), 0, (hipStream_t)stream,
                     n, state, a, delta, vparams(prm, 0));
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// lanes_per_agent: 0 = by batch size (a DPP quad per agent while the batch would leave SIMDs idle with one agent per lane, and the
// course fits in LDS), 1 / 4 = forced (crx_x_lqr_closed_loop_lanes_dev)
static int lqr_closed_loop_launch(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                  const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                  float* traj_hist, int* ticks_done, void* stream, int lanes_per_agent) {
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || !loop || loop->max_ticks < 0 || (n && !state) ||
      (lanes_per_agent != 0 && lanes_per_agent != 1 && lanes_per_agent != 4))
    return fail(CRX_ERR_INVALID, "lqr_closed_loop: bad argument");
  if (lanes_per_agent == 4 && !use_quad(course, 0)) return fail(CRX_ERR_INVALID, "lqr_closed_loop: the four-lane layout needs a course that fits in LDS");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_lqr_params p;
  if (prm) p = *prm; else crx_lqr_default_params(&p);
  const crx::VehicleParams vp = vparams(veh, 0);
  const unsigned bs = iter_block();
  const crx::CourseView cv = view(course);
  hipStream_t s = (hipStream_t)stream;
  const bool quad = lanes_per_agent == 4 || (lanes_per_agent == 0 && use_quad(course, n));
  if (quad) {
    const dim3 grid(blocks_for((size_t)n * 4, crx::kTrackBlock)), block(crx::kTrackBlock);
#define CRX_LAUNCH_LOOPQ(DIM) \
    hipLaunchKernelGGL((crx::lqr_closed_loop_quad_kernel<DIM>), grid, block, lds_bytes(course), s, n, loop->max_ticks, state, cv, \
                       pe, pth_e, ind, p.dt, p.L, p.eps, p.maxiter, vp, loop->goal_x, loop->goal_y, loop->goal_dis, loop->kp,     \
                       loop->stop_speed, traj_hist, ticks_done)
    if (dim == 5) CRX_LAUNCH_LOOPQ(5); else CRX_LAUNCH_LOOPQ(4);
#undef CRX_LAUNCH_LOOPQ
    CRX_HIP(hipGetLastError());
    return CRX_OK;
  }
  const dim3 grid(blocks_for(n, bs)), block(bs);
#define CRX_LAUNCH_LOOP(DIM, LDS, CHAIN) \
  hipLaunchKernelGGL((crx::lqr_closed_loop_kernel<DIM, LDS, CHAIN>), grid, block, (LDS) ? lds_bytes(course) : 0, s, n, loop->max_ticks, state, cv, \
                     pe, pth_e, ind, p.dt, p.L, p.eps, p.maxiter, vp, loop->goal_x, loop->goal_y, loop->goal_dis, loop->kp,        \
                     loop->stop_speed, traj_hist, ticks_done)
  // one agent per lane: the unmasked Riccati loop while a SIMD holds a wave or two (as crx_dare_from_v_batch_dev does), the masked one beyond
  const bool chain = n <= kDareChainMaxAgents && use_lds(course);
  if (dim == 5) { if (chain) CRX_LAUNCH_LOOP(5, true, true); else if (use_lds(course)) CRX_LAUNCH_LOOP(5, true, false); else CRX_LAUNCH_LOOP(5, false, false); }
  else { if (chain) CRX_LAUNCH_LOOP(4, true, true); else if (use_lds(course)) CRX_LAUNCH_LOOP(4, true, false); else CRX_LAUNCH_LOOP(4, false, false); }
#undef CRX_LAUNCH_LOOP
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}
int crx_lqr_closed_loop_batch_dev(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                  const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                  float* traj_hist, int* ticks_done, void* stream) {
  CRX_TRACE();
  return lqr_closed_loop_launch(n, dim, state, course, pe, pth_e, ind, prm, veh, loop, traj_hist, ticks_done, stream, 0);
}
int crx_x_lqr_closed_loop_lanes_dev(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                                    const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                                    float* traj_hist, int* ticks_done, void* stream, int lanes_per_agent) {
  CRX_TRACE();
  return lqr_closed_loop_launch(n, dim, state, course, pe, pth_e, ind, prm, veh, loop, traj_hist, ticks_done, stream, lanes_per_agent);
}

int crx_calc_nearest_index_window_batch_dev(int n, const float* state, const crx_course* course, const int* pind, int nsearch,
                                            int* ind_out, void* stream) {
  CRX_TRACE();
  if (n < 0 || nsearch < 0 || !course_ok(course, false) || (n && (!state || !pind || !ind_out)))
    return fail(CRX_ERR_INVALID, "calc_nearest_index(window): bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::calc_nearest_index_window_kernel, dim3(blocks_for(n, crx::kTrackBlock)), dim3(crx::kTrackBlock), 0,
                     (hipStream_t)stream, n, state, view(course), pind, nsearch, ind_out);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_calc_ref_trajectory_batch_dev(int n, int T, const float* state, const crx_course* course, float dl, double dt,
                                      int nsearch, int* target_ind, float* xref, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 1 || nsearch < 0 || !course_ok(course, true) || (n && (!state || !target_ind || !xref)))
    return fail(CRX_ERR_INVALID, "calc_ref_trajectory: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::calc_ref_trajectory_kernel, dim3(blocks_for(n, crx::kTrackBlock)), dim3(crx::kTrackBlock), 0,
                     (hipStream_t)stream, n, T, state, view(course), dl, dt, nsearch, target_ind, xref, (const int*)nullptr);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// The persistent kernel keeps everything in registers / private memory: no work buffer is needed (kept for source compatibility
// with 0.1: returns 0).
size_t crx_mpc_closed_loop_work_bytes(int n, int T) {
  (void)n; (void)T;
  return 0;
}

int crx_mpc_closed_loop_batch_dev(int n, int T, float* state, const crx_course* course, float dl, int nsearch,
                                  const crx_mpc_params* prm, const crx_loop_params* loop, int* target_ind, float* traj_hist,
                                  int* ticks_done, void* work, void* stream) {
  CRX_TRACE();
  (void)work;    // the 0.2 signature: ignored, never written (0.3.0 had reused this slot for solve_flags)
  return crx_mpc_closed_loop_flags_batch_dev(n, T, state, course, dl, nsearch, prm, loop, target_ind, traj_hist, ticks_done, nullptr, stream);
}

int crx_mpc_closed_loop_flags_batch_dev(int n, int T, float* state, const crx_course* course, float dl, int nsearch,
                                        const crx_mpc_params* prm, const crx_loop_params* loop, int* target_ind, float* traj_hist,
                                        int* ticks_done, int* solve_flags, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 2 || T > 64 || !course_ok(course, true) || !loop || loop->max_ticks < 0 ||
      (n && (!state || !target_ind || !ticks_done)))
    return fail(CRX_ERR_INVALID, "mpc_closed_loop: bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_mpc_params p;
  if (prm) p = *prm; else crx_mpc_default_params(&p);
  const crx::VehicleParams vp{p.dt, p.wb, p.max_steer, p.max_speed, p.min_speed, 1};
  const size_t nn = (size_t)n, nv = 4 * (size_t)T + 2 * ((size_t)T - 1);
  (void)nn; (void)nv;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(blocks_for(n, 64)), block(64);
  const crx::CourseView cv = view(course);
  crx::MpcP q;
  q.dt = p.dt; q.wb = p.wb; q.max_steer = p.max_steer; q.max_accel = p.max_accel; q.max_speed = p.max_speed; q.min_speed = p.min_speed;
  q.r_a = p.r_a; q.r_d = p.r_delta; q.rd_a = p.rd_a; q.rd_d = p.rd_delta; q.qx = p.q_x; q.qy = p.q_y; q.qyaw = p.q_yaw; q.qv = p.q_v;
  q.tol = p.tol; q.max_iter = p.max_iter;
  // ONE persistent kernel for the whole episode (round 1 enqueued three kernels per tick from the host)
  if (T <= 8)
    hipLaunchKernelGGL((crx::mpc_closed_loop_kernel<8>), grid, block, 0, s, n, T, loop->max_ticks, state, cv, dl, nsearch, q, vp, loop->goal_x,
                       loop->goal_y, loop->goal_dis, target_ind, traj_hist, ticks_done, solve_flags);
  else if (T <= 24)
    hipLaunchKernelGGL((crx::mpc_closed_loop_kernel<24>), grid, block, 0, s, n, T, loop->max_ticks, state, cv, dl, nsearch, q, vp, loop->goal_x,
                       loop->goal_y, loop->goal_dis, target_ind, traj_hist, ticks_done, solve_flags);
  else
    hipLaunchKernelGGL((crx::mpc_closed_loop_kernel<CRX_MPC_MAX_T>), grid, block, 0, s, n, T, loop->max_ticks, state, cv, dl, nsearch, q, vp,
                       loop->goal_x, loop->goal_y, loop->goal_dis, target_ind, traj_hist, ticks_done, solve_flags);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// ---- host-pointer variants -----------------------------------------------------------------------
int crx_calc_nearest_index_batch(int n, const float* state, const crx_course* course, int* ind, float* e) {
  CRX_TRACE();
  if (n < 0 || !course_ok(course, false) || (n && (!state || !ind))) return fail(CRX_ERR_INVALID, "calc_nearest_index: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl), ii = hc.add(ind + a0, ind + a0, 4 * nl), ie = hc.add(nullptr, e ? e + a0 : nullptr, 4 * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_calc_nearest_index_batch_dev((int)nl, hc.p<float>(is), &dc, hc.p<int>(ii), hc.p<float>(ie), hc.stream()));
    return hc.finish();
  });
}

int crx_lqr_steering_control_batch(int n, int dim, const float* state, const crx_course* course, int* ind, float* pe,
                                   float* pth_e, const crx_lqr_params* prm, float* control) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || (n && (!state || !pe || !pth_e || !control)))
    return fail(CRX_ERR_INVALID, "lqr_steering_control: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nc = dim == 5 ? 2 : 1;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl);
    const int ii = hc.add(ind ? ind + a0 : nullptr, ind ? ind + a0 : nullptr, 4 * nl, true);
    const int ip = hc.add(pe + a0, pe + a0, 4 * nl), it = hc.add(pth_e + a0, pth_e + a0, 4 * nl);
    const int ic = hc.add(nullptr, control + nc * a0, 4 * nc * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_lqr_steering_control_batch_dev((int)nl, dim, hc.p<float>(is), &dc, hc.p<int>(ii), hc.p<float>(ip), hc.p<float>(it), prm,
                                               hc.p<float>(ic), hc.stream()));
    return hc.finish();
  });
}

int crx_update_batch(int n, float* state, const float* a, const float* delta, const crx_vehicle_params* prm) {
  CRX_TRACE();
  if (n < 0 || (n && (!state || !a || !delta))) return fail(CRX_ERR_INVALID, "update: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc;
    CRX_TRY(hc.open());
    const int is = hc.add(state + 4 * a0, state + 4 * a0, 16 * nl), ia = hc.add(a + a0, nullptr, 4 * nl), id = hc.add(delta + a0, nullptr, 4 * nl);
    CRX_TRY(hc.commit());
    CRX_TRY(crx_update_batch_dev((int)nl, hc.p<float>(is), hc.p<float>(ia), hc.p<float>(id), prm, hc.stream()));
    return hc.finish();
  });
}

int crx_lqr_closed_loop_batch(int n, int dim, float* state, const crx_course* course, float* pe, float* pth_e, int* ind,
                              const crx_lqr_params* prm, const crx_vehicle_params* veh, const crx_loop_params* loop,
                              float* traj_hist, int* ticks_done) {
  CRX_TRACE();
  if (n < 0 || (dim != 4 && dim != 5) || !course_ok(course, true) || !loop || loop->max_ticks < 0 || (n && !state))
    return fail(CRX_ERR_INVALID, "lqr_closed_loop: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nn = (size_t)n, mt = (size_t)loop->max_ticks;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    hc.forbid_zero_copy();
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, state + 4 * a0, 16 * nl);
    const int ip = hc.add(pe ? pe + a0 : nullptr, pe ? pe + a0 : nullptr, 4 * nl, true);
    const int it = hc.add(pth_e ? pth_e + a0 : nullptr, pth_e ? pth_e + a0 : nullptr, 4 * nl, true);
    const int ii = hc.add(ind ? ind + a0 : nullptr, ind ? ind + a0 : nullptr, 4 * nl, true);
    const int ik = hc.add(nullptr, ticks_done ? ticks_done + a0 : nullptr, 4 * nl);
    // the trajectory is time-major [tick][n][4]: a shard's columns, cleared first (agents that reach the goal stop writing)
    const int ih = traj_hist ? hc.add2d(nullptr, traj_hist + 4 * a0, 16 * nl, mt, 16 * nn, true) : -1;
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_lqr_closed_loop_batch_dev((int)nl, dim, hc.p<float>(is), &dc, hc.p<float>(ip), hc.p<float>(it), hc.p<int>(ii), prm, veh,
                                          loop, ih >= 0 ? hc.p<float>(ih) : nullptr, hc.p<int>(ik), hc.stream()));
    return hc.finish();
  });
}

int crx_calc_nearest_index_window_batch(int n, const float* state, const crx_course* course, const int* pind, int nsearch,
                                        int* ind_out) {
  CRX_TRACE();
  if (n < 0 || nsearch < 0 || !course_ok(course, false) || (n && (!state || !pind || !ind_out)))
    return fail(CRX_ERR_INVALID, "calc_nearest_index(window): bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl), ip = hc.add(pind + a0, nullptr, 4 * nl), io = hc.add(nullptr, ind_out + a0, 4 * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_calc_nearest_index_window_batch_dev((int)nl, hc.p<float>(is), &dc, hc.p<int>(ip), nsearch, hc.p<int>(io), hc.stream()));
    return hc.finish();
  });
}

int crx_mpc_closed_loop_batch(int n, int T, float* state, const crx_course* course, float dl, int nsearch, const crx_mpc_params* prm,
                              const crx_loop_params* loop, int* target_ind, float* traj_hist, int* ticks_done, int* solve_flags) {
  CRX_TRACE();
  if (n < 0 || T < 2 || T > 64 || !course_ok(course, true) || !loop || loop->max_ticks < 0 || (n && !state))
    return fail(CRX_ERR_INVALID, "mpc_closed_loop: bad argument (2 <= T <= 64)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0, nn = (size_t)n, mt = (size_t)loop->max_ticks;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    hc.forbid_zero_copy();
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, state + 4 * a0, 16 * nl);
    const int ii = hc.add(target_ind ? target_ind + a0 : nullptr, target_ind ? target_ind + a0 : nullptr, 4 * nl, true);
    const int ik = hc.add(nullptr, ticks_done ? ticks_done + a0 : nullptr, 4 * nl);
    const int iflag = hc.add(nullptr, solve_flags ? solve_flags + a0 : nullptr, 4 * nl);
    const int ih = traj_hist ? hc.add2d(nullptr, traj_hist + 4 * a0, 16 * nl, mt, 16 * nn, true) : -1;
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_mpc_closed_loop_flags_batch_dev((int)nl, T, hc.p<float>(is), &dc, dl, nsearch, prm, loop, hc.p<int>(ii),
                                                ih >= 0 ? hc.p<float>(ih) : nullptr, hc.p<int>(ik), hc.p<int>(iflag), hc.stream()));
    return hc.finish();
  });
}

int crx_calc_ref_trajectory_batch(int n, int T, const float* state, const crx_course* course, float dl, double dt, int nsearch,
                                  int* target_ind, float* xref) {
  CRX_TRACE();
  if (n < 0 || T < 1 || nsearch < 0 || !course_ok(course, true) || (n && (!state || !target_ind || !xref)))
    return fail(CRX_ERR_INVALID, "calc_ref_trajectory: bad argument");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  return run_sharded(n, [&](const crxh::Shard& sh) -> int {
    const size_t a0 = sh.a0, nl = sh.a1 - sh.a0;
    HostCall hc; CallCourse cc;
    CRX_TRY(hc.open());
    cc.add(hc, course);
    const int is = hc.add(state + 4 * a0, nullptr, 16 * nl), ii = hc.add(target_ind + a0, target_ind + a0, 4 * nl);
    const int ix = hc.add(nullptr, xref + 4 * (size_t)T * a0, 16 * (size_t)T * nl);
    CRX_TRY(hc.commit());
    const crx_course dc = cc.dev(hc);
    CRX_TRY(crx_calc_ref_trajectory_batch_dev((int)nl, T, hc.p<float>(is), &dc, dl, dt, nsearch, hc.p<int>(ii), hc.p<float>(ix), hc.stream()));
    return hc.finish();
  });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// particle filter
// ---------------------------------------------------------------------------------------------
extern "C" {

void crx_pf_default_params(crx_pf_params* p) {
  if (!p) return;
  p->rsim0 = 1.0 * 1.0;
  p->rsim1 = (float)(30.0 / 180.0 * 3.141592653 * 30.0 / 180.0 * 3.141592653);
  p->Q = 0.01f;
  p->dt = 0.1;
  p->nth = 0.0f;
}

int crx_pf_run_batch_dev(int n, int np, int T, int L, float* px, float* pw, float* xEst, float* PEst, const float* obs,
                         const int* nobs, const float* u, const float* nrm, const float* uni, const crx_pf_params* prm,
                         float* x_hist, int* n_resampled, void* stream) {
  CRX_TRACE();
  if (n < 0 || T < 0 || L < 0 || (np != 100 && np != 64 && np != 128) ||
      (n && (!px || !pw || !xEst || !PEst)) || (n && T && (!nobs || !u || !nrm || !uni || (L && !obs))))
    return fail(CRX_ERR_INVALID, "pf_run: bad argument (np must be 64, 100 or 128)");
  if (int rc = check_device()) return rc;
  if (n == 0 || T == 0) return CRX_OK;                  // no tick: px, pw, xEst, PEst stay as they are
  crx_pf_params q;
  if (prm) q = *prm; else crx_pf_default_params(&q);
  const crx::PfParams p{q.rsim0, q.rsim1, q.Q, q.dt, q.nth > 0.0f ? q.nth : (float)(np / 2)};
  const dim3 grid(blocks_for(n, crx::kPfWavesPerBlock)), block(64 * crx::kPfWavesPerBlock);
  hipStream_t s = (hipStream_t)stream;
  if (np == 100) hipLaunchKernelGGL((crx::pf_run_kernel<100>), grid, block, 0, s, n, T, L, px, pw, xEst, PEst, obs, nobs, u, nrm, uni, p, x_hist, n_resampled);
  else if (np == 64) hipLaunchKernelGGL((crx::pf_run_kernel<64>), grid, block, 0, s, n, T, L, px, pw, xEst, PEst, obs, nobs, u, nrm, uni, p, x_hist, n_resampled);
  else hipLaunchKernelGGL((crx::pf_run_kernel<128>), grid, block, 0, s, n, T, L, px, pw, xEst, PEst, obs, nobs, u, nrm, uni, p, x_hist, n_resampled);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// dynamic-window planner
// ---------------------------------------------------------------------------------------------
extern "C" {

void crx_dwa_default_config(crx_dwa_config* c) {
  if (!c) return;
  const double PI_ = 3.141592653;   // `#define PI 3.141592653` (src/dynamic_window_approach.cpp:16)
  c->max_speed = 1.0; c->min_speed = -0.5; c->max_yawrate = 40.0 * PI_ / 180.0; c->max_accel = 0.2; c->robot_radius = 1.0;
  c->max_dyawrate = 40.0 * PI_ / 180.0; c->v_reso = 0.01; c->yawrate_reso = 0.1 * PI_ / 180.0; c->dt = 0.1; c->predict_time = 3.0;
  c->to_goal_cost_gain = 1.0; c->speed_cost_gain = 1.0;
}

int crx_dwa_run_batch_dev(int n, int max_ticks, float* state, float* u, const float* goal, const float* ob, int nob,
                          const crx_dwa_config* cfg, float* traj_hist, int* ticks_done, int* status, int* best_idx,
                          int* n_samples, void* stream) {
  CRX_TRACE();
  if (n < 0 || max_ticks < 0 || nob < 0 || nob > crx::kDwaMaxOb || (nob && !ob) || (n && (!state || !u || !goal)))
    return fail(CRX_ERR_INVALID, "dwa_run: bad argument (nob <= 256)");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx_dwa_config q;
  if (cfg) q = *cfg; else crx_dwa_default_config(&q);
  if (!(q.v_reso > 0.0f) || !(q.yawrate_reso > 0.0f) || !(q.dt > 0.0f)) return fail(CRX_ERR_INVALID, "dwa_run: resolutions and dt must be positive");
  crx::DwaCfg c;
  static_assert(sizeof(c) == sizeof(q), "config layouts must agree");
  std::memcpy(&c, &q, sizeof(c));
  hipLaunchKernelGGL(crx::dwa_run_kernel, dim3(blocks_for(n, crx::kDwaWavesPerBlock)), dim3(64 * crx::kDwaWavesPerBlock), 0,
                     (hipStream_t)stream, n, max_ticks, state, u, goal, ob, nob, c, traj_hist, ticks_done, status, best_idx, n_samples);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------
// Frenet optimal-trajectory planner
// ---------------------------------------------------------------------------------------------
namespace {

// host mirror of cubic_spline.h's Spline: coefficients of one coordinate over the knots s (cubic_spline.h:53-65,:95-116)
void spline1d_build(const float* x, const float* y, int nx, float* a, float* b, float* c, float* d) {
  std::vector<float> h(nx - 1);
  for (int i = 1; i < nx; ++i) h[i - 1] = x[i] - x[i - 1];
  for (int i = 0; i < nx; ++i) a[i] = y[i];
  // calc_A :95-109 and calc_B :110-116 as the reference fills them (float entries), then A.colPivHouseholderQr().solve(B) :56
  std::vector<float> A((size_t)nx * nx, 0.0f), B(nx, 0.0f), sol(nx, 0.0f);
  auto at = [&](int i, int j) -> float& { return A[i + (size_t)nx * j]; };
  at(0, 0) = 1;
  for (int i = 0; i < nx - 1; ++i) {
    if (i != nx - 2) at(i + 1, i + 1) = 2 * (h[i] + h[i + 1]);
    at(i + 1, i) = h[i];
    at(i, i + 1) = h[i];
  }
  at(0, 1) = 0.0;
  at(nx - 1, nx - 2) = 0.0;
  at(nx - 1, nx - 1) = 1.0;
  for (int i = 0; i < nx - 2; ++i) B[i + 1] = (float)(3.0 * (a[i + 2] - a[i + 1]) / h[i + 1] - 3.0 * (a[i + 1] - a[i]) / h[i]);
  crx::colpiv_qr_solve<crx::kFrMaxKnots>(nx, A.data(), B.data(), sol.data());
  for (int i = 0; i < nx; ++i) c[i] = sol[i];
  for (int i = 0; i < nx - 1; ++i) {
    d[i] = (float)((c[i + 1] - c[i]) / (3.0 * h[i]));
    b[i] = (float)((a[i + 1] - a[i]) / h[i] - h[i] * (c[i + 1] + 2 * c[i]) / 3.0);
  }
  b[nx - 1] = 0.0f; d[nx - 1] = 0.0f;
}

int host_bisect(const float* x, float t, int start, int end) {   // cubic_spline.h:118-127
  for (;;) {
    const int mid = (start + end) / 2;
    if (t == x[mid] || end - start <= 1) return mid;
    if (t > x[mid]) start = mid; else end = mid;
  }
}

struct FrenetGrid { int ndi, nTi, ntv, ntt, min_nt; std::vector<float> ts, Tis; };
FrenetGrid frenet_grid(const crx_frenet_config& g) {   // the loop trip counts of :55-56,:58,:66-68
  FrenetGrid r{0, 0, 0, 0, 1 << 30, {}, {}};
  const int cap = 1 << 16;
  for (float di = (float)(-1 * g.max_road_width); di < g.max_road_width && r.ndi < cap; di += g.d_road_w) ++r.ndi;
  float Tmax = 0.0f;
  std::vector<float> Tis;
  for (float Ti = (float)g.mint; Ti < g.maxt && r.nTi < cap; Ti += g.dt) { ++r.nTi; Tmax = Ti; Tis.push_back(Ti); }
  for (float tv = (float)(g.target_speed - g.d_t_s * g.n_s_sample); tv < g.target_speed + g.d_t_s * g.n_s_sample && r.ntv < cap; tv += g.d_t_s) ++r.ntv;
  std::vector<float> ts;
  for (float t = 0; t < Tmax && r.ntt < cap; t += g.dt) { ++r.ntt; ts.push_back(t); }
  for (float Ti : Tis) { int c = 0; while (c < r.ntt && ts[c] < Ti) ++c; if (c < r.min_nt) r.min_nt = c; }
  r.ts = ts; r.Tis = Tis;
  return r;
}
int frenet_check_cfg(const crx_frenet_config& q, FrenetGrid* out) {
  if (!(q.dt > 0.0) || !(q.d_road_w > 0.0) || !(q.d_t_s > 0.0))
    return fail(CRX_ERR_INVALID, "frenet: dt, d_road_w and d_t_s must be positive");
  const FrenetGrid gr = frenet_grid(q);
  if (gr.ndi < 1 || gr.nTi < 1 || gr.ntv < 1) return fail(CRX_ERR_INVALID, "frenet: the configuration generates no candidate path");
  if (gr.ndi > crx::kFrMaxDi || gr.nTi > crx::kFrMaxTi || gr.ntv > crx::kFrMaxTv || gr.ntt > crx::kFrMaxT ||
      gr.nTi * gr.ntv > crx::kFrMaxCombos || gr.ndi * gr.nTi * gr.ntv > crx::kFrMaxPaths ||
      (size_t)gr.nTi * gr.ntv * gr.ntt * sizeof(crx::FrTab) > (size_t)crx::kFrTabLdsBytes)
    return fail(CRX_ERR_INVALID, "frenet: sample grid too large (<= 64 offsets, horizons x speeds <= 64, <= 64 time steps, "
                                 "horizons x speeds x time steps <= 2048)");
  if (gr.min_nt < 2) return fail(CRX_ERR_INVALID, "frenet: every horizon needs at least two time steps (mint > dt)");
  if (out) *out = gr;
  return CRX_OK;
}

}  // namespace

extern "C" {

void crx_frenet_default_config(crx_frenet_config* c) {
  if (!c) return;
  c->max_speed = 50.0 / 3.6; c->max_accel = 2.0; c->max_curvature = 1.0; c->max_road_width = 7.0; c->d_road_w = 1.0;
  c->dt = 0.2; c->maxt = 5.0; c->mint = 4.0; c->target_speed = 30.0 / 3.6; c->d_t_s = 5.0 / 3.6; c->n_s_sample = 1;
  c->robot_radius = 1.5; c->kj = 0.1; c->kt = 0.1; c->kd = 1.0; c->klat = 1.0; c->klon = 1.0;
}

int crx_frenet_num_paths(const crx_frenet_config* cfg) {
  crx_frenet_config q;
  if (cfg) q = *cfg; else crx_frenet_default_config(&q);
  FrenetGrid gr;
  if (int rc = frenet_check_cfg(q, &gr)) return rc;
  return gr.ndi * gr.nTi * gr.ntv;
}

int crx_frenet_spline_build(const float* wx, const float* wy, int nx, float* coef) {
  CRX_TRACE();
  if (!wx || !wy || !coef || nx < 2 || nx > crx::kFrMaxKnots) return fail(CRX_ERR_INVALID, "frenet_spline_build: bad argument (2 <= nx <= 64)");
  float* s = coef;
  s[0] = 0.0f;                                   // Spline2D::calc_s :172-186
  float temp = 0;
  for (int i = 1; i < nx; ++i) {
    const float dx = wx[i] - wx[i - 1], dy = wy[i] - wy[i - 1];
    temp += std::sqrt(dx * dx + dy * dy);
    s[i] = temp;
    if (!(s[i] > s[i - 1])) return fail(CRX_ERR_INVALID, "frenet_spline_build: consecutive way-points must be distinct");
  }
  spline1d_build(s, wx, nx, coef + nx, coef + 2 * nx, coef + 3 * nx, coef + 4 * nx);
  spline1d_build(s, wy, nx, coef + 5 * nx, coef + 6 * nx, coef + 7 * nx, coef + 8 * nx);
  return CRX_OK;
}

int crx_frenet_course_samples(const float* coef, int nx, float* rx, float* ry, int cap) {
  CRX_TRACE();
  if (!coef || nx < 2 || cap < 0 || (cap && (!rx || !ry))) return fail(CRX_ERR_INVALID, "frenet_course_samples: bad argument");
  const float* s = coef;
  int k = 0;
  for (float i = 0; i < s[nx - 1]; i += 0.1) {   // main :205-213
    if (k < cap) {
      const int seg = host_bisect(s, i, 0, nx);
      const float dx = i - s[seg];
      rx[k] = coef[nx + seg] + coef[2 * nx + seg] * dx + coef[3 * nx + seg] * dx * dx + coef[4 * nx + seg] * dx * dx * dx;
      ry[k] = coef[5 * nx + seg] + coef[6 * nx + seg] * dx + coef[7 * nx + seg] * dx * dx + coef[8 * nx + seg] * dx * dx * dx;
    }
    ++k;
  }
  return k;
}

// The course the reference's LQR / MPC mains build from their way-points: Spline2D(wx, wy) sampled every `ds`
// (src/lqr_speed_steer_control.cpp:252-265 with ds = 0.1, src/model_predictive_control.cpp:473-486 with ds = 1.0): position
// (calc_postion), heading (calc_yaw = atan2 of the first derivatives) and curvature (calc_curvature) per sample.  Host, once
// per course.  Returns the number of samples; fills up to cap of each non-null array.
int crx_course_from_waypoints(const float* wx, const float* wy, int nx, double ds, float* cx, float* cy, float* cyaw, float* ck, int cap) {
  CRX_TRACE();
  if (!wx || !wy || nx < 2 || nx > crx::kFrMaxKnots || !(ds > 0.0) || cap < 0) return fail(CRX_ERR_INVALID, "course_from_waypoints: bad argument");
  std::vector<float> coef(9 * (size_t)nx);
  if (int rc = crx_frenet_spline_build(wx, wy, nx, coef.data())) return rc;
  const float* s = coef.data();
  const float *ax = s + nx, *bx = s + 2 * nx, *cxx = s + 3 * nx, *dx_ = s + 4 * nx, *ay = s + 5 * nx, *by = s + 6 * nx, *cyy = s + 7 * nx, *dy_ = s + 8 * nx;
  if (!((float)((double)s[nx - 1] + ds) > s[nx - 1])) return fail(CRX_ERR_INVALID, "course_from_waypoints: ds too small for this course (the float walk would not advance)");
  int k = 0;
  for (float i = 0; i < s[nx - 1]; i += ds) {                      // float i += double literal, as the mains write it
    if (k < cap) {
      const int seg = host_bisect(s, i, 0, nx), segd = host_bisect(s, i, 0, nx - 1);   // calc / calc_dd use bisect(t,0,nx), calc_d bisect(t,0,nx-1)
      const float e = i - s[seg], ed = i - s[segd];
      if (cx) cx[k] = ax[seg] + bx[seg] * e + cxx[seg] * e * e + dx_[seg] * e * e * e;
      if (cy) cy[k] = ay[seg] + by[seg] * e + cyy[seg] * e * e + dy_[seg] * e * e * e;
      const float d1x = bx[segd] + 2 * cxx[segd] * ed + 3 * dx_[segd] * ed * ed;
      const float d1y = by[segd] + 2 * cyy[segd] * ed + 3 * dy_[segd] * ed * ed;
      if (cyaw) cyaw[k] = std::atan2(d1y, d1x);
      if (ck) {
        const float ddx = 2 * cxx[seg] + 6 * dx_[seg] * e, ddy = 2 * cyy[seg] + 6 * dy_[seg] * e;
        ck[k] = (ddy * d1x - ddx * d1y) / (d1x * d1x + d1y * d1y);
      }
    }
    ++k;
  }
  return k;
}

// calc_speed_profile of the two tracking files.  variant 5 (src/lqr_speed_steer_control.cpp:40-62): direction flips where the
// heading jumps by pi/4..pi/2, zero at the switch points, then the last 39 entries ramp down as target/(50-k) with a floor of
// 1/3.6 — the reference's k = 0 pass writes one element PAST the end of the vector (:55-56); that write is not made here.
// variant 0 (src/model_predictive_control.cpp:83-105): sign from the direction of travel against the heading; the reference's
// `speed_profile[-1] = 0.0` (:102) writes BEFORE the vector, so the last entry keeps its value, as here.
int crx_calc_speed_profile(int variant, const float* rx, const float* ry, const float* ryaw, int n, float target_speed, float* sp) {
  CRX_TRACE();
  if ((variant != 0 && variant != 4 && variant != 5) || n < 1 || !ryaw || !sp || (variant == 0 && (!rx || !ry)))
    return fail(CRX_ERR_INVALID, "calc_speed_profile: bad argument (variant 5, 4 or 0)");
  for (int i = 0; i < n; ++i) sp[i] = target_speed;
  float direction = 1.0;
  if (variant == 5 || variant == 4) {
    for (int i = 0; i + 1 < n; ++i) {
      const float dyaw = std::abs(ryaw[i + 1] - ryaw[i]);
      const float switch_point = (M_PI / 4.0 < dyaw) && (dyaw < M_PI / 2.0);
      if (switch_point) direction = direction * -1;
      if (direction != 1.0) sp[i] = target_speed * -1; else sp[i] = target_speed;
      if (switch_point) sp[i] = 0.0;
    }
    if (variant == 5) {
      for (int k = 1; k < 40 && k <= n; ++k) {          // :55-60 (its k = 0 writes past the end and is not made)
        sp[n - k] = target_speed / (50 - k);
        if (sp[n - k] <= 1.0 / 3.6) sp[n - k] = 1.0 / 3.6;
      }
    } else {
      sp[n - 1] = 0.0;                                  // src/lqr_steer_control.cpp:50
    }
  } else {
    for (int i = 0; i + 1 < n; ++i) {
      const float dx = rx[i + 1] - rx[i], dy = ry[i + 1] - ry[i];
      const float move_direction = std::atan2(dy, dx);
      if (dx != 0.0 && dy != 0.0) {
        const double a = (double)(move_direction - ryaw[i]);
        const float dangle = std::abs((float)(std::fmod(std::fmod(a + M_PI, 2 * M_PI) - 2 * M_PI, 2 * M_PI) + M_PI));   // YAW_P2P, motion_model.h:18
        if (dangle >= M_PI / 4.0) direction = -1.0; else direction = 1.0;
      }
      if (direction != 1.0) sp[i] = -1 * target_speed; else sp[i] = target_speed;
    }
  }
  return CRX_OK;
}

int crx_smooth_yaw(float* cyaw, int n) {   // src/model_predictive_control.cpp:172-185
  if (n < 0 || (n && !cyaw)) return fail(CRX_ERR_INVALID, "smooth_yaw: bad argument");
  for (int i = 0; i + 1 < n; ++i) {
    float dyaw = cyaw[i + 1] - cyaw[i];
    if (!std::isfinite(dyaw)) return fail(CRX_ERR_INVALID, "smooth_yaw: non-finite heading");
    while (dyaw > M_PI / 2.0) {
      const float before = cyaw[i + 1];
      cyaw[i + 1] -= M_PI * 2.0;
      if (cyaw[i + 1] == before) return fail(CRX_ERR_INVALID, "smooth_yaw: heading too large to unwind in float");
      dyaw = cyaw[i + 1] - cyaw[i];
    }
    while (dyaw < -M_PI / 2.0) {
      const float before = cyaw[i + 1];
      cyaw[i + 1] += M_PI * 2.0;
      if (cyaw[i + 1] == before) return fail(CRX_ERR_INVALID, "smooth_yaw: heading too large to unwind in float");
      dyaw = cyaw[i + 1] - cyaw[i];
    }
  }
  return CRX_OK;
}

// Probe of the device's double sin / cos (crx_dsincos.h, the table staged in LDS as the Frenet kernel does): c[i] = cos(x[i]),
// s[i] = sin(x[i]).  tests/test_dsincos.py compares the bits with the host libm's.
namespace crx {
__global__ void __launch_bounds__(256) dsincos_probe_kernel(int n, const double* __restrict__ x, double* __restrict__ s, double* __restrict__ c) {
  __shared__ uint64_t s_sc[kDsincosTabLen];
  for (int i = threadIdx.x; i < kDsincosTabLen; i += blockDim.x) s_sc[i] = kDsincosTab[i];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  c[i] = dcos_(v, s_sc);
  s[i] = dsin_(v, s_sc);
}
}  // namespace crx
int crx_x_dsincos_dev(int n, const double* x, double* s, double* c, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n > 0 && (!x || !s || !c))) return fail(CRX_ERR_INVALID, "dsincos: bad arguments");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::dsincos_probe_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, x, s, c);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// Probes of the device's double atan2(y, 1.0) (crx_datan2.h).  tests/test_datan2.py compares the bits with the host libm's:
// samples through crx_x_datan2_dev, all 2^32 float curvatures through the block checksums of crx_x_datan2_sweep_dev.
namespace crx {
__global__ void __launch_bounds__(256) datan2_probe_kernel(int n, const double* __restrict__ y, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = datan2_one_(y[i]);
}
__global__ void __launch_bounds__(256) datan2_sweep_kernel(double L, unsigned long long* __restrict__ sums, unsigned long long* __restrict__ ocml_diff,
                                                           unsigned* __restrict__ diff_k) {
  const unsigned base = blockIdx.x << 20;
  unsigned long long sum = 0, dd = 0, df = 0;
  for (unsigned i = 0; i < 4096; ++i) {
    const unsigned w = base + i * 256u + threadIdx.x;
    const double y = L * (double)__uint_as_float(w);
    const double a = datan2_one_(y);
    sum += (a != a) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(a);   // one pattern for every NaN
    const double o = atan(y);                                                                 // OCML's: what rounds 1-3 evaluated here
    if (!(o != o && a != a)) {
      dd += __double_as_longlong(o) != __double_as_longlong(a);
      const bool fd = __float_as_uint((float)o) != __float_as_uint((float)a);
      df += fd;
      if (fd && diff_k) { const unsigned long long slot = atomicAdd(&ocml_diff[2], 1ull); if (slot < 64) diff_k[slot] = w; }
    }
  }
  atomicAdd(&sums[blockIdx.x], sum);
  if (ocml_diff) { if (dd) atomicAdd(&ocml_diff[0], dd); if (df) atomicAdd(&ocml_diff[1], df); }
}
}  // namespace crx
int crx_x_datan2_dev(int n, const double* y, double* out, void* stream) {
  CRX_TRACE();
  if (n < 0 || (n > 0 && (!y || !out))) return fail(CRX_ERR_INVALID, "datan2: bad arguments");
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  hipLaunchKernelGGL(crx::datan2_probe_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, y, out);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}
int crx_x_datan2_sweep_dev(double L, unsigned long long* sums, unsigned long long* ocml_diff, unsigned* diff_k, void* stream) {
  CRX_TRACE();
  if (!sums || (diff_k && !ocml_diff)) return fail(CRX_ERR_INVALID, "datan2_sweep: bad arguments");
  if (int rc = check_device()) return rc;
  CRX_HIP(hipMemsetAsync(sums, 0, 4096 * sizeof(unsigned long long), (hipStream_t)stream));
  if (ocml_diff) CRX_HIP(hipMemsetAsync(ocml_diff, 0, 3 * sizeof(unsigned long long), (hipStream_t)stream));
  if (diff_k) CRX_HIP(hipMemsetAsync(diff_k, 0, 64 * sizeof(unsigned), (hipStream_t)stream));
  hipLaunchKernelGGL(crx::datan2_sweep_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, L, sums, ocml_diff, diff_k);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

// HBM calibration (scripts/gpu_hbm_calib.py): what a plain streaming kernel reaches on this box, next to the 8 TB/s the rooflines are
// priced against.  mode 0: dst = src (read + write), 1: read only (a word per workgroup written), 2: write only, 3: dst += 1 in place
// (read and write of the same lines: the single-step EKF's traffic shape).  16 bytes per lane per access, grid-stride.
namespace crx {
__global__ void __launch_bounds__(256) hbm_stream_kernel(int mode, size_t n16, v4f* __restrict__ dst, const v4f* __restrict__ src) {
  // mode + 8: workgroup b works where workgroup (b % 8) * (gridDim / 8) + b / 8 would — consecutive workgroups go to the eight XCDs
  // in turn, so this hands every XCD one contiguous eighth of the buffer instead of every eighth 4-KiB piece
  unsigned b = blockIdx.x;
  if (mode >= 8) { mode -= 8; b = (b & 7u) * (gridDim.x >> 3) + (b >> 3); }
  const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)b * 256 + threadIdx.x;
  if (mode == 0) {
    for (size_t i = i0; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  } else if (mode == 1) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = i0; i < n16; i += stride) acc += __builtin_nontemporal_load(src + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[blockIdx.x] = acc;     // keeps the loads alive; practically never true
  } else if (mode == 2) {
    for (size_t i = i0; i < n16; i += stride) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, dst + i);
  } else {
    for (size_t i = i0; i < n16; i += stride) dst[i] = dst[i] + v4f{1.f, 1.f, 1.f, 1.f};
  }
}
}  // namespace crx
int crx_x_hbm_stream_dev(int mode, void* dst, const void* src, size_t bytes, int workgroups, void* stream) {
  CRX_TRACE();
  if (mode < 0 || (mode & 7) > 3 || mode > 11 || !dst || (((mode & 7) == 0 || (mode & 7) == 1) && !src) || bytes % 16 || workgroups < 1 ||
      (mode >= 8 && workgroups % 8))
    return fail(CRX_ERR_INVALID, "hbm_stream: bad arguments (bytes a multiple of 16; mode + 8 needs a multiple of 8 workgroups)");
  if (int rc = check_device()) return rc;
  const size_t n16 = bytes / 16;
  auto* d = (crx::v4f*)dst; auto* sp = (const crx::v4f*)src;
  const dim3 g((unsigned)workgroups), b(256);
  hipLaunchKernelGGL(crx::hbm_stream_kernel, g, b, 0, (hipStream_t)stream, mode, n16, d, sp);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

int crx_frenet_run_batch_dev(int n, int max_ticks, float* state, const float* coef, int nx, const float* goal_xy,
                             const float* ob, int nob, const crx_frenet_config* cfg, float* hist, int* ticks_done,
                             int* status, int* best_idx, int* n_valid, float* path_cf, int* path_ok, int path_cap,
                             void* stream) {
  CRX_TRACE();
  if (n < 0 || max_ticks < 0 || nob < 0 || nob > crx::kFrMaxOb || (nob && !ob) || nx < 2 || nx > crx::kFrMaxKnots || !coef ||
      !goal_xy || path_cap < 0 || (n && !state))
    return fail(CRX_ERR_INVALID, "frenet_run: bad argument (2 <= nx <= 64, nob <= 128)");
  crx_frenet_config q;
  if (cfg) q = *cfg; else crx_frenet_default_config(&q);
  FrenetGrid gr;
  if (int rc = frenet_check_cfg(q, &gr)) return rc;
  if (int rc = check_device()) return rc;
  if (n == 0) return CRX_OK;
  crx::FrenetCfg c;
  static_assert(sizeof(c) == sizeof(q), "config layouts must agree");
  std::memcpy(&c, &q, sizeof(c));
  // the per-wave (combo, time step) table lives in dynamic LDS: as many waves per block as fit beside it
  const int stride = gr.nTi * gr.ntv * gr.ntt;
  int wpb = crx::kFrWavesPerBlock;
  while (wpb > 1 && (size_t)wpb * stride * sizeof(crx::FrTab) > (size_t)crx::kFrTabLdsBytes) wpb >>= 1;
  // std::pow(t, k), k = 2..5, of the time grid and of the horizons: libm's own values, what the reference's polynomial classes
  // call (quintic_polynomial.h:41-68, quartic_polynomial.h:39-59) — float argument and int exponent promoted to double
  crx::FrPowArg pw;
  std::memset(&pw, 0, sizeof(pw));
  auto powers = [](float x) { return crx::FrPow{std::pow((double)x, 2.0), std::pow((double)x, 3.0), std::pow((double)x, 4.0), std::pow((double)x, 5.0)}; };
  for (int i = 0; i < gr.ntt; ++i) pw.t[i] = powers(gr.ts[i]);
  for (int i = 0; i < gr.nTi; ++i) pw.T[i] = powers(gr.Tis[i]);
  hipLaunchKernelGGL(crx::frenet_run_kernel, dim3(blocks_for(n, wpb)), dim3(64 * wpb), (size_t)wpb * stride * sizeof(crx::FrTab),
                     (hipStream_t)stream, n, max_ticks, state, coef, nx, goal_xy[0], goal_xy[1], ob, nob, c, hist, ticks_done,
                     status, best_idx, n_valid, path_cf, path_ok, path_cap, stride, pw);
  CRX_HIP(hipGetLastError());
  return CRX_OK;
}

}  // extern "C"
