// api_internal.inl — part of the single translation unit crx_api.hip (#included there, in this order: api_internal, api_core, api_ekf,
// api_lqr, api_mpc, api_track, api_planners, api_frenet, api_probes); what every entry point shares: error string, roctx ranges, HIP error macro, launch-geometry
// constants, the host-pointer call marshalling (HostCall) and the sharding of a host-pointer call over the device set.
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

  // an early return between commit() and finish() (a failed HIP call, a failed launch) must not leave copies in flight on the
  // context's streams when the lock goes: drained here, still under the lock (members are destroyed after this body)
  ~HostCall() { if (c_ && !finished_) c_->drain(); }
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
      finished_ = true;
      return CRX_OK;
    }
    for (auto& a : args_)
      if (a.dst && a.row_bytes * a.rows) { if (int rc = copy_out(a)) return rc; }
    CRX_HIP(hipStreamSynchronize(c_->s_cmp));
    finished_ = true;
    return CRX_OK;
  }

 private:
  crxh::DeviceCtx* c_ = nullptr;
  std::unique_lock<std::mutex> lock_;
  std::vector<Arg> args_;
  bool zc_ = false, allow_zc_ = true, finished_ = false;
  size_t slot_ = crxh::kStageChunk;
  // the two staging slots of the pinned ring are shared by every pageable argument of the call: a slot whose H2D has been queued
  // is refilled only after that DMA's event — across arguments too (the fence used to restart with every argument, so
  // the second large pageable input of a call overwrote the slot the first one's last DMA was still reading)
  bool slot_in_flight_[2] = {false, false};

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
      if (slot_in_flight_[k & 1]) CRX_HIP(hipEventSynchronize(c_->ev_tmp[k & 1]));
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
      slot_in_flight_[k & 1] = true;
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
