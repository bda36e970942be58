// context.cpp — device context, pooled buffers, transfers, streams, timers.
// Replaces Metal/MetalContext.swift:4-62 and the buffer/blit helpers of MetalBackend.swift:34-66, 841-874, 963-993.
#include <mutex>
#include "common.h"

namespace ph {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

#undef getenv
static std::mutex g_tune_mu;
static std::map<std::string, std::string> g_tune_seen;  // switches that were honoured (name → value)
static int g_tune_ignored = 0;                            // PIPER_HIP_* variables found in the environment but not honoured

const char* tuning_getenv(const char* name) {
  const char* v = ::getenv(name);
  if (!v) return nullptr;
  static const bool enabled = [] { const char* t = ::getenv("PIPER_HIP_TUNING"); return t && t[0] == '1'; }();
  std::lock_guard<std::mutex> lk(g_tune_mu);
  if (strncmp(name, "PIPER_HIP_", 10) == 0 && !enabled) {
    g_tune_ignored++;
    return nullptr;
  }
  g_tune_seen[name] = v;
  return v;
}
#define getenv(name) ::ph::tuning_getenv(name)

static std::vector<void (*)()>& warm_list() {
  static std::vector<void (*)()> v;
  return v;
}
WarmReg::WarmReg(void (*f)()) { warm_list().push_back(f); }
void warm_all_modules() {
  static std::once_flag once;
  std::call_once(once, [] { for (auto f : warm_list()) f(); });
}

static size_t bucket_of(size_t bytes) {
  size_t b = 256;
  while (b < bytes) b <<= 1;
  return b;
}

void Pool::reap(bool wait) {  // caller holds mu
  size_t keep = 0;
  for (size_t i = 0; i < pending.size(); i++) {
    Pending& pe = pending[i];
    bool done = true;
    for (hipEvent_t e : pe.evs) {
      if (wait) (void)hipEventSynchronize(e);
      else if (hipEventQuery(e) != hipSuccess) { done = false; break; }
    }
    if (!done) {
      (void)hipGetLastError();  // hipErrorNotReady is not an error
      if (keep != i) pending[keep] = std::move(pe);
      keep++;
      continue;
    }
    for (hipEvent_t e : pe.evs) event_cache.push_back(e);
    free_blocks[pe.bucket].push_back(pe.p);
  }
  pending.resize(keep);
}

int Pool::release_after(void* p, const std::vector<hipStream_t>& streams) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = live.find(p);
  if (it == live.end()) PH_FAIL(PIPER_HIP_ERR_ARG, "piper_hip_free: %p is not a live buffer of this context", p);
  Pending pe{p, it->second, {}};
  for (hipStream_t s : streams) {
    hipEvent_t e = nullptr;
    if (!event_cache.empty()) { e = event_cache.back(); event_cache.pop_back(); }
    else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    if (!e || hipEventRecord(e, s) != hipSuccess) {
      // cannot order the free behind this stream: fall back to a full stop (correct, slow, never expected)
      (void)hipGetLastError();
      if (e) event_cache.push_back(e);
      (void)hipDeviceSynchronize();
      for (hipEvent_t q : pe.evs) event_cache.push_back(q);
      pe.evs.clear();
      break;
    }
    pe.evs.push_back(e);
  }
  live.erase(it);
  pending.push_back(std::move(pe));
  return PIPER_HIP_OK;
}

int Pool::alloc(size_t bytes, void** out) {
  const size_t b = bucket_of(bytes < 1 ? 1 : bytes);
  std::lock_guard<std::mutex> lk(mu);
  if (!pending.empty()) reap(false);
  auto it = free_blocks.find(b);
  if (it != free_blocks.end() && !it->second.empty()) {
    *out = it->second.back();
    it->second.pop_back();
    live[*out] = b;
    return PIPER_HIP_OK;
  }
  if (slab && slab_off + b <= slab_size) {  // carve (b is a power of two ≥ 256: the offset stays 256-byte aligned)
    *out = slab + slab_off;
    slab_off += b;
    live[*out] = b;
    return PIPER_HIP_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, b);
  if (e != hipSuccess) {
    // give cached blocks back to the driver and retry once (blocks carved from the slab stay on their lists)
    (void)hipGetLastError();
    reap(true);
    for (auto& kv : free_blocks) {
      std::vector<void*> keep;
      for (void* q : kv.second) {
        if (in_slab(q)) { keep.push_back(q); continue; }
        (void)hipFree(q);
        bytes_reserved -= kv.first;
      }
      kv.second.swap(keep);
    }
    e = hipMalloc(&p, b);
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_ALLOC, "hipMalloc(%zu) failed: %s", b, hipGetErrorString(e));
  }
  bytes_reserved += b;
  live[p] = b;
  *out = p;
  return PIPER_HIP_OK;
}

int Pool::reserve(size_t bytes) {
  std::lock_guard<std::mutex> lk(mu);
  if (slab || bytes == 0) return PIPER_HIP_OK;  // one slab per context; a second call is a no-op
  void* p = nullptr;
  size_t want = (bytes + 255) & ~(size_t)255;
  hipError_t e = hipMalloc(&p, want);
  while (e != hipSuccess && want > ((size_t)64 << 20)) {  // a smaller card: halve until it fits
    (void)hipGetLastError();
    want >>= 1;
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) { (void)hipGetLastError(); return PIPER_HIP_OK; }  // not fatal: allocations fall through to hipMalloc
  // touch it now: the driver maps (and clears) the pages here, not under the first request that lands on them
  if (hipMemset(p, 0, want) != hipSuccess || hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
  slab = (char*)p;
  slab_size = want;
  slab_off = 0;
  bytes_reserved += want;
  return PIPER_HIP_OK;
}

int Pool::release(void* p) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = live.find(p);
  if (it == live.end()) PH_FAIL(PIPER_HIP_ERR_ARG, "piper_hip_free: %p is not a live buffer of this context", p);
  free_blocks[it->second].push_back(p);
  live.erase(it);
  return PIPER_HIP_OK;
}

void Pool::trim() {
  std::lock_guard<std::mutex> lk(mu);
  reap(true);
  for (hipEvent_t e : event_cache) (void)hipEventDestroy(e);
  event_cache.clear();
  for (auto& kv : free_blocks)
    for (void* q : kv.second)
      if (!in_slab(q)) (void)hipFree(q);
  free_blocks.clear();
  for (auto& kv : live)
    if (!in_slab(kv.first)) (void)hipFree(kv.first);
  live.clear();
  if (slab) (void)hipFree(slab);
  slab = nullptr;
  slab_size = slab_off = 0;
  bytes_reserved = 0;
}

void release_deferred(piper_hip_ctx* ctx) {
  for (void* p : ctx->deferred) (void)ctx->pool.release(p);
  ctx->deferred.clear();
}

int ensure_out(piper_hip_ctx* ctx, float** out, size_t count, int) {
  if (!out) PH_FAIL(PIPER_HIP_ERR_ARG, "null output pointer");
  if (*out) return PIPER_HIP_OK;
  void* p = nullptr;
  int rc = ctx->pool.alloc(count * sizeof(float), &p);
  if (rc) return rc;
  *out = (float*)p;
  return PIPER_HIP_OK;
}

}  // namespace ph

PH_EXPORT const char* piper_hip_last_error(void) { return ph::get_error(); }
PH_EXPORT int piper_hip_abi_version(void) { return PIPER_HIP_ABI_VERSION; }

PH_EXPORT int piper_hip_config_string(char* buf, size_t n) {
  if (!buf || n == 0) PH_FAIL(PIPER_HIP_ERR_ARG, "config_string: null buffer");
  std::string out;
  {
    std::lock_guard<std::mutex> lk(ph::g_tune_mu);
    for (auto& kv : ph::g_tune_seen) out += (out.empty() ? "" : " ") + kv.first + "=" + kv.second;
    if (ph::g_tune_ignored) out += (out.empty() ? "" : " ") + std::string("(PIPER_HIP_* variables present but ignored: set PIPER_HIP_TUNING=1 to honour them)");
  }
  snprintf(buf, n, "%s", out.c_str());
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

PH_EXPORT int piper_hip_create(int device, piper_hip_ctx** out) {
  if (!out) PH_FAIL(PIPER_HIP_ERR_ARG, "null out");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "no HIP device available (%s); this backend has no CPU fallback",
            e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  }
  if (device < 0 || device >= n) PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "device %d out of range (have %d)", device, n);
  PH_HIP(hipSetDevice(device), PIPER_HIP_ERR_UNAVAILABLE);
  piper_hip_ctx* c = new piper_hip_ctx();
  c->device = device;
  if (hipGetDeviceProperties(&c->props, device) != hipSuccess) {
    delete c;
    PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "hipGetDeviceProperties failed");
  }
  // The code objects in this library are gfx950 only (no multi-arch dispatch).
  if (strncmp(c->props.gcnArchName, "gfx950", 6) != 0) {
    std::string arch = c->props.gcnArchName;
    delete c;
    PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "device %d is %s; this library ships gfx950 (MI355X) code only", device,
            arch.c_str());
  }
  c->num_cus = c->props.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->default_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess) {
    delete c;
    PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "stream/event creation failed");
  }
  c->streams.push_back(c->default_stream);
  *out = c;
  return PIPER_HIP_OK;
}

PH_EXPORT void piper_hip_destroy(piper_hip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  ph::release_deferred(ctx);
  ctx->pool.trim();
  for (hipStream_t s : ctx->streams)
    if (s != ctx->default_stream) (void)hipStreamDestroy(s);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  if (ctx->default_stream) (void)hipStreamDestroy(ctx->default_stream);
  delete ctx;
}

PH_EXPORT int piper_hip_alloc(piper_hip_ctx* ctx, size_t bytes, void** out) {
  PH_CHECK_CTX(ctx);
  if (!out) PH_FAIL(PIPER_HIP_ERR_ARG, "null out");
  return ctx->pool.alloc(bytes, out);
}

PH_EXPORT int piper_hip_free(piper_hip_ctx* ctx, void* buf) {
  PH_CHECK_CTX(ctx);
  if (!buf) return PIPER_HIP_OK;
  // Stream-ordered (the header's promise): kernels already enqueued on any stream of this context may still read or write
  // `buf`, so it becomes reusable only after an event recorded NOW on each of them has completed. This is the reference's
  // GraphExecutor pattern — intermediates are dropped while the command buffer is still being encoded (GraphExecutor.swift:216-225).
  return ctx->pool.release_after(buf, ctx->streams);
}

PH_EXPORT int piper_hip_upload_f32(piper_hip_ctx* ctx, const float* host, size_t count, float** out) {
  PH_CHECK_CTX(ctx);
  if (!out || (!host && count)) PH_FAIL(PIPER_HIP_ERR_ARG, "null pointer");
  void* p = nullptr;
  int rc = ctx->pool.alloc(count * sizeof(float), &p);
  if (rc) return rc;
  if (count) {
    // ordered after everything already enqueued on the default stream; blocks like uploadFloat32
    hipError_t e = hipMemcpyAsync(p, host, count * sizeof(float), hipMemcpyHostToDevice, ctx->default_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->default_stream);
    if (e != hipSuccess) {
      ctx->pool.release(p);
      PH_FAIL(PIPER_HIP_ERR_LAUNCH, "upload failed: %s", hipGetErrorString(e));
    }
  }
  *out = (float*)p;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_upload_i64(piper_hip_ctx* ctx, const int64_t* host, size_t count, int64_t** out) {
  PH_CHECK_CTX(ctx);
  if (!out || (!host && count)) PH_FAIL(PIPER_HIP_ERR_ARG, "null pointer");
  void* p = nullptr;
  int rc = ctx->pool.alloc(count * sizeof(int64_t), &p);
  if (rc) return rc;
  if (count) {
    hipError_t e = hipMemcpyAsync(p, host, count * sizeof(int64_t), hipMemcpyHostToDevice, ctx->default_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->default_stream);
    if (e != hipSuccess) {
      ctx->pool.release(p);
      PH_FAIL(PIPER_HIP_ERR_LAUNCH, "upload failed: %s", hipGetErrorString(e));
    }
  }
  *out = (int64_t*)p;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_host_alloc(piper_hip_ctx* ctx, size_t bytes, void** out) {
  PH_CHECK_CTX(ctx);
  if (!out) PH_FAIL(PIPER_HIP_ERR_ARG, "host_alloc: null out");
  *out = nullptr;
  if (bytes == 0) return PIPER_HIP_OK;
  if (hipHostMalloc(out, bytes) != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    PH_FAIL(PIPER_HIP_ERR_ALLOC, "host_alloc: %zu bytes of page-locked memory not available", bytes);
  }
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_host_free(piper_hip_ctx* ctx, void* host) {
  PH_CHECK_CTX(ctx);
  if (host && hipHostFree(host) != hipSuccess) {
    (void)hipGetLastError();
    PH_FAIL(PIPER_HIP_ERR_ARG, "host_free: not a pointer returned by piper_hip_host_alloc");
  }
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_download_f32(piper_hip_ctx* ctx, const float* buf, float* host, size_t count) {
  PH_CHECK_CTX(ctx);
  if ((!buf || !host) && count) PH_FAIL(PIPER_HIP_ERR_ARG, "null pointer");
  // downloadFloat32 is a full sync point in the reference ("hydration", GraphExecutor.swift:408-467)
  PH_HIP(hipDeviceSynchronize(), PIPER_HIP_ERR_LAUNCH);
  ph::release_deferred(ctx);
  if (count) PH_HIP(hipMemcpy(host, buf, count * sizeof(float), hipMemcpyDeviceToHost), PIPER_HIP_ERR_LAUNCH);
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_stream_create(piper_hip_ctx* ctx, piper_hip_stream* out) {
  PH_CHECK_CTX(ctx);
  if (!out) PH_FAIL(PIPER_HIP_ERR_ARG, "null out");
  hipStream_t s;
  PH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), PIPER_HIP_ERR_LAUNCH);
  {
    std::lock_guard<std::mutex> lk(ctx->pool.mu);
    ctx->streams.push_back(s);
  }
  *out = (piper_hip_stream)s;
  return PIPER_HIP_OK;
}
PH_EXPORT int piper_hip_stream_destroy(piper_hip_ctx* ctx, piper_hip_stream s) {
  PH_CHECK_CTX(ctx);
  if (!s) return PIPER_HIP_OK;
  if ((hipStream_t)s == ctx->default_stream) PH_FAIL(PIPER_HIP_ERR_ARG, "stream_destroy: not a stream of piper_hip_stream_create");
  // pending frees hold events recorded on this stream: let them complete before the stream goes away
  PH_HIP(hipStreamSynchronize((hipStream_t)s), PIPER_HIP_ERR_LAUNCH);
  {
    std::lock_guard<std::mutex> lk(ctx->pool.mu);
    ctx->pool.reap(false);
    for (size_t i = 0; i < ctx->streams.size(); i++)
      if (ctx->streams[i] == (hipStream_t)s) { ctx->streams.erase(ctx->streams.begin() + i); break; }
  }
  PH_HIP(hipStreamDestroy((hipStream_t)s), PIPER_HIP_ERR_LAUNCH);
  return PIPER_HIP_OK;
}
PH_EXPORT int piper_hip_stream_sync(piper_hip_ctx* ctx, piper_hip_stream s) {
  PH_CHECK_CTX(ctx);
  // temporaries may have been used on any stream of this context: make them all quiescent before recycling
  PH_HIP(hipStreamSynchronize(s ? (hipStream_t)s : ctx->default_stream), PIPER_HIP_ERR_LAUNCH);
  if (!ctx->deferred.empty()) {
    PH_HIP(hipDeviceSynchronize(), PIPER_HIP_ERR_LAUNCH);
    ph::release_deferred(ctx);
  }
  return PIPER_HIP_OK;
}
PH_EXPORT int piper_hip_timer_begin(piper_hip_ctx* ctx, piper_hip_stream s) {
  PH_CHECK_CTX(ctx);
  PH_HIP(hipEventRecord(ctx->t0, s ? (hipStream_t)s : ctx->default_stream), PIPER_HIP_ERR_LAUNCH);
  return PIPER_HIP_OK;
}
PH_EXPORT int piper_hip_timer_end(piper_hip_ctx* ctx, piper_hip_stream s, double* gpu_ms) {
  PH_CHECK_CTX(ctx);
  PH_HIP(hipEventRecord(ctx->t1, s ? (hipStream_t)s : ctx->default_stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipEventSynchronize(ctx->t1), PIPER_HIP_ERR_LAUNCH);
  float ms = 0;
  PH_HIP(hipEventElapsedTime(&ms, ctx->t0, ctx->t1), PIPER_HIP_ERR_LAUNCH);
  if (gpu_ms) *gpu_ms = ms;
  return PIPER_HIP_OK;
}

// ---- memory accounting for long-running hosts (serving: many utterance shapes over the life of a process)
PH_EXPORT int piper_hip_memory_stats(piper_hip_ctx* ctx, size_t* reserved_bytes, size_t* live_bytes) {
  PH_CHECK_CTX(ctx);
  std::lock_guard<std::mutex> lk(ctx->pool.mu);
  size_t live = 0;
  for (auto& kv : ctx->pool.live) live += kv.second;
  if (reserved_bytes) *reserved_bytes = ctx->pool.bytes_reserved;
  if (live_bytes) *live_bytes = live;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_memory_reserve(piper_hip_ctx* ctx, size_t bytes) {
  PH_CHECK_CTX(ctx);
  PH_HIP(hipSetDevice(ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  return ctx->pool.reserve(bytes);
}

PH_EXPORT int piper_hip_memory_trim(piper_hip_ctx* ctx) {
  PH_CHECK_CTX(ctx);
  PH_HIP(hipSetDevice(ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  PH_HIP(hipDeviceSynchronize(), PIPER_HIP_ERR_LAUNCH);  // cached blocks may still be referenced by queued work
  ph::release_deferred(ctx);
  std::lock_guard<std::mutex> lk(ctx->pool.mu);
  ctx->pool.reap(true);
  for (auto& kv : ctx->pool.free_blocks) {  // blocks carved from the reserved slab stay cached: the slab is one allocation
    std::vector<void*> keep;
    for (void* q : kv.second) {
      if (ctx->pool.in_slab(q)) { keep.push_back(q); continue; }
      (void)hipFree(q);
      ctx->pool.bytes_reserved -= kv.first;
    }
    kv.second.swap(keep);
  }
  return PIPER_HIP_OK;
}

