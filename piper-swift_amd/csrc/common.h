// common.h — internals shared by the piper_hip translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/piper_hip.h"

#define PH_EXPORT extern "C" __attribute__((visibility("default")))
#include <cstdlib>

namespace ph {

// thread-local error text behind piper_hip_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

// Tuning / A-B switches: ≈ 45 PIPER_HIP_* environment variables are read, once each, by the dispatchers. They exist
// so that every design decision can be re-measured — not so that an inherited environment can silently change which kernel runs
// (ADVICE r2). Every lookup in this library goes through tuning_getenv (the macro below replaces the C library's getenv in all of
// csrc/): a PIPER_HIP_* switch is HONOURED only when PIPER_HIP_TUNING=1 is set as well, and every switch that was honoured is
// recorded and reported by piper_hip_config_string() — bench.py prints it in its JSON line.
const char* tuning_getenv(const char* name);

#define PH_FAIL(code, ...)        \
  do {                            \
    ph::set_error(__VA_ARGS__);   \
    return (code);                \
  } while (0)

#define PH_HIP(expr, code)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) PH_FAIL(code, "%s failed: %s", #expr, hipGetErrorString(_e));       \
  } while (0)

// Every entry point that takes a context runs on THAT context's device, whatever the calling thread's current device is
// (two contexts in one process, or a context used from a second host thread).
#define PH_CHECK_CTX(ctx)                                                                                                   \
  do {                                                                                                                      \
    if (!(ctx)) PH_FAIL(PIPER_HIP_ERR_ARG, "null context");                                                                 \
    if (hipSetDevice((ctx)->device) != hipSuccess) PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "hipSetDevice(%d) failed", (ctx)->device); \
  } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Size-bucketed caching device allocator: the reference allocates a fresh MTLBuffer per op output
// (MetalBackend.swift:1184); here alloc/free recycle blocks so op-level callers never hit hipMalloc in
// steady state. Blocks are rounded up to a power of two ≥ 256 B.
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;            // ptr -> bucket bytes
  std::map<size_t, std::vector<void*>> free_blocks;  // bucket bytes -> ptrs
  // Blocks freed by the HOST while work may still be queued on the context's streams: each carries one event per stream,
  // recorded at free time; the block goes back on the free list only once all of them have completed (stream-ordered free).
  struct Pending {
    void* p;
    size_t bucket;
    std::vector<hipEvent_t> evs;
  };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_cache;
  size_t bytes_reserved = 0;
  // One slab taken from the driver up front (reserve): blocks the free lists cannot serve are carved from it before hipMalloc is asked.
  // A fresh hipMalloc is cheap on the host but its first use can stall the GPU for tens of milliseconds while the driver maps and clears
  // the pages (r3, tools/probe/request_max.py: 27 ms in the first synchronisation after a plan build, on requests worth 1 ms) — a voice
  // reserves its plan memory while it loads instead (piper_hip_memory_reserve; voice_create does it once per context).
  char* slab = nullptr;
  size_t slab_size = 0, slab_off = 0;
  bool in_slab(const void* p) const { return slab && (const char*)p >= slab && (const char*)p < slab + slab_size; }
  int reserve(size_t bytes);
  int alloc(size_t bytes, void** out);
  int release(void* p);  // immediate reuse: only for blocks whose users are known to be complete
  // stream-ordered release: reusable once everything enqueued so far on `streams` has run
  int release_after(void* p, const std::vector<hipStream_t>& streams);
  void reap(bool wait);  // move completed pending blocks to the free list (wait = block until all are complete)
  void trim();
};

// Code objects are loaded lazily, one per translation unit, when its first kernel is launched — 1 … 13 ms that land on whichever request
// first needs a kernel of a unit nothing touched before (r3, tools/probe/request_max.py: the first long request of a process took 14 ms
// instead of 2 because it was the first to leave the short-row kernels). Every unit registers one of its kernels here; voice_create asks
// the runtime for that kernel's attributes (warm_all_modules), which loads the unit while the voice loads.
struct WarmReg { explicit WarmReg(void (*f)()); };
void warm_all_modules();
#define PH_WARM(tag, kernel)                                                                              \
  static const ::ph::WarmReg warm_reg_##tag([] {                                                          \
    hipFuncAttributes at_;                                                                                \
    if (hipFuncGetAttributes(&at_, (const void*)(kernel)) != hipSuccess) (void)hipGetLastError();        \
  })

// per-device "already opted in to > 64 KiB of dynamic LDS" flags (hipFuncSetAttribute is per device)
constexpr int kMaxDevices = 64;
inline bool lds_optin_needed(bool (&flags)[kMaxDevices]) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return true;
  if (flags[d]) return false;
  flags[d] = true;
  return true;
}

}  // namespace ph

struct piper_hip_ctx {
  int device = 0;
  hipDeviceProp_t props{};
  hipStream_t default_stream = nullptr;  // used when the caller passes stream == NULL (blocking semantics)
  std::vector<hipStream_t> streams;      // default_stream + every stream handed out by piper_hip_stream_create
  ph::Pool pool;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  int num_cus = 256;
  // op-internal temporaries (packed weights, fused-op intermediates) of non-blocking calls: returned to the pool at
  // the next host-visible sync point of this context
  std::vector<void*> deferred;
};

namespace ph {

void release_deferred(piper_hip_ctx* ctx);
inline void defer_free(piper_hip_ctx* ctx, void* p) {
  if (p) ctx->deferred.push_back(p);
}

// Resolve the stream argument of an op entry point: NULL ⇒ ctx default stream + block at the end.
struct StreamScope {
  piper_hip_ctx* ctx;
  hipStream_t s;
  bool blocking;
  StreamScope(piper_hip_ctx* c, piper_hip_stream user) : ctx(c) {
    blocking = (user == nullptr);
    s = blocking ? c->default_stream : (hipStream_t)user;
  }
  // call at the end of an op: surfaces launch errors, and blocks when commandBuffer == nil semantics apply
  int finish(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    if (blocking) {
      e = hipStreamSynchronize(s);
      if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "%s: execution failed: %s", what, hipGetErrorString(e));
      if (!ctx->deferred.empty()) {
        // temporaries of earlier non-blocking calls may live on other streams of this context
        e = hipDeviceSynchronize();
        if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "%s: device sync failed: %s", what, hipGetErrorString(e));
        release_deferred(ctx);
      }
    }
    return PIPER_HIP_OK;
  }
};

// Output-buffer convention: *out == NULL ⇒ allocate `count` floats from the pool (≥ 1 byte like allocateBuffer).
int ensure_out(piper_hip_ctx* ctx, float** out, size_t count, int alloc_err_code_note);

}  // namespace ph

// after every include of this header: csrc/ code that says getenv(...) gets the gated, recorded lookup
#define getenv(name) ::ph::tuning_getenv(name)

