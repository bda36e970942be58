// piper_hip_comm_*: the one collective the path has — a one-shot broadcast of the voice blob from the rank
// that parsed the .onnx to every other GPU of the node, over RCCL / xGMI — exposed through the C-ABI so that a host that is
// not Python (the reference's Swift CLI, a C++ server) needs no torch.distributed for it.
//
// RCCL is bound at RUN time (dlopen of librccl.so.1, symbols by name): the library keeps loading on machines without RCCL,
// and in a process where PyTorch has already mapped its own copy the same image is reused (same soname).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* why = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // RCCL must drive the SAME copy of the HIP runtime this library does: a process may hold two (PyTorch bundles its own
    // libamdhip64 + librccl next to ROCm's), and a librccl bound to the other copy sees "no ROCm-capable device" (r3k). So the
    // first candidates are the librccl files that sit next to the libamdhip64 our own HIP calls resolve to.
    std::string dir;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      dir = info.dli_fname;
      const size_t slash = dir.rfind('/');
      dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    std::vector<std::string> names;
    if (!dir.empty()) { names.push_back(dir + "librccl.so.1"); names.push_back(dir + "librccl.so"); }
    names.push_back("librccl.so.1"); names.push_back("librccl.so"); names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& name : names) {
      r.h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
    if (!r.h) { r.why = "librccl.so.1 not found (dlopen)"; return; }
    auto sym = [&](const char* n) { void* p = dlsym(r.h, n); if (!p && !r.why) r.why = n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return r;
}

}  // namespace

struct piper_hip_comm {
  piper_hip_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, world = 1;
  double* scratch = nullptr;  // one double on the device for the MAX reduction
};

#define PH_RCCL(expr)                                                                                            \
  do {                                                                                                           \
    ncclResult_t _r = (expr);                                                                                    \
    if (_r != ncclSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "%s failed: %s", #expr, R.GetErrorString ? R.GetErrorString(_r) : "?"); \
  } while (0)

static_assert(sizeof(ncclUniqueId) == PIPER_HIP_COMM_ID_BYTES, "PIPER_HIP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

PH_EXPORT int piper_hip_comm_unique_id(void* id_out) {
  if (!id_out) PH_FAIL(PIPER_HIP_ERR_ARG, "null id");
  Rccl& R = rccl();
  if (R.why) PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "RCCL unavailable: %s", R.why);
  ncclUniqueId id;
  PH_RCCL(R.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_comm_create(piper_hip_ctx* ctx, const void* id, int rank, int world, piper_hip_comm** out) {
  PH_CHECK_CTX(ctx);
  if (!id || !out) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (world < 1 || rank < 0 || rank >= world) PH_FAIL(PIPER_HIP_ERR_ARG, "rank %d outside world of %d", rank, world);
  Rccl& R = rccl();
  if (R.why) PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "RCCL unavailable: %s", R.why);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  auto* c = new piper_hip_comm;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  (void)hipGetLastError();  // RCCL's init trips over a stale "last error" left by an earlier, already reported HIP failure
  ncclResult_t r = R.CommInitRank(&c->comm, world, uid, rank);  // collective: every rank of the world calls it
  if (r != ncclSuccess) {
    delete c;
    PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, R.GetErrorString ? R.GetErrorString(r) : "?");
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->scratch, sizeof(double)) != hipSuccess) {
    R.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    PH_FAIL(PIPER_HIP_ERR_ALLOC, "comm: stream / scratch allocation failed");
  }
  *out = c;
  return PIPER_HIP_OK;
}

PH_EXPORT void piper_hip_comm_destroy(piper_hip_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->stream);
  Rccl& R = rccl();
  if (R.CommDestroy && c->comm) R.CommDestroy(c->comm);
  (void)hipFree(c->scratch);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

PH_EXPORT int piper_hip_comm_rank(const piper_hip_comm* c) { return c ? c->rank : -1; }

PH_EXPORT int piper_hip_comm_world(const piper_hip_comm* c) {
  if (!c) return -1;
  int n = -1;
  Rccl& R = rccl();
  if (!R.CommCount || R.CommCount(c->comm, &n) != ncclSuccess) return -1;
  return n;  // what the communicator says, not what create() was told
}

PH_EXPORT int piper_hip_comm_broadcast_f32(piper_hip_comm* c, float* device_buf, size_t count, int root) {
  if (!c || !device_buf) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (root < 0 || root >= c->world) PH_FAIL(PIPER_HIP_ERR_ARG, "root %d outside world of %d", root, c->world);
  PH_CHECK_CTX(c->ctx);
  Rccl& R = rccl();
  // in place, one call for the whole blob: a 63–113 MB message is far above the size where xGMI links are bandwidth-bound
  PH_RCCL(R.Broadcast(device_buf, device_buf, count, ncclFloat32, root, c->comm, c->stream));
  PH_HIP(hipStreamSynchronize(c->stream), PIPER_HIP_ERR_LAUNCH);
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_comm_max_f64(piper_hip_comm* c, double* value) {
  if (!c || !value) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  PH_CHECK_CTX(c->ctx);
  Rccl& R = rccl();
  PH_HIP(hipMemcpyAsync(c->scratch, value, sizeof(double), hipMemcpyHostToDevice, c->stream), PIPER_HIP_ERR_LAUNCH);
  PH_RCCL(R.AllReduce(c->scratch, c->scratch, 1, ncclFloat64, ncclMax, c->comm, c->stream));
  PH_HIP(hipMemcpyAsync(value, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipStreamSynchronize(c->stream), PIPER_HIP_ERR_LAUNCH);
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_comm_barrier(piper_hip_comm* c) {
  double z = 0.0;
  return piper_hip_comm_max_f64(c, &z);
}
