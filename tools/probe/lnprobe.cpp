// lnprobe.cpp — time library kernels inside a 100-node graph (per-kernel period), through the C-ABI.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/piper_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  piper_hip_ctx* ctx; if (piper_hip_create(0, &ctx)) { printf("%s\n", piper_hip_last_error()); return 1; }
  piper_hip_stream ps; piper_hip_stream_create(ctx, &ps); hipStream_t st = (hipStream_t)ps;
  const int C = 192, T = 112;
  std::vector<float> h(C * T, 1.0f), g(C, 1.0f);
  float *x, *y, *ga, *be, *o1 = nullptr, *o2 = nullptr;
  piper_hip_upload_f32(ctx, h.data(), h.size(), &x); piper_hip_upload_f32(ctx, h.data(), h.size(), &y);
  piper_hip_upload_f32(ctx, g.data(), C, &ga); piper_hip_upload_f32(ctx, g.data(), C, &be);
  piper_hip_alloc(ctx, C * T * 4, (void**)&o1); piper_hip_alloc(ctx, C * T * 4, (void**)&o2);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_graph = [&](const char* name, auto body) -> int {
    body(0);  CK(hipStreamSynchronize(st));
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; i++) body(i);
    CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-40s %.2f us per kernel\n", name, best * 10);
    return 0;
  };
  time_graph("add_layernorm 192x112", [&](int i) { float* o = (i & 1) ? o1 : o2; piper_hip_add_layernorm_f32(ctx, (i & 1) ? o2 : x, y, ga, be, 1, C, T, 1e-5f, &o, ps); });
  time_graph("unary tanh 21504", [&](int i) { float* o = (i & 1) ? o1 : o2; piper_hip_unary_f32(ctx, PIPER_HIP_TANH, (i & 1) ? o2 : x, C * T, 0.f, &o, ps); });
  int64_t shp[3] = {1, C, T};
  time_graph("softmax 192 rows x112", [&](int i) { float* o = (i & 1) ? o1 : o2; piper_hip_softmax_lastdim_f32(ctx, (i & 1) ? o2 : x, shp, 3, &o, ps); });
  return 0;
}
