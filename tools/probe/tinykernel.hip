// tinykernel.hip — how long does a tiny *real* kernel take inside a graph (per-kernel floor), by access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_copy(const float* a, float* b, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) b[i] = a[i] * 2.0f; }
__global__ void k_dep(const float* a, const int* idx, float* b, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) b[i] = a[idx[i]]; }
__global__ void k_chain(const float* a, float* b, int n, int depth) {  // depth dependent loads per thread
  int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return; int j = i; float v = 0;
  for (int d = 0; d < depth; d++) { v = a[j]; j = (j + (int)(v * 0.0f) + 977) % n; } b[i] = v; }
int main() {
  const int n = 21504, nb = (n + 255) / 256;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int NB = 8; float* buf[NB]; int* idx;
  std::vector<int> hi(n); for (int i = 0; i < n; i++) hi[i] = (i * 7919) % n;
  for (int i = 0; i < NB; i++) { CK(hipMalloc(&buf[i], (size_t)(1 << 20) * (i + 1))); CK(hipMemset(buf[i], 0, n * 4)); }
  CK(hipMalloc(&idx, n * 4)); CK(hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice));
  auto run = [&](const char* name, auto launch) -> int {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; i++) launch(i);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s %.2f us per kernel\n", name, best * 10);
    return 0;
  };
  run("copy 21504 floats (84 blocks), ping-pong", [&](int i) { hipLaunchKernelGGL(k_copy, dim3(nb), dim3(256), 0, st, buf[i % NB], buf[(i + 1) % NB], n); });
  run("copy 1 block", [&](int i) { hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, buf[i % NB], buf[(i + 1) % NB], 256); });
  run("2 dependent loads (gather)", [&](int i) { hipLaunchKernelGGL(k_dep, dim3(nb), dim3(256), 0, st, buf[i % NB], idx, buf[(i + 1) % NB], n); });
  for (int depth : {1, 4, 16, 64})
    run((std::string("chain depth ") + std::to_string(depth)).c_str(), [&](int i) { hipLaunchKernelGGL(k_chain, dim3(nb), dim3(256), 0, st, buf[i % NB], buf[(i + 1) % NB], n, depth); });
  run("copy 2.7M floats (10752 blocks)", [&](int i) { hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, st, buf[i % NB], buf[(i + 1) % NB], 1 << 18); });
  return 0;
}
