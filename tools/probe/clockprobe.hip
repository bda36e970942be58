// clockprobe.hip — diagnostic: effective shader clock, MFMA rate, kernel-boundary cost, dependent-load latency.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/clockprobe.hip -o tools/probe/clockprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void mfma_loop(float* out, int iters, unsigned long long* clk) {
  f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void chase(const int* next, int* out, int steps) {
  int i = 0;
  for (int s = 0; s < steps; s++) i = next[i];
  out[0] = i;
}
int main() {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float))); CK(hipMalloc(&clk, 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 3; rep++) {
    int iters = rep == 0 ? 2000 : 200000;
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256), 0, st, out, iters, clk);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double flops = 256.0 * 4 * iters * 4 * 4096.0;
    printf("mfma_loop iters=%d: %.3f ms  %.1f TFLOP/s  shader clock ≈ %.0f MHz (memtime/memrealtime)\n", iters, ms, flops / ms / 1e9,
           (double)h[0] / (double)h[1] * 100.0);
  }
  // kernel boundary cost
  for (int rep = 0; rep < 2; rep++) {
    const int n = 2000;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, nullptr);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel x%d back-to-back: %.2f us each\n", n, ms * 1000 / n);
  }
  // graph of empty kernels
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, nullptr);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("graph of 100 empty kernels: %.2f us per kernel\n", ms * 1000 / 100);
    }
  }
  // dependent load latency: stride through 64 MiB (beyond L2) and 1 MiB (L2) rings
  for (size_t bytes : {(size_t)256 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
    size_t n = bytes / 4; std::vector<int> h(n);
    size_t stride = 4099 * 16;  // pseudo-random walk with a large odd stride (in ints)
    for (size_t i = 0; i < n; i++) h[i] = (int)((i + stride) % n);
    int* d; int* o; CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    const int steps = 20000;
    hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, st, d, o, 1000);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, st, d, o, steps);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dependent load chain over %zu KiB: %.1f ns per load\n", bytes >> 10, ms * 1e6 / steps);
    hipFree(d); hipFree(o);
  }
  return 0;
}
