// mfmaprobe — what does v_mfma_f32_32x32x2_f32 sustain on gfx950 by number of independent accumulators per wave and waves per
// SIMD, and which clock does the chip hold meanwhile? (tools/probe; decides the register blocking of conv_pipe_kernel)
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfmaprobe.hip -o mfmaprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SMALL>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float seed, unsigned long long* stamps, float* sink) {
  f32x16 acc[NACC];
  f32x4 acc4[NACC];
  for (int a = 0; a < NACC; a++) {
    for (int q = 0; q < 16; q++) acc[a][q] = seed * (float)(a + q);
    for (int q = 0; q < 4; q++) acc4[a][q] = seed * (float)(a + q);
  }
  float av = seed + (float)threadIdx.x * 1e-3f, bv = seed - (float)threadIdx.x * 1e-3f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int a = 0; a < NACC; a++) {
        if constexpr (SMALL) acc4[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc4[a], 0, 0, 0);
        else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int a = 0; a < NACC; a++) {
    for (int q = 0; q < 16; q++) s += acc[a][q];
    for (int q = 0; q < 4; q++) s += acc4[a][q];
  }
  if (s == 1.2345e30f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int NACC, bool SMALL>
void run(int blocks_per_cu, int iters) {
  const int grid = 256 * blocks_per_cu;
  unsigned long long* st;
  float* sink;
  hipMalloc(&st, grid * 4 * 2 * 8);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL((mfma_loop<NACC, SMALL>), dim3(grid), dim3(256), 0, 0, iters, 0.5f, st, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((mfma_loop<NACC, SMALL>), dim3(grid), dim3(256), 0, 0, iters, 0.5f, st, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int w = 0; w < grid * 4; w++) { cyc += h[2 * w]; real += h[2 * w + 1]; }
  cyc /= grid * 4; real /= grid * 4;
  const double n_mfma = (double)iters * 8 * NACC;
  const double flop = n_mfma * (SMALL ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2) * grid * 4;
  printf("%s NACC=%d waves/SIMD=%d: %.1f cycles per MFMA per wave, %.1f per SIMD; clock %.2f GHz; %.1f TFLOP/s (kernel %.1f us)\n",
         SMALL ? "16x16x4" : "32x32x2", NACC, blocks_per_cu, cyc / n_mfma, cyc / n_mfma / blocks_per_cu, cyc / real * 0.1, flop / (ms * 1e-3) * 1e-12,
         ms * 1e3);
  hipFree(st); hipFree(sink);
}

int main() {
  const int it = 2000;
  run<1, false>(1, it); run<2, false>(1, it); run<4, false>(1, it);
  run<1, false>(2, it); run<2, false>(2, it); run<4, false>(2, it);
  run<1, false>(4, it);
  run<1, true>(1, it); run<2, true>(1, it); run<4, true>(1, it); run<1, true>(2, it); run<2, true>(2, it); run<1, true>(4, it);
  return 0;
}
