// ldprobe — how fast can ONE CU pull a private, cold stream through its vector memory path at the occupancy the short-utterance
// convs run at (8 waves per CU, every wave its own 60 KB of weights), and does the width of the load instruction matter?
// Every wave reads `bytes_per_wave` from its own region, in groups of 32 dwords per lane in flight (like the operand ring).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/ldprobe.hip -o tools/probe/bin/ldprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int W>  // dwords per lane per load instruction: 1, 2, 4
__global__ __launch_bounds__(512) void stream_kernel(const float* __restrict__ buf, float* __restrict__ out, int dwords_per_wave, size_t wave_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wid = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
  const float* base = buf + wid * wave_stride;
  float acc = 0.0f;
  constexpr int G = 32 / W;  // load instructions per group: 32 dwords per lane in flight
  for (int o = 0; o < dwords_per_wave; o += 64 * 32) {
    if constexpr (W == 1) {
      float v[G];
#pragma unroll
      for (int i = 0; i < G; i++) v[i] = base[o + i * 64 + lane];
#pragma unroll
      for (int i = 0; i < G; i++) acc += v[i];
    } else if constexpr (W == 2) {
      float2 v[G];
#pragma unroll
      for (int i = 0; i < G; i++) v[i] = ((const float2*)(base + o))[i * 64 + lane];
#pragma unroll
      for (int i = 0; i < G; i++) acc += v[i].x + v[i].y;
    } else {
      float4 v[G];
#pragma unroll
      for (int i = 0; i < G; i++) v[i] = ((const float4*)(base + o))[i * 64 + lane];
#pragma unroll
      for (int i = 0; i < G; i++) acc += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t total = (size_t)3 << 30;  // 3 GiB of floats region to rotate through (cold every time)
  float *buf, *out;
  CK(hipMalloc(&buf, total));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("# %d CUs; per-CU rate = bytes one block pulled / kernel time (includes ~2 us of launch)\n", cus);
  printf("%-8s %-8s %-10s %-12s %-10s %-12s %s\n", "blocks", "waves", "KB/wave", "width", "us", "B/clk/CU", "GB/s total");
  size_t cursor = 0;
  for (int blocks : {cus / 4, cus}) {
    for (int waves : {4, 8, 16}) {
      if (waves * 64 > 1024) continue;
      for (int kb : {16, 64, 256}) {
        for (int w : {1, 2, 4}) {
          const int dwords = kb * 256;
          const size_t need = (size_t)blocks * waves * dwords;
          float best = 1e30f;
          for (int rep = 0; rep < 3; rep++) {
            if ((cursor + need) * 4 > total) cursor = 0;
            const float* src = buf + cursor;
            cursor += need;
            CK(hipEventRecord(e0, 0));
            if (w == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(waves * 64), 0, 0, src, out, dwords, (size_t)dwords);
            if (w == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(waves * 64), 0, 0, src, out, dwords, (size_t)dwords);
            if (w == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(blocks), dim3(waves * 64), 0, 0, src, out, dwords, (size_t)dwords);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const double us = best * 1000.0;
          const double bytes_cu = (double)waves * dwords * 4;
          printf("%-8d %-8d %-10d %-12s %-10.2f %-12.1f %.0f\n", blocks, waves, kb, w == 1 ? "dword" : w == 2 ? "dwordx2" : "dwordx4", us,
                 bytes_cu / (us * 1e-6 * 2.1e9), (double)need * 4 / (us * 1e-6) / 1e9);
        }
      }
    }
  }
  return 0;
}
