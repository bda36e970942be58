// sharedprobe — how fast do the CUs pull a stream that OTHER CUs pull too (weights of a row tile shared by all column-chunk
// blocks; activations shared by all row-tile blocks)?  ldprobe measured the private cold stream (HBM-bound, 9–11 B/clk/CU).
// Here every block reads one of `nbuf` buffers of `kb` KiB (block b reads buffer b % nbuf, so nbuf = 8 ⇒ one buffer per XCD,
// nbuf = 1 ⇒ the whole chip reads the same bytes, nbuf = blocks ⇒ private), its waves reading disjoint slices, optionally
// starting at a per-block rotation (does lock-step access to the same lines hot-spot an L2 channel?).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/sharedprobe.hip -o tools/probe/bin/sharedprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// pattern 0: 256 contiguous bytes per wave load (an A fragment); pattern 1: 4 rows × 64 B, rows `row_stride` floats apart (a B fragment)
template <int PAT>
__global__ __launch_bounds__(1024) void shared_kernel(const float* __restrict__ buf, float* __restrict__ out, int dwords_per_buf, int nbuf, int rotate,
                                                       int row_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float* base = buf + (size_t)(blockIdx.x % nbuf) * dwords_per_buf;
  const int per_wave = dwords_per_buf / nw;  // multiple of 64·16
  const int loads = per_wave / 64;
  const int rot = rotate ? (int)((blockIdx.x / nbuf) * 37u % (unsigned)loads) : 0;
  float acc = 0.0f;
  for (int o = 0; o < loads; o += 16) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      int l = o + i + rot;
      if (l >= loads) l -= loads;
      int idx;
      if (PAT == 0) idx = wave * per_wave + l * 64 + lane;
      else idx = (wave * per_wave + l * 64 + (lane >> 4) * row_stride + (lane & 15)) % dwords_per_buf;
      v[i] = base[idx];
    }
#pragma unroll
    for (int i = 0; i < 16; i++) acc += v[i];
  }
  if (acc == 123.456f) out[0] = acc;
}

// LDS broadcast variant: the block stages its buffer slice through LDS once with float4 loads, then all waves read it from LDS
// `reuse` times (models weight or activation reuse inside a block) — the global side is touched once per block.
__global__ __launch_bounds__(1024) void staged_kernel(const float* __restrict__ buf, float* __restrict__ out, int dwords_per_buf, int nbuf, int reuse) {
  extern __shared__ float lds[];
  const float4* base = (const float4*)(buf + (size_t)(blockIdx.x % nbuf) * dwords_per_buf);
  float acc = 0.0f;
  const int chunk = 16 * 1024;  // dwords per stage (64 KB)
  for (int o = 0; o < dwords_per_buf; o += chunk) {
    const int n4 = min(chunk, dwords_per_buf - o) / 4;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) ((float4*)lds)[i] = base[o / 4 + i];
    __syncthreads();
    for (int r = 0; r < reuse; r++)
      for (int i = threadIdx.x; i < n4 * 4; i += blockDim.x) acc += lds[(i + r * 64) % (n4 * 4)];
    __syncthreads();
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t total = (size_t)1 << 30;
  float *buf, *out;
  CK(hipMalloc(&buf, total));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)staged_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  printf("# %d CUs. us = kernel time between events (includes ~2-3 us eager launch); B/clk/CU = bytes one block pulled / time at 2.1 GHz\n", cus);
  printf("%-7s %-6s %-6s %-7s %-5s %-4s %-9s %-10s %s\n", "blocks", "waves", "KB", "nbuf", "pat", "rot", "us", "B/clk/CU", "L1-side GB/s total");
  size_t cursor = 0;
  auto run = [&](int blocks, int waves, int kb, int nbuf, int pat, int rot, int staged) {
    const int dwords = kb * 256;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      const size_t need = (size_t)nbuf * dwords;
      if ((cursor + need) * 4 > total) cursor = 0;
      const float* src = buf + cursor;
      cursor += need;  // fresh addresses every repetition: cold in every cache
      CK(hipEventRecord(e0, 0));
      if (staged) hipLaunchKernelGGL(staged_kernel, dim3(blocks), dim3(waves * 64), 64 * 1024, 0, src, out, dwords, nbuf, staged);
      else if (pat == 0) hipLaunchKernelGGL(shared_kernel<0>, dim3(blocks), dim3(waves * 64), 0, 0, src, out, dwords, nbuf, rot, 336);
      else hipLaunchKernelGGL(shared_kernel<1>, dim3(blocks), dim3(waves * 64), 0, 0, src, out, dwords, nbuf, rot, 336);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double us = best * 1000.0, bytes_cu = (double)dwords * 4;
    printf("%-7d %-6d %-6d %-7d %-5s %-4d %-9.2f %-10.1f %.0f\n", blocks, waves, kb, nbuf, staged ? "lds" : pat ? "B" : "A", rot, us,
           bytes_cu / (us * 1e-6 * 2.1e9), bytes_cu * blocks / (us * 1e-6) / 1e9);
  };
  for (int kb : {64, 128, 256})
    for (int waves : {8, 16}) {
      for (int nbuf : {1, 8, 24, cus}) {
        run(cus, waves, kb, nbuf, 0, 0, 0);
        if (nbuf != cus) run(cus, waves, kb, nbuf, 0, 1, 0);
      }
      run(cus, waves, kb, 8, 1, 0, 0);
      run(cus, waves, kb, 8, 0, 0, 1);
      run(cus, waves, kb, 8, 0, 0, 4);
    }
  // two blocks per CU
  for (int kb : {64, 128}) {
    run(2 * cus, 8, kb, 8, 0, 0, 0);
    run(2 * cus, 8, kb, 24, 0, 0, 0);
  }
  return 0;
}
