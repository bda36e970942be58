// floorprobe — what is the ≈ 5.2 µs of a tiny short-row conv launch (k = 1, 24 KB of operands per CU, 6 MFMAs per wave) made of?
// The same skeleton with its steps switched on one by one, 100 dependent launches in a captured graph each:
//   V0  operand loads (24 KB per block, contiguous) + one store per thread
//   V1  V0 + a 320-byte by-value argument struct whose fields are read where they are used (lazy scalar loads)
//   V2  V1 + the fields pinned in one burst at kernel start
//   V3  V0 + split-K exchange: LDS write, __syncthreads, LDS reads by the first waves
//   V4  V3 + an epilogue load AFTER the barrier (the residual / skip addend: a cold dependent round trip) before the store
//   V5  V4 with that load issued at kernel start instead
//   V6  V4 + the MFMAs and a dependent "true length" load in front of the operands (what conv_stream_kernel does)
//   hipcc -O3 --offload-arch=gfx950 tools/probe/floorprobe.hip -o tools/probe/bin/floorprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct BigArgs {  // the shape of ConvArgs: pointers and ints, 320 bytes
  const float* p[12];
  int i[40];
  long long s[8];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512) void floor_kernel(const float* __restrict__ w, const float* __restrict__ xin, float* __restrict__ out, const int* __restrict__ lens,
                                                     const BigArgs a) {
  __shared__ float red[8 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* res = xin + 8192;
  int fields = 0;
  if (V == 2) {
    int f0 = a.i[0], f1 = a.i[7], f2 = a.i[15], f3 = a.i[23], f4 = a.i[31], f5 = a.i[39];
    long long s0 = a.s[0], s7 = a.s[7];
    asm volatile("" ::"s"(f0), "s"(f1), "s"(f2), "s"(f3), "s"(f4), "s"(f5), "s"(s0), "s"(s7));
    fields = f0 + f1 + f2 + f3 + f4 + f5 + (int)s0 + (int)s7;
  }
  float pre = 0.0f;
  if (V == 5) pre = res[(blockIdx.x * 64 + lane) % 4096];
  int Lv = 1 << 30;
  if (V == 6) Lv = lens[blockIdx.x & 7];  // a dependent load in front of the operands
  // operands: 12 dword loads per thread (24 KB per block), shared by groups of blocks like weights are
  const float* src = w + (size_t)(blockIdx.x % 24) * 6144 + (V == 6 ? (Lv & 1) : 0);
  float v[12];
#pragma unroll
  for (int k = 0; k < 12; k++) v[k] = src[k * 512 + tid];
  float acc0 = xin[(blockIdx.x * 64 + lane) % 4096];  // depends on the previous launch
  if (V == 1) fields = a.i[0];
  f32x4 acc = {acc0, 0.0f, 0.0f, 0.0f};
  if (V == 6) {
#pragma unroll
    for (int k = 0; k < 6; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2 * k], v[2 * k + 1], acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k & 3] += v[k];
  }
  if (V == 1) { fields += a.i[7] + a.i[15]; acc[1] += (float)fields; fields = a.i[23] + a.i[31] + a.i[39] + (int)a.s[0] + (int)a.s[7]; }
  if (V < 3) {
    out[blockIdx.x * 512 + tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]) + (float)fields * 1e-30f;
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave >= 4) return;
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; q++) s += red[(q * 4 + wave) * 64 + lane];
  if (V == 4 || V == 6) s += res[(blockIdx.x * 64 + lane) % 4096];  // cold, behind the barrier
  if (V == 5) s += pre;
  out[(blockIdx.x * 4 + wave) * 64 + lane] = s + (float)fields * 1e-30f;
}

int main() {
  CK(hipSetDevice(0));
  float *w, *xa, *xb;
  int* lens;
  CK(hipMalloc(&w, 64 << 20));
  CK(hipMemset(w, 0, 64 << 20));
  CK(hipMalloc(&xa, 8 << 20));
  CK(hipMalloc(&xb, 8 << 20));
  CK(hipMemset(xa, 0, 8 << 20));
  CK(hipMemset(xb, 0, 8 << 20));
  CK(hipMalloc(&lens, 64));
  CK(hipMemset(lens, 0, 64));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  BigArgs a = {};
  const int iters = 100, blocks = 252;
  printf("# graph of %d dependent launches, %d blocks x 512 threads, us per launch\n", iters, blocks);
  auto run = [&](int variant, const char* what) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < iters; i++) {
      const float* in = (i & 1) ? xb : xa;
      float* out = (i & 1) ? xa : xb;
      switch (variant) {
        case 0: hipLaunchKernelGGL(floor_kernel<0>, dim3(blocks), dim3(512), 0, s, w, in, out, lens, a); break;
        case 1: hipLaunchKernelGGL(floor_kernel<1>, dim3(blocks), dim3(512), 0, s, w, in, out, lens, a); break;
        case 2: hipLaunchKernelGGL(floor_kernel<2>, dim3(blocks), dim3(512), 0, s, w, in, out, lens, a); break;
        case 3: hipLaunchKernelGGL(floor_kernel<3>, dim3(blocks), dim3(512), 0, s, w, in, out, lens, a); break;
        case 4: hipLaunchKernelGGL(floor_kernel<4>, dim3(blocks), dim3(512), 0, s, w, in, out, lens, a); break;
        case 5: hipLaunchKernelGGL(floor_kernel<5>, dim3(blocks), dim3(512), 0, s, w, in, out, lens, a); break;
        default: hipLaunchKernelGGL(floor_kernel<6>, dim3(blocks), dim3(512), 0, s, w, in, out, lens, a); break;
      }
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("V%d %-92s %.3f\n", variant, what, best * 1000.0f / iters);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  };
  run(0, "operand loads (24 KB per block) + store");
  run(1, "+ 320-byte by-value argument struct, fields read lazily");
  run(2, "+ the same fields pinned in one burst at kernel start");
  run(3, "V0 + split-K exchange (LDS write, barrier, LDS reads)");
  run(4, "V3 + epilogue addend loaded behind the barrier");
  run(5, "V3 + epilogue addend requested at kernel start");
  run(6, "V4 + MFMAs + a dependent length load in front of the operands");
  return 0;
}
