// tailprobe — what does a deterministic cross-block split-K reduction ("last arriver reduces") cost per launch, against the
// bytes per CU it saves?  Decides whether the short-row convs of the encoder / flow (DESIGN.md finding 10: bound by the bytes each
// CU pulls in) should split their contraction over BLOCKS instead of over the waves of one block.
//
// Kernel: grid = tiles × S blocks. Every block pulls `kb` KiB of operands (wide contiguous loads, shared between the blocks of
// equal tile index modulo 24 like weights are), writes a P-float partial per thread, then
//   mode 0: nothing else (the baseline: as if the block had owned the whole contraction — run it with S× the bytes to compare)
//   mode 1: release fence, atomic arrive on the tile's counter; the last arriver acquires, sums the S partials in FIXED order,
//           writes the result and resets the counter (deterministic, graph-replayable).
// 100 dependent launches in a captured graph, time per launch.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/tailprobe.hip -o tools/probe/bin/tailprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE, int P>
__global__ __launch_bounds__(512) void tail_kernel(const float4* __restrict__ w, const float* __restrict__ xin, float* __restrict__ partial,
                                                    float* __restrict__ out, unsigned* counters, int S, int f4_per_thread, int stamp, unsigned* flags) {
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int tile = blockIdx.x / S, ks = blockIdx.x % S;
  // operands: a slice shared with other tiles (weights) — 16 B per lane per load, all requested before the first use
  const float4* src = w + ((size_t)(tile % 24) * S + ks) * (size_t)f4_per_thread * nt;
  float acc = xin[(tile * 64 + (tid & 63)) % 4096];  // depends on the previous launch's output
  float4 v[16];
  for (int o = 0; o < f4_per_thread; o += 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = (o + i < f4_per_thread) ? src[(size_t)(o + i) * nt + tid] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) acc += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  float r[P];
#pragma unroll
  for (int p = 0; p < P; p++) r[p] = acc * 0.0f + (float)(stamp + p);
  if (MODE == 0) {
#pragma unroll
    for (int p = 0; p < P; p++) out[((size_t)tile * P + p) * nt + tid] = r[p];
    return;
  }
  float* mine = partial + ((size_t)tile * S + ks) * P * nt;
  if (MODE == 1) {
#pragma unroll
    for (int p = 0; p < P; p++) mine[(size_t)p * nt + tid] = r[p];
  } else {  // MODE 2: every partial element is an agent-scope relaxed atomic store (sc1: written through to the coherence point), no cache-wide fence
#pragma unroll
    for (int p = 0; p < P; p++) __hip_atomic_store(mine + (size_t)p * nt + tid, r[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);  // this thread's stores are acknowledged
  }
  __syncthreads();  // every thread's partial stores are issued (mode 2: complete) …
  if (tid == 0) {
    if (MODE == 1) __atomic_thread_fence(__ATOMIC_RELEASE);  // … and made visible device-wide (agent scope) before the arrive
    const unsigned old = __hip_atomic_fetch_add(&counters[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == (unsigned)S - 1u) ? 1u : 0u;
    if (old == (unsigned)S - 1u) {
      __hip_atomic_store(&counters[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
  }
  __syncthreads();
  if (!s_last) return;
  float sum[P];
#pragma unroll
  for (int p = 0; p < P; p++) sum[p] = 0.0f;
  for (int s2 = 0; s2 < S; s2++) {  // fixed order: slice 0 + slice 1 + …
    const float* theirs = partial + ((size_t)tile * S + s2) * P * nt;
#pragma unroll
    for (int p = 0; p < P; p++) {
      float v;
      if (MODE == 1) v = __builtin_nontemporal_load(theirs + (size_t)p * nt + tid);
      else v = __hip_atomic_load(theirs + (size_t)p * nt + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (float)(stamp + p)) flags[0] = 1;  // a stale partial crossed the arrive
      sum[p] += v;
    }
  }
#pragma unroll
  for (int p = 0; p < P; p++) out[((size_t)tile * P + p) * nt + tid] = sum[p];
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  float4* w;
  float *xa, *xb, *partial;
  unsigned* counters;
  unsigned* flags;
  const size_t wbytes = (size_t)256 << 20;
  CK(hipMalloc(&w, wbytes));
  CK(hipMemset(w, 0, wbytes));
  CK(hipMalloc(&xa, 8 << 20));
  CK(hipMalloc(&xb, 8 << 20));
  CK(hipMemset(xa, 0, 8 << 20));
  CK(hipMemset(xb, 0, 8 << 20));
  CK(hipMalloc(&partial, 64 << 20));
  CK(hipMalloc(&counters, 4096 * 4));
  CK(hipMemset(counters, 0, 4096 * 4));
  CK(hipMalloc(&flags, 16));
  CK(hipMemset(flags, 0, 16));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 100;
  printf("# %s: %d CUs; graph of %d dependent launches, us per launch\n", prop.name, prop.multiProcessorCount, iters);
  printf("%-6s %-4s %-8s %-8s %-6s %-10s\n", "tiles", "S", "threads", "KB/blk", "mode", "us/launch");
  auto run = [&](int tiles, int S, int threads, int kb, int mode) {
    const int f4_per_thread = kb * 1024 / 16 / threads;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < iters; i++) {
      const float* in = (i & 1) ? xb : xa;
      float* out = (i & 1) ? xa : xb;
      if (mode == 0) hipLaunchKernelGGL((tail_kernel<0, 2>), dim3(tiles * S), dim3(threads), 0, s, w, in, partial, out, counters, S, f4_per_thread, i, flags);
      else if (mode == 1) hipLaunchKernelGGL((tail_kernel<1, 2>), dim3(tiles * S), dim3(threads), 0, s, w, in, partial, out, counters, S, f4_per_thread, i, flags);
      else hipLaunchKernelGGL((tail_kernel<2, 2>), dim3(tiles * S), dim3(threads), 0, s, w, in, partial, out, counters, S, f4_per_thread, i, flags);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-6d %-4d %-8d %-8d %-6d %-10.3f\n", tiles, S, threads, kb, mode, best * 1000.0f / iters);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  };
  // the whole contraction in one block (mode 0) with S× the bytes, against S blocks with a tail reduction
  for (int kb : {16, 32, 48}) {
    run(84, 1, 512, 3 * kb, 0);
    run(252, 1, 512, 3 * kb, 0);
    run(252, 1, 512, kb, 0);
    run(84, 3, 512, kb, 1);
    run(84, 3, 512, kb, 2);
    run(63, 4, 512, kb, 2);
    run(126, 2, 512, kb, 2);
    run(36, 7, 512, kb, 2);
    run(168, 3, 256, kb, 2);
  }
  run(252, 1, 512, 1, 0);
  run(84, 3, 512, 1, 2);
  unsigned hc[8];
  CK(hipMemcpy(hc, counters, 32, hipMemcpyDeviceToHost));
  unsigned hf[4];
  CK(hipMemcpy(hf, flags, 16, hipMemcpyDeviceToHost));
  printf("counters after: %u %u %u %u (must be 0); stale-partial flag: %u (must be 0)\n", hc[0], hc[1], hc[2], hc[3], hf[0]);
  return 0;
}
