// gridbarrier — what does a device-wide barrier inside ONE persistent kernel cost on this part, against the ≈ 3.85 µs boundary
// between two dependent kernels of a HIP graph?  Decides whether a persistent encoder/flow kernel (stages separated by grid
// barriers instead of launches) can beat the launch chain at all.
//
// Every block: write its slice of a buffer (so the L2 of its XCD holds dirty lines the next stage's readers on OTHER XCDs need),
// release fence, one atomic arrive, spin on the counter, acquire fence, read a slice another block wrote and check it.
// The spin is bounded: a block that does not see the others within `kMaxSpin` polls gives up and flags it (every wave exits).
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe/gridbarrier.hip -o tools/probe/bin/gridbarrier && tools/probe/bin/gridbarrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int kMaxSpin = 1 << 22;

template <bool PAYLOAD>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, unsigned* flags, float* buf, int per_block, int iters) {
  const unsigned nb = gridDim.x;
  const int tid = threadIdx.x;
  float sum = 0.0f;
  for (int it = 0; it < iters; it++) {
    if (PAYLOAD) {
      float* mine = buf + (size_t)blockIdx.x * per_block;
      for (int i = tid; i < per_block; i += 256) mine[i] = (float)(it + 1);
    }
    __syncthreads();
    if (tid == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);  // device scope: write back what other XCDs will read
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(it + 1) * nb;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kMaxSpin) {
          flags[0] = 1;  // gave up: the grid was not co-resident or something hung
          break;
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    if (PAYLOAD) {
      // read the slice of a block on another XCD (blocks are dealt round-robin to the 8 XCDs)
      const float* theirs = buf + (size_t)((blockIdx.x + 1) % nb) * per_block;
      for (int i = tid; i < per_block; i += 256) {
        const float v = __builtin_nontemporal_load(theirs + i);
        if (v < (float)(it + 1)) flags[1] = 1;  // stale data crossed the barrier (a neighbour may already be one stage ahead: newer is fine)
        sum += v;
      }
    }
  }
  if (sum == -1.0f) flags[2] = 1;
}


// A stage of a dependent chain: every block reads `kb` KiB that the PREVIOUS launch wrote (a neighbour block's slice), adds one, writes
// its own slice for the next launch. What a kernel of this path cannot avoid paying: boundary + first touch of the producer's output.
__global__ __launch_bounds__(256) void chain_kernel(const float* __restrict__ in, float* __restrict__ out, int per_block) {
  const float* src = in + (size_t)((blockIdx.x + 1) % gridDim.x) * per_block;
  float* dst = out + (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += 256) dst[i] = src[i] + 1.0f;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  unsigned *counter, *flags;
  float* buf;
  const int max_blocks = cus * 2, max_per_block = 16384;
  CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&flags, 16));
  CK(hipMalloc(&buf, (size_t)max_blocks * max_per_block * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("# %s, %d CUs, %d barriers per launch\n", prop.name, cus, iters);
  printf("%-10s %-12s %-14s %s\n", "blocks", "payload B", "us/barrier", "flags(gave_up,stale)");
  for (int blocks : {cus / 4, cus, cus * 2}) {
    for (int per_block : {0, 256, 4096, 16384}) {
      float best = 1e30f;
      unsigned hf[4] = {0, 0, 0, 0};
      for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(counter, 0, 4));
        CK(hipMemset(flags, 0, 16));
        CK(hipEventRecord(e0, 0));
        void* args[] = {&counter, &flags, &buf, (void*)&per_block, (void*)&iters};
        const void* fn = per_block ? (const void*)barrier_kernel<true> : (const void*)barrier_kernel<false>;
        CK(hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned f[4];
        CK(hipMemcpy(f, flags, 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; i++) hf[i] |= f[i];
      }
      printf("%-10d %-12d %-14.3f %u,%u\n", blocks, per_block * 4, best * 1000.0f / iters, hf[0], hf[1]);
      if (hf[0]) { fprintf(stderr, "a block gave up waiting: stopping\n"); return 1; }
    }
  }
  // reference: the same number of dependent empty launches in a captured graph
  {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t g;
    hipGraphExec_t ge;
    int zero = 0, one = 1;
    CK(hipMemset(counter, 0, 4));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(barrier_kernel<false>, dim3(cus), dim3(256), 0, s, counter, flags, buf, zero, zero);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)one;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("graph of %d dependent empty launches: %.3f us per launch\n", iters, best * 1000.0f / iters);
  }
  // dependent chains in a graph: ping-pong between two buffers
  {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float* buf2;
    CK(hipMalloc(&buf2, (size_t)max_blocks * max_per_block * 4));
    CK(hipMemset(buf, 0, (size_t)max_blocks * max_per_block * 4));
    CK(hipMemset(buf2, 0, (size_t)max_blocks * max_per_block * 4));
    printf("%-10s %-14s %s\n", "blocks", "KiB per block", "us per launch (graph of dependent launches, each reads what the previous wrote)");
    for (int blocks : {16, 64, cus, 2 * cus}) {
      for (int per_block : {64, 1024, 4096, 16384}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < iters; i++) {
          float* a = (i & 1) ? buf2 : buf;
          float* b = (i & 1) ? buf : buf2;
          hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), 0, s, a, b, per_block);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          CK(hipEventRecord(e0, s));
          CK(hipGraphLaunch(ge, s));
          CK(hipEventRecord(e1, s));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("%-10d %-14.2f %.3f\n", blocks, per_block * 4 / 1024.0, best * 1000.0f / iters);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
      }
    }
  }
  return 0;
}
