// ops_rng.hip — MetalBackend.randomNormalLike (MetalBackend.swift:3398-3426) → random_normal_like_f32 (elementwise.metal:139-163).
#include "common.h"
#include "rng.h"

namespace {
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void random_normal_like_kernel(float* __restrict__ out, size_t n, unsigned seed_lo) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = ph::rnl_normal(seed_lo, (unsigned)i);
}

__global__ __launch_bounds__(kBlock) void random_draws_kernel(unsigned* __restrict__ out, size_t n, unsigned seed_lo) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    unsigned a, b;
    ph::rnl_draws(seed_lo, (unsigned)i, &a, &b);
    out[2 * i] = a;
    out[2 * i + 1] = b;
  }
}
}  // namespace

PH_EXPORT int piper_hip_random_normal_like_f32(piper_hip_ctx* ctx, size_t count, uint64_t seed, float** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (count > 0xffffffffull) PH_FAIL(PIPER_HIP_ERR_SHAPE, "random_normal_like: count exceeds the kernel's 32-bit element index (RNGParams.count is UInt32)");
  int rc = ph::ensure_out(ctx, out, count, 0);
  if (rc) return rc;
  if (count == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  const int grid = (int)std::min<int64_t>(ph::ceil_div((int64_t)count, kBlock), (int64_t)ctx->num_cus * 8);
  hipLaunchKernelGGL(random_normal_like_kernel, dim3(grid), dim3(kBlock), 0, ss.s, *out, count, (unsigned)(seed & 0xffffffffu));
  return ss.finish("random_normal_like_f32");
}

PH_EXPORT int piper_hip_random_draws_u32(piper_hip_ctx* ctx, size_t count, uint64_t seed, uint32_t** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (!out) PH_FAIL(PIPER_HIP_ERR_ARG, "null output pointer");
  if (count > 0x7fffffffull) PH_FAIL(PIPER_HIP_ERR_SHAPE, "random_draws: count too large");
  if (!*out) {
    void* p = nullptr;
    int rc = ctx->pool.alloc(2 * count * sizeof(uint32_t), &p);
    if (rc) return rc;
    *out = (uint32_t*)p;
  }
  if (count == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  const int grid = (int)std::min<int64_t>(ph::ceil_div((int64_t)count, kBlock), (int64_t)ctx->num_cus * 8);
  hipLaunchKernelGGL(random_draws_kernel, dim3(grid), dim3(kBlock), 0, ss.s, *out, count, (unsigned)(seed & 0xffffffffu));
  return ss.finish("random_draws_u32");
}
namespace { PH_WARM(ops_rng, random_normal_like_kernel); }
