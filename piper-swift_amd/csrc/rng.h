// rng.h — the reference's RandomNormalLike stream (elementwise.metal:132-163), shared by the op entry point and the kernels
// that generate their noise in place (path expansion of the voice, duration-predictor input).
#pragma once
#include <hip/hip_runtime.h>

namespace ph {

__host__ __device__ __forceinline__ unsigned rnl_xorshift32(unsigned x) {  // elementwise.metal:132-137
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return x;
}

// the two raw 32-bit draws of element `gid` (elementwise.metal:148-152)
__host__ __device__ __forceinline__ void rnl_draws(unsigned seed_lo, unsigned gid, unsigned* u0, unsigned* u1) {
  unsigned state = seed_lo ^ (gid * 747796405u + 2891336453u);
  state = rnl_xorshift32(state);
  *u0 = state;
  state = rnl_xorshift32(state);
  *u1 = state;
}

// Box-Muller on the two draws mapped to (0, 1] (elementwise.metal:154-162). The uniforms are exact in fp32 on both sides;
// sqrt / log / cos are the device's (Metal's fast-math forms in the reference): equal to ~1e-6, not bit for bit.
__device__ __forceinline__ float rnl_normal(unsigned seed_lo, unsigned gid) {
  unsigned a, b;
  rnl_draws(seed_lo, gid, &a, &b);
  const float u0 = ((float)a + 1.0f) / 4294967296.0f;
  const float u1 = ((float)b + 1.0f) / 4294967296.0f;
  const float r = sqrtf(-2.0f * logf(u0));
  const float theta = 6.28318530718f * u1;
  return r * cosf(theta);
}

}  // namespace ph
