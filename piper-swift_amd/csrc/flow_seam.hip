// flow_seam.hip — the seam between two couplings of the reverse flow in ONE launch.
//
// Reference graph per ResidualCouplingLayer (reverse): … m = post(WN(pre(x0))) ; x1 ← x1 − m ; Flip ; next layer: h' = pre'(x0')
// where x0' is exactly the half the previous layer just updated (GraphExecutor dispatches Conv, Sub, Slice/Concat/Flip and the
// next Conv separately). Both convs are 1×1, i.e. column-local: a block that owns 16 columns computes
//   m      = W_post · skip + b_post                (half rows, K = H)
//   x1new  = x1 − m            → zp (in place, through the coupling's channel map) and → LDS
//   h'     = W_pre' · x1new + b_pre'               (H rows, K = half; logical input channel c = half − 1 − r: the Flip)
// with no global round trip in between — one launch (≈ 8 µs at these sizes, DESIGN §4 finding 9) less per coupling boundary.
// fp32 on v_mfma_f32_16x16x4_f32, weights from the 16-wide fragment images the streaming kernel uses.
#include "common.h"

namespace ph {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kSeamMaxH = 192;                       // H / 16 waves per block (≤ 12 ⇒ ≤ 170 VGPRs each)
constexpr int kSeamS1 = kSeamMaxH / 4, kSeamS2 = kSeamMaxH / 8;  // contraction steps: K = H (post), K = half ≤ H / 2 (pre')

// One wave per 16-row tile of the WIDER conv (pre': H rows); the first half/16 waves also own a tile of post. Every operand a
// wave will need — its column of skip, both weight-fragment streams, the old x1 — is requested at kernel start: the kernel is
// ONE memory round trip, two short MFMA chains and a barrier (the first version walked its row tiles one after the other and
// paid a cold weight round trip per tile: 22 µs against 2 × 7.8 for the launches it replaced).
__global__ __launch_bounds__(64 * (kSeamMaxH / 16)) void flow_seam_kernel(const float* __restrict__ skip, float* __restrict__ zp, float* __restrict__ h,
                                                                          const float* __restrict__ post16, const float* __restrict__ post_b,
                                                                          const float* __restrict__ pre16, const float* __restrict__ pre_b, int H,
                                                                          int half, int F, int post_steps, int pre_steps, int ob, int os,
                                                                          const int* __restrict__ len_ptr) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [half][16]: x1new by the NEXT coupling's logical input channel
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r16 = lane & 15, kq = lane >> 4;
  const int n = blockIdx.y, t0 = blockIdx.x * 16;
  const int Tv = len_ptr ? min(len_ptr[n], F) : F;
  if (t0 >= Tv) return;  // block-uniform
  const int tc = min(t0 + r16, F - 1);
  const bool valid = t0 + r16 < Tv;
  const float* sb = skip + (int64_t)n * H * F;
  float* zb = zp + (int64_t)n * 2 * half * F;
  float* hb = h + (int64_t)n * H * F;
  const int nsp = H >> 2, nsq = half >> 2;
  const bool has1 = wave < (half >> 4);  // wave-uniform: this wave owns post's row tile `wave`

  // ---- every load up front
  float b[kSeamS1], a1[kSeamS1], a2[kSeamS2], zold[4], bias1[4], bias2[4];
  const float* wa1 = post16 + (int64_t)(has1 ? wave : 0) * post_steps * 64 + lane;
  const float* wa2 = pre16 + (int64_t)wave * pre_steps * 64 + lane;
#pragma unroll
  for (int s = 0; s < kSeamS1; s++) {
    b[s] = sb[(int64_t)min(4 * s + kq, H - 1) * F + tc];
    a1[s] = wa1[min(s, post_steps - 1) * 64];
  }
  // phase 1's small operands BEFORE phase 2's weight stream (loads return in order: phase 1 must not wait for fragments it does not use)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row1 = min(16 * wave + 4 * kq + r, half - 1);
    bias1[r] = post_b[row1];
    zold[r] = zb[(int64_t)(ob + os * row1) * F + tc];
  }
#pragma unroll
  for (int s = 0; s < kSeamS2; s++) a2[s] = wa2[min(s, pre_steps - 1) * 64];
#pragma unroll
  for (int r = 0; r < 4; r++) bias2[r] = pre_b[16 * wave + 4 * kq + r];
  // ---- phase 1: m = post(skip), x1new = x1 − m → zp and LDS
  if (has1) {
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = bias1[r];
#pragma unroll
    for (int s = 0; s < kSeamS1; s++)
      if (s < nsp) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * wave + 4 * kq + r;
      const float v = zold[r] - acc[r];
      if (valid) zb[(int64_t)(ob + os * row) * F + t0 + r16] = v;
      xs[(half - 1 - row) * 16 + r16] = valid ? v : 0.0f;
    }
  }
  __syncthreads();
  // ---- phase 2: h' = pre'(x1new): B from LDS
  {
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = bias2[r];
#pragma unroll
    for (int s = 0; s < kSeamS2; s++)
      if (s < nsq) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], xs[min(4 * s + kq, half - 1) * 16 + r16], acc, 0, 0, 0);
    if (valid) {
#pragma unroll
      for (int r = 0; r < 4; r++) hb[(int64_t)(16 * wave + 4 * kq + r) * F + t0 + r16] = acc[r];
    }
  }
}

}  // namespace

bool flow_seam_eligible(int H, int half) { return H >= 16 && H <= kSeamMaxH && (H & 15) == 0 && half >= 16 && 2 * half <= H && (half & 15) == 0; }

// post16 / pre16: 16-wide fragment images (pack_conv_weights tm = 16) of post [half × H × 1] and of the NEXT coupling's pre [H × half × 1];
// ob / os: post's output channel map into zp (row r ↦ channel ob + os·r). skip [N][H][F], zp [N][2·half][F] (updated in place), h [N][H][F].
int launch_flow_seam(hipStream_t s, const float* skip, float* zp, float* h, const float* post16, const float* post_b, const float* pre16,
                     const float* pre_b, int N, int H, int half, int F, int post_steps, int pre_steps, int ob, int os, const int* len_ptr) {
  if (N <= 0 || F <= 0) return PIPER_HIP_OK;
  if (!flow_seam_eligible(H, half) || N > 65535 || !post_b || !pre_b) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "flow_seam: H=%d half=%d not covered", H, half);
  const dim3 grid((unsigned)ceil_div(F, 16), (unsigned)N);
  hipLaunchKernelGGL(flow_seam_kernel, grid, dim3(64 * (H / 16)), (size_t)half * 16 * sizeof(float), s, skip, zp, h, post16, post_b, pre16, pre_b, H,
                     half, F, post_steps, pre_steps, ob, os, len_ptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "flow_seam launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(flow_seam, flow_seam_kernel); } }
