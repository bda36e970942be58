// dp.hip — kernels of the stochastic duration predictor (VITS `dp`, inference direction).
//
// In the reference this is the part of the graph made of the shape / mask / spline arms of GraphExecutor.swift:2379-2645
// (Softplus, CumSum, GatherElements, NonZero, GatherND, ScatterND, Where, …) plus depthwise Conv, the LayerNorm chain
// (:2071-2125) and GELU as Div / Erf / Add / Mul — a few hundred nodes on tensors of T columns. Here:
//   * dds_layer_kernel: one layer of a dilated depth-separable conv stack (modules.DDSConv) in ONE launch — depthwise conv
//     (k taps, dilation k^i) → LayerNorm → GELU → 1×1 conv (v_mfma_f32_16x16x4_f32 on the packed 16-wide image) → LayerNorm →
//     GELU → + x. A block owns 16 columns and every channel, so both LayerNorms are exact two-pass reductions inside the block.
//   * dp_spline_kernel: the inverse rational-quadratic spline of a ConvFlow, one thread per column (10 bins: registers).
//   * dp_init_kernel / dp_final_kernel: the latent (injected `dp` noise, or RandomNormalLike on the device), the
//     ElementwiseAffine and w = exp(logw)·length_scale → ceil → int32 frames per id.
// The k = 1 convs around them (pre / proj) are ordinary conv launches (conv.hip).
#include <algorithm>

#include "common.h"
#include "rng.h"

namespace ph {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxBins = 16;

// the graph's Div(√2) → Erf → Add(1) → Mul(x) → Mul(0.5). r3: x·(1/√2) instead of the correctly rounded division (≈ 10 vector
// instructions each; a wave of dds_layer_kernel issued ≈ 2 000 of them and the launch is bound by exactly that — PMC SQ_INSTS_VALU):
// one rounding step (≤ 1 ulp) from the graph's value.
__device__ __forceinline__ float gelu_erf(float v) {
  const float e = erff(v * 0.70710678118654752f);
  return (v * (e + 1.0f)) * 0.5f;
}

// x, out: [N][H][T]. dw_w [H][KD], dw_b [H]; pw16: packed 16-wide fragment image of the 1×1 conv (pack_conv_weights tm = 16),
// pw_steps its padded step count; g1/b1, g2/b2: LayerNorm parameters. Tv = true length (columns ≥ Tv read as zero, are not
// written).
// FAST: H % 64 == 0 and an unpadded fragment image (pw_steps = H / 4): no clamps / selects in the K loop.
// kW waves per block. kW = 12 (H ≤ 192: one 16-row tile of the pointwise conv per wave) is the PREFETCH variant: a wave requests its
// tile's whole weight slab (H / 4 ≤ 48 fragments) and the epilogue's operands at kernel start, before the depthwise phase, so that the
// pointwise phase never waits for memory — with batches of 16 fragments fetched one batch ahead, each batch's ≈ 1 µs cold round trip
// stood behind 0.25 µs of MFMAs (r3: the launch is a chain of such waits, not vector-ALU work: halving the instruction count moved nothing).
#ifdef PH_DDS_TRACE
__device__ unsigned long long* ph_dds_trace_buf;
#define PH_DSTAMP(k) do { if ((threadIdx.x & 63) == 0 && ph_dds_trace_buf) ph_dds_trace_buf[((size_t)blockIdx.x * kW + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_DSTAMP(k) do { } while (0)
#endif
template <int KD, bool FAST, int kW>
__global__ __launch_bounds__(64 * kW) void dds_layer_kernel(const float* __restrict__ x, const float* __restrict__ dw_w,
                                                       const float* __restrict__ dw_b, const float* __restrict__ g1,
                                                       const float* __restrict__ b1, const float* __restrict__ pw16,
                                                       const float* __restrict__ pw_b, const float* __restrict__ g2,
                                                       const float* __restrict__ b2, float* __restrict__ out, int H, int T, int dil,
                                                       int pw_steps, const int* __restrict__ len_ptr, float eps) {
  constexpr int kNT = 64 * kW;
  constexpr int kRP = kNT / 16;                       // channel rows per pass of the depthwise phase
  constexpr int kMaxRows = (256 + kRP - 1) / kRP;     // passes: H ≤ 256
  constexpr int kMaxTiles = (16 + kW - 1) / kW;       // 16-row tiles per wave in the pointwise phase
  constexpr bool PRE = FAST && kW == 12;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* act = sm;             // [H][16]   B operand of the pointwise conv
  float* red = act + H * 16;   // [kRP][16] column partials
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y, t0 = blockIdx.x * 16;
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;
  if (t0 >= Tv) return;
  PH_DSTAMP(0);
  const float* xb = x + (int64_t)n * H * T;
  float* ob = out + (int64_t)n * H * T;
  // ---- 1. depthwise conv: thread ↔ (channel c = tid / 16 + 32·i, column tid % 16)
  const int col = tid & 15, crow = tid >> 4;  // kRP channel rows per pass
  const int t = t0 + col;
  float y[kMaxRows];
  float s1 = 0.0f;
  // every load is unconditional (clamped address, value masked afterwards): with the loads inside the range tests the
  // compiler waits for each one before the next test — 18 dependent round trips, 22 µs per layer at T = 112 (r2 profile)
  float xv[kMaxRows][KD], wv[kMaxRows][KD], bv[kMaxRows], g1v[kMaxRows], b1v[kMaxRows];
#pragma unroll
  for (int i = 0; i < kMaxRows; i++) {
    const int c = min(crow + kRP * i, H - 1);
    bv[i] = dw_b[c];
    g1v[i] = g1[c];
    b1v[i] = b1[c];
#pragma unroll
    for (int k = 0; k < KD; k++) {
      const int pos = t + (k - (KD - 1) / 2) * dil;
      xv[i][k] = xb[(int64_t)c * T + min(max(pos, 0), T - 1)];
      wv[i][k] = dw_w[c * KD + k];
    }
  }
  // PRE: this wave's weight slab, bias, LayerNorm-2 operands and residual — requested right BEHIND the depthwise conv's operands (loads return in
  // order: ahead of them, as until round 3, the first phase waited for 48 weight loads per lane it does not need)
  float apre[PRE ? 48 : 1], pbias[4], pg2[4], pb2[4], pxr[4];
  if constexpr (PRE) {
    const int mtc = min(wave, ((H + 15) >> 4) - 1);  // one tile per wave (≤ 12 tiles)
    const float* wa0 = pw16 + (int64_t)mtc * pw_steps * 64 + lane;
    const int nst0 = H >> 2;
#pragma unroll
    for (int u = 0; u < 48; u++) apre[u] = wa0[min(u, nst0 - 1) * 64];
    const int tc0 = min(t0 + (lane & 15), T - 1);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = min(16 * mtc + 4 * (lane >> 4) + r, H - 1);
      pbias[r] = pw_b[c];
      pg2[r] = g2[c];
      pb2[r] = b2[c];
      pxr[r] = xb[(int64_t)c * T + tc0];
    }
  }
#pragma unroll
  for (int i = 0; i < kMaxRows; i++) {
    const int c = crow + kRP * i;
    float acc = bv[i];  // bias first, then the taps in order (CPUBackend.conv1d with Cin/g = 1)
#pragma unroll
    for (int k = 0; k < KD; k++) {
      const int pos = t + (k - (KD - 1) / 2) * dil;
      acc += ((pos >= 0 && pos < Tv) ? xv[i][k] : 0.0f) * wv[i][k];
    }
    y[i] = c < H ? acc : 0.0f;
    s1 += y[i];
  }
  PH_DSTAMP(1);
  // ---- 2. LayerNorm over the channels of each column (two passes, like the graph) → GELU → act
  red[crow * 16 + col] = s1;
  __syncthreads();
  float mean = 0.0f;
#pragma unroll
  for (int q = 0; q < kRP; q++) mean += red[q * 16 + col];
  mean = mean / (float)H;
  __syncthreads();
  float s2 = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxRows; i++) {
    const int c = crow + kRP * i;
    if (c < H) {
      y[i] -= mean;
      s2 += y[i] * y[i];
    }
  }
  red[crow * 16 + col] = s2;
  __syncthreads();
  float var = 0.0f;
#pragma unroll
  for (int q = 0; q < kRP; q++) var += red[q * 16 + col];
  var = var / (float)H;
  const float rsd = 1.0f / sqrtf(var + eps);  // one division per column instead of one per element (≤ 1 ulp from y / sd)
#pragma unroll
  for (int i = 0; i < kMaxRows; i++) {
    const int c = crow + kRP * i;
    if (c < H) act[c * 16 + col] = gelu_erf((y[i] * rsd) * g1v[i] + b1v[i]);
  }
  PH_DSTAMP(2);
  __syncthreads();
  PH_DSTAMP(3);
  // ---- 3. pointwise conv on the tile: D[row = channel][col]; wave ↔ row tiles wave, wave + 8
  const int r16 = lane & 15, kq = lane >> 4;
  const int ntiles = (H + 15) >> 4;
  const int nst = (H + 3) >> 2;
  float val[kMaxTiles][4];
  float p1 = 0.0f;
#pragma unroll
  for (int ti = 0; ti < kMaxTiles; ti++) {
    const int mt = wave + kW * ti;
#pragma unroll
    for (int r = 0; r < 4; r++) val[ti][r] = 0.0f;
    if (mt < ntiles) {  // wave-uniform
      f32x4 acc;
      if constexpr (PRE) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = pbias[r];
        const float* ab = act + kq * 16 + r16;
#pragma unroll
        for (int u = 0; u < 48; u++)
          if (u < nst) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(apre[u], ab[u * 64], acc, 0, 0, 0);  // wave-uniform bound
      } else {
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = pw_b[min(16 * mt + 4 * kq + r, H - 1)];
      const float* wa = pw16 + (int64_t)mt * pw_steps * 64 + lane;
      // weight fragments two batches of 16 steps ahead of the MFMAs that use them (the batches used to alternate load → wait → multiply)
      if constexpr (FAST) {
        // 16 steps per batch, the next batch's weight fragments in flight: plain immediate-offset loads, no per-step index arithmetic
        const float* ab = act + kq * 16 + r16;
        float a[2][16];
#pragma unroll
        for (int u = 0; u < 16; u++) a[0][u] = wa[u * 64];
        const int nb = nst >> 4;
        for (int bt = 0; bt < nb; bt += 2) {
#pragma unroll
          for (int half = 0; half < 2; half++) {
            const int sb = (bt + half) << 4;
            if (bt + half < nb) {  // wave-uniform
              const int nx = min(sb + 16, nst - 16);  // the last batch re-requests itself (unused)
#pragma unroll
              for (int u = 0; u < 16; u++) a[half ^ 1][u] = wa[(nx + u) * 64];
              const float* abs_ = ab + sb * 64;
#pragma unroll
              for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[half][u], abs_[u * 64], acc, 0, 0, 0);
            }
          }
        }
      } else {
      float a[2][16];
#pragma unroll
      for (int u = 0; u < 16; u++) a[0][u] = wa[min(u, pw_steps - 1) * 64];
      for (int s0 = 0; s0 < nst; s0 += 32) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const int sb = s0 + 16 * half;
#pragma unroll
          for (int u = 0; u < 16; u++) a[half ^ 1][u] = wa[min(sb + 16 + u, pw_steps - 1) * 64];
          float b[16];
#pragma unroll
          for (int u = 0; u < 16; u++) b[u] = act[min(4 * (sb + u) + kq, H - 1) * 16 + r16];
#pragma unroll
          for (int u = 0; u < 16; u++)
            if (sb + u < nst) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[half][u], 4 * (sb + u) + kq < H ? b[u] : 0.0f, acc, 0, 0, 0);
        }
      }
      }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (16 * mt + 4 * kq + r < H) {
          val[ti][r] = acc[r];
          p1 += acc[r];
        }
      }
    }
  }
  PH_DSTAMP(4);
  // ---- 4. second LayerNorm (column on lane & 15, channels on lane >> 4 / register / wave) → GELU → + x
  p1 += __shfl_xor(p1, 16, 64);
  p1 += __shfl_xor(p1, 32, 64);
  if (lane < 16) red[wave * 16 + lane] = p1;
  __syncthreads();
  float mean2 = 0.0f;
#pragma unroll
  for (int q = 0; q < kW; q++) mean2 += red[q * 16 + r16];
  mean2 = mean2 / (float)H;
  __syncthreads();
  float p2 = 0.0f;
#pragma unroll
  for (int ti = 0; ti < kMaxTiles; ti++) {
    const int mt = wave + kW * ti;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (mt < ntiles && 16 * mt + 4 * kq + r < H) {
        val[ti][r] -= mean2;
        p2 += val[ti][r] * val[ti][r];
      }
  }
  p2 += __shfl_xor(p2, 16, 64);
  p2 += __shfl_xor(p2, 32, 64);
  if (lane < 16) red[wave * 16 + lane] = p2;
  __syncthreads();
  float var2 = 0.0f;
#pragma unroll
  for (int q = 0; q < kW; q++) var2 += red[q * 16 + r16];
  var2 = var2 / (float)H;
  const float rsd2 = 1.0f / sqrtf(var2 + eps);
  PH_DSTAMP(5);
  {
    const int tc = min(t0 + r16, T - 1);
    float g2v[kMaxTiles][4], b2v[kMaxTiles][4], xr[kMaxTiles][4];  // loads first (clamped), masked stores after
#pragma unroll
    for (int ti = 0; ti < kMaxTiles; ti++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int c = min(16 * (wave + kW * ti) + 4 * kq + r, H - 1);
        if constexpr (PRE) {
          g2v[ti][r] = pg2[r]; b2v[ti][r] = pb2[r]; xr[ti][r] = pxr[r];
        } else {
          g2v[ti][r] = g2[c];
          b2v[ti][r] = b2[c];
          xr[ti][r] = xb[(int64_t)c * T + tc];
        }
      }
#pragma unroll
    for (int ti = 0; ti < kMaxTiles; ti++) {
      const int mt = wave + kW * ti;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int c = 16 * mt + 4 * kq + r;
        const float hid = gelu_erf((val[ti][r] * rsd2) * g2v[ti][r] + b2v[ti][r]);
        if (t0 + r16 < Tv && mt < ntiles && c < H) ob[(int64_t)c * T + t0 + r16] = xr[ti][r] + hid;
      }
    }
  }
  PH_DSTAMP(6);
}

// per-item scalars of the predictor, in device memory so that a replayed graph sees new values
struct DpScalars {
  float noise_w, length_scale;
  unsigned gen, seed;  // gen ≠ 0: draw the `dp` RandomNormalLike tensor on the device
};

// z [N][2][T] = noise·noise_w, rows already FLIPPED for the first ConvFlow (reverse pass: flip, then flow): row 0 ← noise row 1.
__global__ __launch_bounds__(256) void dp_init_kernel(const float* __restrict__ noise, const DpScalars* __restrict__ sc, float* __restrict__ z, int T,
                                                     const int* __restrict__ len_ptr) {
  const int n = blockIdx.y;
  const DpScalars s = sc[n];
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * T; i += gridDim.x * 256) {
    const int row = i / T, t = i - row * T;
    // source row before the flip; the device draw uses the element index of the item's OWN [1, 2, Tv] tensor (RandomNormalLike
    // mirrors the true shape, not the bucket), the injected tensor was laid out with the bucket's row stride by the host
    float nz = 0.0f;
    if (t < Tv) nz = s.gen ? rnl_normal(s.seed, (unsigned)((1 - row) * Tv + t)) : noise[(int64_t)n * 2 * T + (1 - row) * T + t];
    z[(int64_t)n * 2 * T + i] = nz * s.noise_w;
  }
}

__device__ __forceinline__ float softplus_ref(float v) { return v > 0.0f ? v + logf(1.0f + expf(-v)) : logf(1.0f + expf(v)); }

// ConvFlow tail: z1 ← spline⁻¹(z1; h), then the Flip that precedes the next module, written in place as a row swap:
// on exit row 0 = new z1, row 1 = z0. h [N][3·bins − 1][T].
// NBM = compile-time bound of the bin count (10: every Piper voice; 16: the general case). Every operand of a column — the two latent
// rows and all 3·nb − 1 spline parameters — is requested in ONE burst at kernel start (r3: the parameter loads used to sit behind the
// latent's round trip and a range test, the two derivative loads behind the bin search — three dependent cold round trips, 11 µs per
// launch for 2 waves of arithmetic); the bin's derivatives are then picked out of registers.
template <int NBM>
__global__ __launch_bounds__(256) void dp_spline_kernel(const float* __restrict__ h, float* __restrict__ z, int T, int nb, float B,
                                                       float filter_channels, const int* __restrict__ len_ptr) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;
  const int tc = min(t, T - 1);
  float* zb = z + (int64_t)n * 2 * T;
  const float* hb = h + (int64_t)n * (3 * nb - 1) * T + tc;
  const float z0 = zb[tc], x = zb[T + tc];
  float hw[NBM], hh[NBM], hd[NBM];  // widths, heights, derivatives (nb − 1 of them)
#pragma unroll
  for (int i = 0; i < NBM; i++) {
    const int ic = min(i, nb - 1);
    hw[i] = hb[(int64_t)ic * T];
    hh[i] = hb[(int64_t)(nb + ic) * T];
    hd[i] = hb[(int64_t)(2 * nb + min(i, nb - 2)) * T];
  }
  if (t >= Tv) return;
  float outv = x;
  if (x >= -B && x <= B) {
    const float mbw = 1e-3f, mbh = 1e-3f, md = 1e-3f;
    const float inv = sqrtf(filter_channels);
    float w[NBM], cw[NBM + 1], ch[NBM + 1];
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int i = 0; i < NBM; i++)
      if (i < nb) {
        w[i] = hw[i] / inv;
        hh[i] = hh[i] / inv;
        mw = fmaxf(mw, w[i]);
        mh = fmaxf(mh, hh[i]);
      }
    float sw = 0.0f, sh = 0.0f;
#pragma unroll
    for (int i = 0; i < NBM; i++)
      if (i < nb) { w[i] = expf(w[i] - mw); sw += w[i]; hh[i] = expf(hh[i] - mh); sh += hh[i]; }
    const float isw = 1.0f / sw, ish = 1.0f / sh;  // softmax.metal: multiply by 1/sum
    cw[0] = 0.0f; ch[0] = 0.0f;
#pragma unroll
    for (int i = 0; i < NBM; i++)
      if (i < nb) {
        cw[i + 1] = cw[i] + (mbw + (1.0f - mbw * nb) * (w[i] * isw));
        ch[i + 1] = ch[i] + (mbh + (1.0f - mbh * nb) * (hh[i] * ish));
      }
#pragma unroll
    for (int i = 0; i <= NBM; i++)
      if (i <= nb) { cw[i] = (2.0f * B) * cw[i] + -B; ch[i] = (2.0f * B) * ch[i] + -B; }
    int idx = -1;  // Σ (x ≥ location) − 1, the last location nudged by 1e-6
#pragma unroll
    for (int i = 0; i <= NBM; i++)
      if (i <= nb) {
        if (i == 0) { cw[i] = -B; ch[i] = -B; }
        if (i == nb) { cw[i] = B; ch[i] = B; }
        idx += (x >= (i == nb ? ch[i] + 1e-6f : ch[i])) ? 1 : 0;
      }
    idx = min(max(idx, 0), nb - 1);
    const float cdv = softplus_ref(logf(expf(1.0f - md) - 1.0f));
    // derivative rows idx − 1 and idx, the bin's edges: picked out of the registers loaded above
    float dl = 0.0f, dr = 0.0f, cwl = 0.0f, cwr = 0.0f, chl = 0.0f, chr_ = 0.0f;
#pragma unroll
    for (int i = 0; i < NBM; i++) {
      if (i == idx - 1) dl = hd[i];
      if (i == idx) { dr = hd[i]; cwl = cw[i]; cwr = cw[i + 1]; chl = ch[i]; chr_ = ch[i + 1]; }
    }
    const float d0 = md + (idx == 0 ? cdv : softplus_ref(dl));
    const float d1 = md + (idx == nb - 1 ? cdv : softplus_ref(dr));
    const float ibw = cwr - cwl, ih = chr_ - chl;
    const float idl = ih / ibw;
    const float i1 = d0 + d1 - 2.0f * idl;
    const float i2 = x - chl;
    const float i3 = i2 * i1;
    const float a = ih * (idl - d0) + i3;
    const float b = ih * d0 - i3;
    const float cc = -idl * i2;
    const float disc = b * b - 4.0f * a * cc;
    const float root = (2.0f * cc) / (-b - sqrtf(disc));
    outv = root * ibw + cwl;
  }
  zb[t] = outv;    // Flip: the transformed half becomes row 0 …
  zb[T + t] = z0;  // … and the untouched half row 1
}

// ElementwiseAffine (reverse) on row 0 — the rows were already flipped by the last spline — then logw → frames:
// w = exp(logw)·length_scale, durations = ceil(w) (Piper infer: Exp / Mul / Ceil / Cast).
__global__ __launch_bounds__(256) void dp_final_kernel(const float* __restrict__ z, const float* __restrict__ m, const float* __restrict__ logs,
                                                      const DpScalars* __restrict__ sc, float* __restrict__ logw, int32_t* __restrict__ dur,
                                                      int T, const int* __restrict__ len_ptr) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;
  float lw = 0.0f;
  int32_t d = 0;
  if (t < Tv) {
    lw = (z[(int64_t)n * 2 * T + t] - m[0]) * expf(-logs[0]);
    const float w = expf(lw) * sc[n].length_scale;
    const float cwv = ceilf(w);
    d = cwv > 0.0f ? (cwv < 1e6f ? (int32_t)cwv : 1000000) : 0;  // NaN → 0 like the reference's cast (cast.metal)
  }
  logw[(int64_t)n * T + t] = lw;
  dur[(int64_t)n * T + t] = d;
}

}  // namespace

size_t dp_scalars_bytes(int n) { return sizeof(DpScalars) * (size_t)n; }
void dp_scalars_fill(void* host, int i, float noise_w, float length_scale, unsigned gen, unsigned seed) {
  DpScalars* p = (DpScalars*)host;
  p[i].noise_w = noise_w; p[i].length_scale = length_scale; p[i].gen = gen; p[i].seed = seed;
}

#ifdef PH_DDS_TRACE
void ph_dds_set_trace(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(ph_dds_trace_buf), &buf, sizeof buf); }
#endif
bool dds_layer_eligible(int H, int K) { return H >= 16 && H <= 256 && (K == 1 || K == 3 || K == 5 || K == 7); }

int launch_dds_layer(piper_hip_ctx* ctx, hipStream_t s, const float* x, const float* dw_w, const float* dw_b, const float* g1, const float* b1,
                     const float* pw16, const float* pw_b, const float* g2, const float* b2, float* out, int N, int H, int T, int K, int dil,
                     int pw_steps, const int* len_ptr, float eps) {
  if (N <= 0 || T <= 0) return PIPER_HIP_OK;
  if (!dds_layer_eligible(H, K) || N > 65535) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "dds_layer: H=%d K=%d not covered", H, K);
  const dim3 grid((unsigned)ceil_div(T, 16), (unsigned)N);
  const bool fast = (H % 64) == 0 && pw_steps == H / 4;
  static const bool no_pre = getenv("PIPER_HIP_DDS_NO_PREFETCH") != nullptr;
  const bool pre = fast && H <= 192 && !no_pre;  // ≤ 12 row tiles: one per wave of a 768-thread block, weight slab prefetched
  const size_t lds = ((size_t)H * 16 + (pre ? 48 : 32) * 16) * sizeof(float);
#define PH_DDS_ARGS(NW_) grid, dim3(64 * NW_), lds, s, x, dw_w, dw_b, g1, b1, pw16, pw_b, g2, b2, out, H, T, dil, pw_steps, len_ptr, eps
#define PH_DDS(KK)                                                                             \
  do {                                                                                         \
    if (pre) hipLaunchKernelGGL((dds_layer_kernel<KK, true, 12>), PH_DDS_ARGS(12));            \
    else if (fast) hipLaunchKernelGGL((dds_layer_kernel<KK, true, 8>), PH_DDS_ARGS(8));        \
    else hipLaunchKernelGGL((dds_layer_kernel<KK, false, 8>), PH_DDS_ARGS(8));                 \
  } while (0)
  switch (K) {
    case 1: PH_DDS(1); break;
    case 3: PH_DDS(3); break;
    case 5: PH_DDS(5); break;
    default: PH_DDS(7); break;
  }
#undef PH_DDS
#undef PH_DDS_ARGS
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "dds_layer launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

int launch_dp_init(hipStream_t s, const float* noise, const void* scalars, float* z, int N, int T, const int* len_ptr) {
  const dim3 grid((unsigned)std::min<int64_t>(ceil_div(2 * (int64_t)T, 256), 64), (unsigned)N);
  hipLaunchKernelGGL(dp_init_kernel, grid, dim3(256), 0, s, noise, (const DpScalars*)scalars, z, T, len_ptr);
  return PIPER_HIP_OK;
}
int launch_dp_spline(hipStream_t s, const float* h, float* z, int N, int T, int bins, float tail_bound, float filter_channels, const int* len_ptr) {
  if (bins > kMaxBins || bins < 2) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "dp spline: %d bins (2 … %d)", bins, kMaxBins);
  const dim3 grid((unsigned)ceil_div(T, 64), (unsigned)N);  // one-wave blocks: a column is one thread, and 112 of them should not share a CU
  if (bins <= 10) hipLaunchKernelGGL(dp_spline_kernel<10>, grid, dim3(64), 0, s, h, z, T, bins, tail_bound, filter_channels, len_ptr);
  else hipLaunchKernelGGL(dp_spline_kernel<kMaxBins>, grid, dim3(64), 0, s, h, z, T, bins, tail_bound, filter_channels, len_ptr);
  return PIPER_HIP_OK;
}
int launch_dp_final(hipStream_t s, const float* z, const float* m, const float* logs, const void* scalars, float* logw, int32_t* dur, int N, int T,
                    const int* len_ptr) {
  const dim3 grid((unsigned)ceil_div(T, 256), (unsigned)N);
  hipLaunchKernelGGL(dp_final_kernel, grid, dim3(256), 0, s, z, m, logs, (const DpScalars*)scalars, logw, dur, T, len_ptr);
  return PIPER_HIP_OK;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(dp, dp_init_kernel); } }
