// conv_lean.hip — the k = 1 convs of one utterance (flow res/skip, pre, post; the encoder's first qkv) with as few instructions as the
// job allows.
//
// Reference: conv1d_f32 with one tap (Kernels/conv1d.metal:28-71) behind the Conv arm (GraphExecutor.swift:1739-1810); the WaveNet
// res/skip routing is the Slice / Add arms behind it (:1240-1489, :861-899).
//
// Why (round 3, `tools/probe/floorprobe.hip` → profiles/r3_probe_floor.txt): a kernel with the whole STRUCTURE of such a launch —
// 24 KB of operands per block, a 320-byte argument struct, the split-K exchange through LDS with its barrier, an epilogue load behind
// the barrier, the MFMAs, a dependent true-length load in front of everything — costs 2.4 µs per launch in a captured graph. The
// general streaming kernel costs 5.25 µs for the same job, and the difference is instruction volume: PMC SQ_INSTS_* says one wave of it
// issues ≈ 340 instructions (235 scalar: tile decode with integer divisions, descriptor set-up, ring bookkeeping, a switch over
// epilogues …) for 6 MFMAs, four waves per SIMD take turns on one scalar unit, and 340 × 4 waves × 4 clocks is the 2.7 µs that are
// missing. This kernel is that launch stripped to what it needs:
//   * grid = (row tiles, 16-column chunks, batch): no tile decode; 8 waves split the contraction (Cin / 32 channel quads each,
//     compile-time), so every load is base + lane + immediate;
//   * one tap ⇒ no zero padding, no window, no masks: a column's output depends on that column only, columns past the row end are
//     simply not stored;
//   * the epilogue is a template parameter; after the LDS exchange waves 0 … 3 finish one accumulator register each.
// ≈ 70 instructions per wave. Everything it does not cover falls through to the kernels that do.
#include "conv.h"

namespace ph {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LeanArgs {
  const float *x, *w, *bias, *res, *skip;
  float *y, *y2;
  const int* len_ptr;
  int len_mul, Lin, Lout, Cout, y_len, nsteps, wn_c;
  int x_row_bytes;   // Lin · 4
  int x_base_bytes;  // byte offset of (first physical channel row of quad 0, lane group 0) in the batch item
  int q_stride;      // bytes from one channel quad to the next (± 16 · Lin)
  int kk_sign;       // +1: lane group kk reads row +kk; −1: a reversed channel map, group kk reads row 3 − kk of the (lowered) base
  int out_ch_base, out_ch_sign;
  int x_batch_bytes;
  long long x_bs, y_bs, y2_bs;  // floats between batch items
};

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// NQ channel quads per wave; 8 waves = the whole contraction (Cin = 32 · NQ).
// LONG ROWS (round 3): gridDim.y may be smaller than the number of 16-column chunks — block y then walks chunks y, y + gridDim.y, … with the
// weight fragments it loaded ONCE (the launch used to be one block per chunk, each pulling its 24 … 61 KB slab again: at 2 688 columns the
// general kernels took over at 10–29 µs per launch). The next chunk's operands are requested before the exchange of the current one, the
// exchange buffer alternates so that one barrier per chunk is enough. One chunk per block (a short utterance) is the same code with one trip.
template <int NQ, int MODE>
__global__ __launch_bounds__(512) void conv_k1_kernel(const LeanArgs a, const int nch) {
  __shared__ float red_all[2 * 8 * 4 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mt = blockIdx.x, n = blockIdx.z;
  // bucketed / ragged batches: a chunk past the item's true length produces nothing anyone reads
  const int lim = a.len_ptr ? min(a.len_ptr[n] * a.len_mul, a.Lout) : a.Lout;
  int ch = blockIdx.y;
  if (ch * 16 >= lim) return;
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)n * a.x_bs), 0, a.x_batch_bytes, 0x00020000);
  const float* wa = a.w + (((long long)mt * a.nsteps + wave * NQ) << 6) + lane;
  const int vrow = (a.kk_sign > 0 ? kk : 3 - kk) * a.x_row_bytes;
  const int soff0 = a.x_base_bytes + wave * NQ * a.q_stride;
  float av[NQ], bv[NQ];
#pragma unroll
  for (int i = 0; i < NQ; i++) av[i] = wa[i * 64];
  {
    const int voff = vrow + min(ch * 16 + j, a.Lin - 1) * 4;  // (columns past the row end are not stored)
#pragma unroll
    for (int i = 0; i < NQ; i++) bv[i] = bload(rx, voff, soff0 + i * a.q_stride);
  }
  // wave w (< 4) finishes accumulator register w: rows 16·mt + 4·kk + w; slices added in fixed order on top of the bias
  const int row = 16 * mt + 4 * kk + wave;
  const float bias = (wave < 4 && a.bias) ? a.bias[min(row, a.Cout - 1)] : 0.0f;  // bias first (CPUBackend.swift:46-63)
  for (int it = 0;; it++) {
    const int t0 = ch * 16;
    const int nxt = ch + (int)gridDim.y;
    const bool more = nxt < nch && nxt * 16 < lim;  // block-uniform
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < NQ; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
    if (more) {  // the next chunk's operands: on their way during the exchange and the epilogue
      const int voff = vrow + min(nxt * 16 + j, a.Lin - 1) * 4;
#pragma unroll
      for (int i = 0; i < NQ; i++) bv[i] = bload(rx, voff, soff0 + i * a.q_stride);
    }
    float* red = red_all + (it & 1) * (8 * 4 * 64);
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave < 4) {
      const int col = t0 + j;
      float v = bias;
      float part[8];
#pragma unroll
      for (int s = 0; s < 8; s++) part[s] = red[(s * 4 + wave) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 8; s++) v += part[s];
      if (row < a.Cout && col < a.Lout) {
        if constexpr (MODE == EPI_STORE) {
          const long long idx = (long long)n * a.y_bs + (a.out_ch_base + a.out_ch_sign * row) * a.y_len + col;
          a.y[idx] = a.res ? v + a.res[idx] : v;
        } else if constexpr (MODE == EPI_RSUB) {
          const long long idx = (long long)n * a.y_bs + (a.out_ch_base + a.out_ch_sign * row) * a.y_len + col;
          a.y[idx] = a.res[idx] - v;
        } else if constexpr (MODE == EPI_WN_RES_SKIP) {
          if (row < a.wn_c) {
            const long long idx = (long long)n * a.y_bs + row * a.y_len + col;
            a.y[idx] = a.res[idx] + v;
          } else {
            const long long idx = (long long)n * a.y2_bs + (row - a.wn_c) * a.y_len + col;
            a.y2[idx] = (a.skip ? a.skip[idx] : 0.0f) + v;
          }
        } else {  // EPI_WN_SKIP_LAST
          const long long idx = (long long)n * a.y2_bs + row * a.y_len + col;
          a.y2[idx] = (a.skip ? a.skip[idx] : 0.0f) + v;
        }
      }
    }
    if (!more) break;
    ch = nxt;
  }
}

// ---- the WaveNet gated conv (k taps, tanh·sigmoid) of one utterance, same diet ----
// Weights: the gate-packed 16-row image (8 tanh rows + their 8 sigmoid rows per tile, pack_conv_weights_gate16). 8 waves split the
// contraction (NQ channel quads each); a wave stages the [4·NQ] × [16 + K − 1] window of ITS channels in its own piece of LDS — the
// main 16 columns four rows per load, the K − 1 halo columns sixteen rows per load; a position outside [0, true length) is requested
// at an out-of-range offset, which returns the conv's zero padding without touching memory — and feeds every tap from it.
struct GateArgs {
  const float *x, *w, *bias;
  float* y;
  const int* len_ptr;
  int len_mul, Lin, Lout, rows_out, y_len, nsteps, pad, x_batch_bytes;
  long long x_bs, y_bs;
};

template <int K, int NQ>
__global__ __launch_bounds__(512) void conv_gate_kernel(const GateArgs a, const int nch) {
  constexpr int W = 16 + K - 1, PITCH = (W + 3) & ~3, ROWS = 4 * NQ, NS = NQ * K;
  constexpr int NLB = (ROWS + 15) / 16;  // halo loads: 16 rows × 4 columns each (K − 1 ≤ 4)
  constexpr int OOB = 0x7fffffff;
  static_assert(K - 1 <= 4, "halo piece is four columns wide");
  __shared__ float xs_all[8 * ROWS * PITCH];
  __shared__ float red_all[2 * 8 * 4 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mt = blockIdx.x, n = blockIdx.z;
  const int Lv = a.len_ptr ? min(a.len_ptr[n] * a.len_mul, a.Lin) : a.Lin;
  int ch = blockIdx.y;
  if (ch * 16 >= Lv) return;  // a chunk past the item's true length produces nothing anyone reads
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)n * a.x_bs), 0, a.x_batch_bytes, 0x00020000);
  const float* wa = a.w + (((long long)mt * a.nsteps + wave * NS) << 6) + lane;
  float av[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) av[s] = wa[s * 64];  // (requesting the window first measured 0.2 µs SLOWER here, unlike in the k = 3 kernels below)
  // window: main piece (column j of rows 4i + kk), halo piece (column 16 + (lane & 3) of rows 16i + (lane >> 2))
  const int rB = lane >> 2, cB = 16 + (lane & 3);
  const int sbase = wave * ROWS * a.Lin * 4;
  float xa[NQ], xb[NLB];
  auto request = [&](int t0) {
    const int posA = t0 - a.pad + j;
    const int voffA = (posA >= 0 && posA < Lv) ? (kk * a.Lin + posA) * 4 : OOB;
    const int posB = t0 - a.pad + cB;
    const int voffB = ((lane & 3) < K - 1 && posB >= 0 && posB < Lv) ? (rB * a.Lin + posB) * 4 : OOB;
#pragma unroll
    for (int i = 0; i < NQ; i++) xa[i] = bload(rx, voffA, sbase + i * 16 * a.Lin);
#pragma unroll
    for (int i = 0; i < NLB; i++) xb[i] = bload(rx, (rB + 16 * i < ROWS) ? voffB : OOB, sbase + i * 64 * a.Lin);
  };
  request(ch * 16);
  float* xs = xs_all + wave * (ROWS * PITCH);
  const float* xw = xs + kk * PITCH + j;
  // wave w (< 4) finishes register w: tile rows i = 4·kk + w — lanes 0–31 hold tanh rows (i < 8), lanes 32–63 their sigmoid partners
  const int i8 = 4 * kk + wave;
  const int h = 8 * mt + (i8 & 7);
  const float bias = (wave < 4 && a.bias) ? a.bias[(i8 < 8 ? 0 : a.rows_out) + min(h, a.rows_out - 1)] : 0.0f;  // bias first (CPUBackend.swift:46-63)
  for (int it = 0;; it++) {  // chunks ch, ch + gridDim.y, … with the same weight fragments (see conv_k1_kernel)
    const int t0 = ch * 16;
    const int nxt = ch + (int)gridDim.y;
    const bool more = nxt < nch && nxt * 16 < Lv;  // block-uniform
#pragma unroll
    for (int i = 0; i < NQ; i++) xs[(4 * i + kk) * PITCH + j] = xa[i];
#pragma unroll
    for (int i = 0; i < NLB; i++)
      if ((lane & 3) < K - 1 && rB + 16 * i < ROWS) xs[(rB + 16 * i) * PITCH + cB] = xb[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (more) request(nxt * 16);  // on its way during the MFMAs, the exchange and the epilogue
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int qi = 0; qi < NQ; qi++)
#pragma unroll
      for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[qi * K + k], xw[4 * qi * PITCH + k], acc, 0, 0, 0);
    float* red = red_all + (it & 1) * (8 * 4 * 64);
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave < 4) {
      float v = bias;
      float part[8];
#pragma unroll
      for (int s = 0; s < 8; s++) part[s] = red[(s * 4 + wave) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 8; s++) v += part[s];
      const float sg = __shfl_xor(v, 32, 64);
      const int col = t0 + j;
      if (lane < 32 && h < a.rows_out && col < a.Lout) {
        const float ta = tanhf(v);
        float g;  // the stable sigmoid form of elementwise.metal:253-268
        if (sg >= 0.0f) { const float z = expf(-sg); g = 1.0f / (1.0f + z); }
        else { const float z = expf(sg); g = z / (1.0f + z); }
        a.y[(long long)n * a.y_bs + (long long)h * a.y_len + col] = ta * g;
      }
    }
    if (!more) break;
    ch = nxt;
  }
}

// ---- the same two shapes behind a LayerNorm that the CONSUMER computes itself (ConvArgs::ln_self) ----
// A block holds every channel of its columns (8 waves × 4·NQ rows), so the channel LayerNorm of the graph (ReduceMean / Sub / Pow /
// ReduceMean / Add ε / Sqrt / Div / Mul γ / Add β, GraphExecutor.swift:2071-2125) needs nothing from the producer: each wave reduces its
// rows per column (lane shuffles), the waves exchange (Σ, centred Σ²) through LDS and combine them with Chan's formula (two-pass
// accuracy), and the operand is normalised in registers before it meets the matrix pipe. Until round 3 the producer's epilogue wrote
// per-slot statistics and the consumer re-read them (12 float2 loads per column and wave, ≈ 650 vector instructions per wave in the
// k = 3 case). The blocks of row tile 0 also write the normalised tensor (the residual operand of the next Add, the `enc_out` tap).
struct LnArgs {
  const float *gamma, *beta;
  float* ln_out;
  float eps;
};

template <int NQ>
__global__ __launch_bounds__(512) void conv_k1_ln_kernel(const LeanArgs a, const LnArgs ln) {
  constexpr int RW = 4 * NQ, C = 8 * RW;  // rows per wave, channels
  __shared__ float red[8 * 4 * 64];
  __shared__ float2 st[8 * 16];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mt = blockIdx.x, t0 = blockIdx.y * 16, n = blockIdx.z;
  if (a.len_ptr) {
    if (t0 >= a.len_ptr[n] * a.len_mul) return;
  }
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)n * a.x_bs), 0, a.x_batch_bytes, 0x00020000);
  const float* wa = a.w + (((long long)mt * a.nsteps + wave * NQ) << 6) + lane;
  const int voff = kk * a.x_row_bytes + min(t0 + j, a.Lin - 1) * 4;
  const int soff0 = wave * NQ * a.q_stride;
  float av[NQ], bv[NQ], gv[NQ], bev[NQ];
  // in the order of use (loads return in order): the columns the statistics need, γ / β, then the weight fragments and the bias
#pragma unroll
  for (int i = 0; i < NQ; i++) bv[i] = bload(rx, voff, soff0 + i * a.q_stride);
#pragma unroll
  for (int i = 0; i < NQ; i++) {
    gv[i] = ln.gamma[4 * (wave * NQ + i) + kk];
    bev[i] = ln.beta[4 * (wave * NQ + i) + kk];
  }
#pragma unroll
  for (int i = 0; i < NQ; i++) av[i] = wa[i * 64];
  // the epilogue's bias with the first burst (waves 0–3 finish one accumulator register each: rows 16·mt + 4·kk + wave)
  const float bias_early = (wave < 4 && a.bias) ? a.bias[min(16 * mt + 4 * kk + wave, a.Cout - 1)] : 0.0f;
  // the wave's rows of column j: Σ and the Σ² centred on the wave's own mean
  float s1 = 0.0f;
#pragma unroll
  for (int i = 0; i < NQ; i++) s1 += bv[i];
  s1 += __shfl_xor(s1, 16, 64);
  s1 += __shfl_xor(s1, 32, 64);
  const float mw = s1 * (1.0f / RW);
  float q2 = 0.0f;
#pragma unroll
  for (int i = 0; i < NQ; i++) q2 += (bv[i] - mw) * (bv[i] - mw);
  q2 += __shfl_xor(q2, 16, 64);
  q2 += __shfl_xor(q2, 32, 64);
  if (lane < 16) st[wave * 16 + lane] = make_float2(s1, q2);
  __syncthreads();
  float2 pr[8];
#pragma unroll
  for (int w = 0; w < 8; w++) pr[w] = st[w * 16 + j];
  float tot = 0.0f;
#pragma unroll
  for (int w = 0; w < 8; w++) tot += pr[w].x;
  const float mean = tot / (float)C;
  float m2 = 0.0f;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const float dm = pr[w].x * (1.0f / RW) - mean;
    m2 += pr[w].y + (float)RW * (dm * dm);
  }
  const float rstd = 1.0f / sqrtf(m2 / (float)C + ln.eps);
#pragma unroll
  for (int i = 0; i < NQ; i++) bv[i] = ((bv[i] - mean) * rstd) * gv[i] + bev[i];
  if (mt == 0 && ln.ln_out && t0 + j < a.Lin) {  // block-uniform but for the column test
    float* lo = ln.ln_out + (long long)n * a.x_bs + (long long)(4 * wave * NQ + kk) * a.Lin + t0 + j;
#pragma unroll
    for (int i = 0; i < NQ; i++) lo[(long long)4 * i * a.Lin] = bv[i];
  }
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NQ; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave >= 4) return;
  const int row = 16 * mt + 4 * kk + wave, col = t0 + j;
  float v = bias_early;
  float part[8];
#pragma unroll
  for (int s = 0; s < 8; s++) part[s] = red[(s * 4 + wave) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 8; s++) v += part[s];
  if (row < a.Cout && col < a.Lout) a.y[(long long)n * a.y_bs + (long long)row * a.y_len + col] = v;
}

// k = 3 ('same' padding 1 / 1) behind the self-computed LayerNorm, ReLU or plain store out: the encoder FFN's first conv.
template <int NQ, int MODE>
__global__ __launch_bounds__(512) void conv_k3_ln_kernel(const LeanArgs a, const LnArgs ln) {
  constexpr int K = 3, W = 18, PITCH = 20, RW = 4 * NQ, C = 8 * RW, NS = NQ * K;
  constexpr int NLB = (RW + 15) / 16;
  constexpr int OOB = 0x7fffffff;
  __shared__ float xs_all[8 * RW * PITCH];
  __shared__ float red[8 * 4 * 64];
  __shared__ float2 st[8 * W];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mt = blockIdx.x, t0 = blockIdx.y * 16, n = blockIdx.z;
  int Lv = a.Lin;
  if (a.len_ptr) {
    Lv = min(a.len_ptr[n] * a.len_mul, a.Lin);
    if (t0 >= Lv) return;
  }
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)n * a.x_bs), 0, a.x_batch_bytes, 0x00020000);
  const float* wa = a.w + (((long long)mt * a.nsteps + wave * NS) << 6) + lane;
  float av[NS];
  // window columns 0 … 15 ↔ positions t0 − 1 … t0 + 14 (main piece), columns 16, 17 ↔ t0 + 15, t0 + 16 (halo piece)
  const int posA = t0 - 1 + j;
  const bool okA = posA >= 0 && posA < Lv;
  const int voffA = okA ? (kk * a.Lin + posA) * 4 : OOB;
  const int rB = lane >> 2, cB = lane & 3;
  const int posB = t0 + 15 + cB;
  const bool okB = cB < 2 && posB < Lv;
  const int voffB = okB ? (rB * a.Lin + posB) * 4 : OOB;
  const int sbase = wave * RW * a.Lin * 4;
  const int ch0 = wave * RW;  // first channel of this wave
  float xa[NQ], xb[NLB], ga[NQ], ba[NQ], gb[NLB], bb[NLB];
#pragma unroll
  for (int i = 0; i < NQ; i++) xa[i] = bload(rx, voffA, sbase + i * 16 * a.Lin);
#pragma unroll
  for (int i = 0; i < NLB; i++) xb[i] = bload(rx, (rB + 16 * i < RW) ? voffB : OOB, sbase + i * 64 * a.Lin);
#pragma unroll
  for (int i = 0; i < NQ; i++) {
    ga[i] = ln.gamma[ch0 + 4 * i + kk];
    ba[i] = ln.beta[ch0 + 4 * i + kk];
  }
#pragma unroll
  for (int i = 0; i < NLB; i++) {
    const int ch = ch0 + min(rB + 16 * i, RW - 1);
    gb[i] = ln.gamma[ch];
    bb[i] = ln.beta[ch];
  }
  // the weight fragments LAST: loads return in order, and the statistics / the normalised window come before the first MFMA
#pragma unroll
  for (int s = 0; s < NS; s++) av[s] = wa[s * 64];
  const float bias_early = (wave < 4 && a.bias) ? a.bias[min(16 * mt + 4 * kk + wave, a.Cout - 1)] : 0.0f;  // with the first burst
  // per window column: Σ and centred Σ² over this wave's RW rows — main columns by lanes (kk, j), halo columns by lanes (rB, cB)
  float s1 = 0.0f, h1 = 0.0f;
#pragma unroll
  for (int i = 0; i < NQ; i++) s1 += xa[i];
  s1 += __shfl_xor(s1, 16, 64);
  s1 += __shfl_xor(s1, 32, 64);
#pragma unroll
  for (int i = 0; i < NLB; i++) h1 += (rB + 16 * i < RW) ? xb[i] : 0.0f;
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) h1 += __shfl_xor(h1, m, 64);
  const float mwA = s1 * (1.0f / RW), mwB = h1 * (1.0f / RW);
  float q2 = 0.0f, hq = 0.0f;
#pragma unroll
  for (int i = 0; i < NQ; i++) q2 += (xa[i] - mwA) * (xa[i] - mwA);
  q2 += __shfl_xor(q2, 16, 64);
  q2 += __shfl_xor(q2, 32, 64);
#pragma unroll
  for (int i = 0; i < NLB; i++) hq += (rB + 16 * i < RW) ? (xb[i] - mwB) * (xb[i] - mwB) : 0.0f;
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) hq += __shfl_xor(hq, m, 64);
  if (lane < 16) st[wave * W + lane] = make_float2(s1, q2);
  if (lane < 2) st[wave * W + 16 + lane] = make_float2(h1, hq);  // lanes 0, 1: rB = 0, cB = lane
  __syncthreads();
  auto column_stats = [&](const int colw, float& mean, float& rstd) {
    float2 pr[8];
#pragma unroll
    for (int w = 0; w < 8; w++) pr[w] = st[w * W + colw];
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) tot += pr[w].x;
    mean = tot / (float)C;
    float m2 = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      const float dm = pr[w].x * (1.0f / RW) - mean;
      m2 += pr[w].y + (float)RW * (dm * dm);
    }
    rstd = 1.0f / sqrtf(m2 / (float)C + ln.eps);
  };
  float meanA, rstdA, meanB, rstdB;
  column_stats(j, meanA, rstdA);
  column_stats(16 + min(cB, 1), meanB, rstdB);
  float* xs = xs_all + wave * (RW * PITCH);
  const bool wr = mt == 0 && ln.ln_out != nullptr;
  float* lo = ln.ln_out + (long long)n * a.x_bs;
#pragma unroll
  for (int i = 0; i < NQ; i++) {
    const float v = okA ? ((xa[i] - meanA) * rstdA) * ga[i] + ba[i] : 0.0f;  // the conv zero-pads the NORMALISED tensor
    xs[(4 * i + kk) * PITCH + j] = v;
    if (wr && okA && j >= 1) lo[(long long)(ch0 + 4 * i + kk) * a.Lin + posA] = v;
  }
#pragma unroll
  for (int i = 0; i < NLB; i++) {
    const float v = okB ? ((xb[i] - meanB) * rstdB) * gb[i] + bb[i] : 0.0f;
    if (cB < 2 && rB + 16 * i < RW) {
      xs[(rB + 16 * i) * PITCH + 16 + cB] = v;
      if (wr && okB && cB == 0) lo[(long long)(ch0 + rB + 16 * i) * a.Lin + posB] = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const float* xw = xs + kk * PITCH + j;
#pragma unroll
  for (int qi = 0; qi < NQ; qi++)
#pragma unroll
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[qi * K + k], xw[4 * qi * PITCH + k], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave >= 4) return;
  const int row = 16 * mt + 4 * kk + wave, col = t0 + j;
  float v = bias_early;
  float part[8];
#pragma unroll
  for (int s = 0; s < 8; s++) part[s] = red[(s * 4 + wave) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 8; s++) v += part[s];
  if constexpr (MODE == EPI_RELU) v = v > 0.0f ? v : 0.0f;
  if (row < a.Cout && col < a.Lout) a.y[(long long)n * a.y_bs + (long long)row * a.y_len + col] = v;
}

// ---- k = 3 with MANY input channels and few output rows (the FFN's second conv: 768 → 192 on ≤ 900 columns) ----
// 16-row tiles give 12 × 7 = 84 blocks at factor 8, each pulling a 147 KB weight slab: a third of the chip busy on the launch's
// largest byte stream. Here a tile has EIGHT rows (fragment image `w8`: 128 bytes per contraction step; rows 8 … 15 of the MFMA tile
// are zero and cost nothing to fetch — their lanes ask at an out-of-range offset): twice the blocks, half the slab each.
// 8 waves × NQ channel quads; the wave's [4·NQ] × 18 window goes through its own piece of LDS as in conv_gate_kernel.
template <int NQ>
__global__ __launch_bounds__(512) void conv_k3_r8_kernel(const LeanArgs a, const int w_bytes) {
  constexpr int K = 3, PITCH = 20, RW = 4 * NQ, NS = NQ * K;
  constexpr int NLB = (RW + 15) / 16;
  constexpr int OOB = 0x7fffffff;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;                      // [8][4][64]
  float* xs = sm + 8 * 4 * 64 + (threadIdx.x >> 6) * (RW * PITCH);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mt = blockIdx.x, t0 = blockIdx.y * 16, n = blockIdx.z;
  int Lv = a.Lin;
  if (a.len_ptr) {
    Lv = min(a.len_ptr[n] * a.len_mul, a.Lin);
    if (t0 >= Lv) return;
  }
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)n * a.x_bs), 0, a.x_batch_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
  const int voffW = j < 8 ? (kk * 8 + j) * 4 : OOB;  // A[i = j][k = kk] of an 8-row fragment
  const int sW = (mt * a.nsteps + wave * NS) * 128;
  float av[NS];
  const int posA = t0 - 1 + j;
  const int voffA = (posA >= 0 && posA < Lv) ? (kk * a.Lin + posA) * 4 : OOB;
  const int rB = lane >> 2, cB = lane & 3;
  const int posB = t0 + 15 + cB;
  const int voffB = (cB < 2 && posB < Lv) ? (rB * a.Lin + posB) * 4 : OOB;
  const int sbase = wave * RW * a.Lin * 4;
  float xa[NQ], xb[NLB];
#pragma unroll
  for (int i = 0; i < NQ; i++) xa[i] = bload(rx, voffA, sbase + i * 16 * a.Lin);
#pragma unroll
  for (int i = 0; i < NLB; i++) xb[i] = bload(rx, (rB + 16 * i < RW) ? voffB : OOB, sbase + i * 64 * a.Lin);
  // the 72 weight fragments BEHIND the window (loads return in order; the window goes through LDS before the first MFMA)
#pragma unroll
  for (int s = 0; s < NS; s++) av[s] = bload(rw, voffW, sW + s * 128);
  const float bias_early = (wave < 4 && a.bias) ? a.bias[min(8 * mt + 4 * kk + wave, a.Cout - 1)] : 0.0f;  // with the first burst
#pragma unroll
  for (int i = 0; i < NQ; i++) xs[(4 * i + kk) * PITCH + j] = xa[i];
#pragma unroll
  for (int i = 0; i < NLB; i++)
    if (cB < 2 && rB + 16 * i < RW) xs[(rB + 16 * i) * PITCH + 16 + cB] = xb[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const float* xw = xs + kk * PITCH + j;
#pragma unroll
  for (int qi = 0; qi < NQ; qi++)
#pragma unroll
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[qi * K + k], xw[4 * qi * PITCH + k], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave >= 4) return;
  // wave w finishes register w: tile rows 4·kk + w — only kk < 2 are real rows (lanes 0–31)
  const int row = 8 * mt + 4 * kk + wave, col = t0 + j;
  float v = bias_early;
  float part[8];
#pragma unroll
  for (int s = 0; s < 8; s++) part[s] = red[(s * 4 + wave) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 8; s++) v += part[s];
  if (lane < 32 && row < a.Cout && col < a.Lout) {
    const long long idx = (long long)n * a.y_bs + (long long)row * a.y_len + col;
    a.y[idx] = a.res ? v + a.res[idx] : v;
  }
}

template <int NQ>
bool launch_lean_nq(hipStream_t s, dim3 grid, const LeanArgs& a, int mode, int nch) {
  switch (mode) {
    case EPI_STORE: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_STORE>), grid, dim3(512), 0, s, a, nch); return true;
    case EPI_RSUB: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_RSUB>), grid, dim3(512), 0, s, a, nch); return true;
    case EPI_WN_RES_SKIP: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_WN_RES_SKIP>), grid, dim3(512), 0, s, a, nch); return true;
    case EPI_WN_SKIP_LAST: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_WN_SKIP_LAST>), grid, dim3(512), 0, s, a, nch); return true;
  }
  return false;
}

// blocks along the columns: one per 16-column chunk while that keeps the launch within `per_cu` blocks per CU, else as many as that allows —
// each then walks several chunks with the weight fragments it holds
int chunk_groups(piper_hip_ctx* ctx, int row_tiles, int nch, int N) {
  static const int per_cu = [] { const char* e = getenv("PIPER_HIP_LEAN_BLOCKS_PER_CU"); return e ? std::max(1, atoi(e)) : 4; }();
  const int64_t want = (int64_t)per_cu * ctx->num_cus / std::max<int64_t>(1, (int64_t)row_tiles * N);
  return (int)std::max<int64_t>(1, std::min<int64_t>(nch, want));
}

}  // namespace

// Can a conv behind a LayerNorm compute the statistics itself (ConvArgs::ln_self)? The block must hold every channel (192 = 8 waves × 24
// rows) and the launch must be one this file takes (few tiles). The schedule builder asks BEFORE it decides how the producer hands the
// LayerNorm over (voice.hip), so producer and consumer always agree.
bool conv_lean_ln_self_ok(piper_hip_ctx* ctx, int Cin, int Cout, int K, int padL, int L, int N) {
  static const bool off = getenv("PIPER_HIP_NO_LEAN") != nullptr || getenv("PIPER_HIP_NO_LN_SELF") != nullptr;
  if (off || Cin != 192 || (K != 1 && K != 3) || padL != (K - 1) / 2 || L < 1 || N < 1 || N > 65535) return false;
  const int64_t tiles = ceil_div(Cout, 16) * ceil_div(L, 16) * (int64_t)N;
  return tiles <= 8 * (int64_t)ctx->num_cus && ceil_div(L, 16) <= 65535 && (int64_t)Cin * L * 4 < 0x7fffffffLL;
}

// 1 = enqueued, 0 = not this kernel's case, < 0 = error. Called by launch_conv_mfma ahead of the general short-row kernels.
int try_launch_conv_lean(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& c) {
  static const bool off = getenv("PIPER_HIP_NO_LEAN") != nullptr;
  if (off) return 0;
  if (c.gate) {  // the flow's gated conv: k 5 (3), 192 (96) channels, identity channel maps
    if (!c.w16g || c.prologue != PRO_NONE || c.epilogue != EPI_STORE || c.res || c.stats_out || c.dil != 1 || c.Lin != c.Lout || (c.Cout % 16)) return 0;
    if (c.in_ch_sign != 1 || c.in_ch_base != 0 || c.out_ch_sign != 1 || c.out_ch_base != 0 || c.N > 65535) return 0;
    if ((c.K != 5 && c.K != 3) || c.Cin != 192 || c.padL != (c.K - 1) / 2) return 0;
    const int mt_g = c.Cout / 16, nch = (int)ceil_div(c.Lout, 16);
    if ((int64_t)mt_g * nch * c.N > 64 * (int64_t)ctx->num_cus || c.x_batch_stride * 4 >= 0x7fffffffLL) return 0;  // very long rows: the kernels that tile the columns through LDS
    GateArgs g;
    g.x = c.x; g.w = c.w16g; g.bias = c.bias; g.y = c.y; g.len_ptr = c.len_ptr; g.len_mul = c.len_mul;
    g.Lin = c.Lin; g.Lout = c.Lout; g.rows_out = c.Cout / 2; g.y_len = c.y_len; g.nsteps = (c.Cin / 4) * c.K; g.pad = c.padL;
    g.x_batch_bytes = (int)(c.x_batch_stride * 4); g.x_bs = c.x_batch_stride; g.y_bs = c.y_batch_stride;
    const dim3 grid((unsigned)mt_g, (unsigned)chunk_groups(ctx, mt_g, nch, c.N), (unsigned)c.N);
    if (c.K == 5) hipLaunchKernelGGL((conv_gate_kernel<5, 6>), grid, dim3(512), 0, s, g, nch);
    else hipLaunchKernelGGL((conv_gate_kernel<3, 6>), grid, dim3(512), 0, s, g, nch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_gate launch failed: %s", hipGetErrorString(e));
    return 1;
  }
  if (c.prologue == PRO_LN && c.ln_self) {  // LayerNorm computed by the consumer itself: qkv / proj (k 1) and the FFN's first conv (k 3)
    if (!conv_lean_ln_self_ok(ctx, c.Cin, c.Cout, c.K, c.padL, c.Lout, c.N)) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv: ln_self on a shape conv_lean does not cover");
    if (!c.w16 || c.stats_out || c.res || c.dil != 1 || c.Lin != c.Lout || c.in_ch_sign != 1 || c.in_ch_base != 0 || c.out_ch_sign != 1 || c.out_ch_base != 0 ||
        !c.ln_gamma || !c.ln_beta || (c.epilogue != EPI_STORE && c.epilogue != EPI_RELU) || (c.K == 1 && c.epilogue != EPI_STORE))
      PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv: ln_self needs the plain channel maps and a store / ReLU epilogue");
    LeanArgs a;
    a.x = c.x; a.w = c.w16; a.bias = c.bias; a.res = nullptr; a.skip = nullptr; a.y = c.y; a.y2 = nullptr;
    a.len_ptr = c.len_ptr; a.len_mul = c.len_mul;
    a.Lin = c.Lin; a.Lout = c.Lout; a.Cout = c.Cout; a.y_len = c.y_len; a.wn_c = 0;
    a.nsteps = (c.Cin / 4) * c.K;
    a.x_row_bytes = c.Lin * 4; a.kk_sign = 1; a.x_base_bytes = 0; a.q_stride = 16 * c.Lin;
    a.out_ch_base = 0; a.out_ch_sign = 1;
    a.x_batch_bytes = (int)(c.x_batch_stride * 4);
    a.x_bs = c.x_batch_stride; a.y_bs = c.y_batch_stride; a.y2_bs = 0;
    LnArgs ln{c.ln_gamma, c.ln_beta, c.ln_out, c.ln_eps};
    const dim3 grid((unsigned)ceil_div(c.Cout, 16), (unsigned)ceil_div(c.Lout, 16), (unsigned)c.N);
    if (c.K == 1) hipLaunchKernelGGL((conv_k1_ln_kernel<6>), grid, dim3(512), 0, s, a, ln);
    else if (c.epilogue == EPI_RELU) hipLaunchKernelGGL((conv_k3_ln_kernel<6, EPI_RELU>), grid, dim3(512), 0, s, a, ln);
    else hipLaunchKernelGGL((conv_k3_ln_kernel<6, EPI_STORE>), grid, dim3(512), 0, s, a, ln);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_lean (LayerNorm) launch failed: %s", hipGetErrorString(e));
    return 1;
  }
  if (c.K == 3 && c.w8 && c.Cin == 768 && c.prologue == PRO_NONE && c.epilogue == EPI_STORE && !c.stats_out && !c.gate && c.dil == 1 && c.padL == 1 &&
      c.Lin == c.Lout && c.in_ch_sign == 1 && c.in_ch_base == 0 && c.out_ch_sign == 1 && c.out_ch_base == 0 && c.N <= 65535) {
    const int mt8 = (int)ceil_div(c.Cout, 8), nch = (int)ceil_div(c.Lout, 16);
    // up to two blocks per CU: beyond that the 73 KB slab per block costs more than the idle CUs did (factor 64, 1 344 blocks: 36.7 µs against
    // 17.8 on the streaming kernel; factor 8, 168 blocks: 10.0 against 10.8)
    if ((int64_t)mt8 * nch * c.N <= 2 * (int64_t)ctx->num_cus && nch <= 65535 && c.x_batch_stride * 4 < 0x7fffffffLL) {
      LeanArgs a;
      a.x = c.x; a.w = c.w8; a.bias = c.bias; a.res = c.res; a.skip = nullptr; a.y = c.y; a.y2 = nullptr;
      a.len_ptr = c.len_ptr; a.len_mul = c.len_mul;
      a.Lin = c.Lin; a.Lout = c.Lout; a.Cout = c.Cout; a.y_len = c.y_len; a.wn_c = 0; a.nsteps = (c.Cin / 4) * c.K;
      a.x_row_bytes = c.Lin * 4; a.kk_sign = 1; a.x_base_bytes = 0; a.q_stride = 16 * c.Lin; a.out_ch_base = 0; a.out_ch_sign = 1;
      a.x_batch_bytes = (int)(c.x_batch_stride * 4); a.x_bs = c.x_batch_stride; a.y_bs = c.y_batch_stride; a.y2_bs = 0;
      const int w_bytes = mt8 * a.nsteps * 128;
      constexpr int NQ = 24;
      const size_t lds = (size_t)(8 * 4 * 64 + 8 * (4 * NQ) * 20) * sizeof(float);
      static bool configured[kMaxDevices] = {};
      if (lds_optin_needed(configured))
        (void)hipFuncSetAttribute((const void*)conv_k3_r8_kernel<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((conv_k3_r8_kernel<NQ>), dim3((unsigned)mt8, (unsigned)nch, (unsigned)c.N), dim3(512), lds, s, a, w_bytes);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_lean (8-row tiles) launch failed: %s", hipGetErrorString(e));
      return 1;
    }
  }
  if (c.K != 1 || c.prologue != PRO_NONE || c.stats_out || !c.w16) return 0;
  if (c.epilogue != EPI_STORE && c.epilogue != EPI_RSUB && c.epilogue != EPI_WN_RES_SKIP && c.epilogue != EPI_WN_SKIP_LAST) return 0;
  if ((c.epilogue == EPI_RSUB || c.epilogue == EPI_WN_RES_SKIP) && !c.res) return 0;
  if (c.Cin % 32 || c.padL != 0 || c.Lin != c.Lout || c.N > 65535 || c.Lout < 1) return 0;
  const int NQ = c.Cin / 32;
  if (NQ != 3 && NQ != 6) return 0;
  if (c.in_ch_sign != 1 && c.in_ch_sign != -1) return 0;
  const int mtiles = (int)ceil_div(c.Cout, 16), nchunks = (int)ceil_div(c.Lout, 16);
  if ((int64_t)mtiles * nchunks * c.N > 64 * (int64_t)ctx->num_cus) return 0;  // very long rows: the kernels that tile the columns through LDS
  if (c.x_batch_stride * 4 >= 0x7fffffffLL) return 0;
  LeanArgs a;
  a.x = c.x; a.w = c.w16; a.bias = c.bias; a.res = c.res; a.skip = c.skip; a.y = c.y; a.y2 = c.y2;
  a.len_ptr = c.len_ptr; a.len_mul = c.len_mul;
  a.Lin = c.Lin; a.Lout = c.Lout; a.Cout = c.Cout; a.y_len = c.y_len; a.wn_c = c.wn_c;
  a.nsteps = c.Cin / 4;  // = padded_steps(Cin, 1, 16): Cin % 32 == 0 leaves nothing to pad
  a.x_row_bytes = c.Lin * 4;
  // quad q, lane group kk reads physical channel in_base + sign·(4q + kk): for a reversed map that is (in_base − 4q − 3) + (3 − kk)
  a.kk_sign = c.in_ch_sign;
  a.x_base_bytes = (c.in_ch_sign > 0 ? c.in_ch_base : c.in_ch_base - 3) * c.Lin * 4;
  a.q_stride = c.in_ch_sign * 16 * c.Lin;
  a.out_ch_base = c.out_ch_base; a.out_ch_sign = c.out_ch_sign;
  a.x_batch_bytes = (int)(c.x_batch_stride * 4);
  a.x_bs = c.x_batch_stride; a.y_bs = c.y_batch_stride; a.y2_bs = c.y2_batch_stride;
  const dim3 grid((unsigned)mtiles, (unsigned)chunk_groups(ctx, mtiles, nchunks, c.N), (unsigned)c.N);
  const bool ok = NQ == 3 ? launch_lean_nq<3>(s, grid, a, c.epilogue, nchunks) : launch_lean_nq<6>(s, grid, a, c.epilogue, nchunks);
  if (!ok) return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_lean launch failed: %s", hipGetErrorString(e));
  return 1;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(conv_lean, (conv_k1_kernel<6, 0>)); } }
