// conv_kernels.hpp — the MFMA Conv1d kernels (streaming + LDS-tiled) and their launch templates.
// Included by conv.hip (host dispatch; sees only `extern template` declarations) and by one small translation unit per
// tap count (conv_inst_k*.hip) that instantiates the kernels — so a clean build compiles them in parallel.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "conv.h"

namespace ph {
namespace detail {


typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kBlock = 256;

__device__ __forceinline__ float lrelu(float v, float a) { return v >= 0.0f ? v : a * v; }

__device__ __forceinline__ float sigmoid_stable(float x) {  // elementwise.metal:253-268
  if (x >= 0.0f) {
    const float z = expf(-x);
    return 1.0f / (1.0f + z);
  }
  const float z = expf(x);
  return z / (1.0f + z);
}

// accumulator register r of lane → tile row (guide §3 "Fragment layout")
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int TM> struct AccSel { using type = f32x16; };
template <> struct AccSel<16> { using type = f32x4; };
// 16x16x4: C/D col = lane & 15, row = 4·(lane >> 4) + reg; A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]
template <int TM>
__device__ __forceinline__ int acc_row_t(int r, int lane) {
  if constexpr (TM == 32) return acc_row(r, lane);
  else return 4 * (lane >> 4) + r;
}
template <int TM>
__device__ __forceinline__ typename AccSel<TM>::type mfma_t(float a, float b, typename AccSel<TM>::type c) {
  if constexpr (TM == 32) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int true_len(const ConvArgs& p, int n) { return p.len_ptr ? min(p.len_ptr[n] * p.len_mul, p.Lin) : p.Lin; }

template <int PRO>
__device__ __forceinline__ float load_b(const ConvArgs& p, const float* xrow, const float* x2row, const float* x3row, int pos,
                                        bool ch_ok, int Lv) {
  if (!ch_ok || pos < 0 || pos >= Lv) return 0.0f;
  float v = xrow[pos];
  if constexpr (PRO == PRO_AVG3_LRELU) v = ((v + x2row[pos]) + x3row[pos]) / 3.0f;
  if constexpr (PRO != PRO_NONE) v = lrelu(v, p.alpha);
  return v;
}

// Epilogue of one output element, specialised per mode so that the (wave-uniform) mode switch happens once per tile, not
// once per element.  Indices are 32-bit element offsets from the batch item's base (host checks they fit).
template <int MODE>
__device__ __forceinline__ void store_mode(const ConvArgs& p, int n, int row, int col, float v) {
  if constexpr (MODE == EPI_STORE || MODE == EPI_RELU || MODE == EPI_TANH || MODE == EPI_RSUB) {
    const int idx = (p.out_ch_base + p.out_ch_sign * row) * p.y_len + col;
    float* yb = p.y + (int64_t)n * p.y_batch_stride;
    if constexpr (MODE == EPI_RELU) v = v > 0.0f ? v : 0.0f;
    else if constexpr (MODE == EPI_TANH) v = tanhf(v);
    else if constexpr (MODE == EPI_RSUB) v = (p.res + (int64_t)n * p.y_batch_stride)[idx] - v;
    else if (p.res) v = v + (p.res + (int64_t)n * p.y_batch_stride)[idx];
    yb[idx] = v;
  } else if constexpr (MODE == EPI_WN_RES_SKIP) {
    if (row < p.wn_c) {
      const int idx = row * p.y_len + col;
      (p.y + (int64_t)n * p.y_batch_stride)[idx] = (p.res + (int64_t)n * p.y_batch_stride)[idx] + v;
    } else {
      const int idx = (row - p.wn_c) * p.y_len + col;
      const float sk = p.skip ? (p.skip + (int64_t)n * p.y2_batch_stride)[idx] : 0.0f;
      (p.y2 + (int64_t)n * p.y2_batch_stride)[idx] = sk + v;
    }
  } else if constexpr (MODE == EPI_WN_SKIP_LAST) {
    const int idx = row * p.y_len + col;
    const float sk = p.skip ? (p.skip + (int64_t)n * p.y2_batch_stride)[idx] : 0.0f;
    (p.y2 + (int64_t)n * p.y2_batch_stride)[idx] = sk + v;
  } else if constexpr (MODE == EPI_MRF_MEAN) {
    const int idx = row * p.y_len + col;
    const int64_t bo = (int64_t)n * p.y_batch_stride;
    const float r2 = v + (p.res + bo)[idx];
    const float m = (((p.mrf_a + bo)[idx] + (p.mrf_b + bo)[idx]) + r2) / 3.0f;
    (p.y + bo)[idx] = lrelu(m, p.alpha2);
  } else if constexpr (MODE == EPI_CONVT) {
    int co, ph;
    if (p.ct_shift >= 0) {  // stride is a power of two in every Piper voice (8, 8, 4 / 8, 8, 2, 2)
      co = row >> p.ct_shift;
      ph = row & (p.ct_stride - 1);
    } else {
      co = row / p.ct_stride;
      ph = row - co * p.ct_stride;
    }
    const int xo = col * p.ct_stride + ph - p.ct_padL;
    if (xo >= 0 && xo < p.ct_Lout) (p.y + (int64_t)n * p.y_batch_stride)[co * p.y_len + xo] = v;
  }
}

__device__ __forceinline__ void store_elem(const ConvArgs& p, int n, int row, int col, float v) {
  switch (p.epilogue) {
    case EPI_STORE: store_mode<EPI_STORE>(p, n, row, col, v); break;
    case EPI_RELU: store_mode<EPI_RELU>(p, n, row, col, v); break;
    case EPI_TANH: store_mode<EPI_TANH>(p, n, row, col, v); break;
    case EPI_RSUB: store_mode<EPI_RSUB>(p, n, row, col, v); break;
    case EPI_WN_RES_SKIP: store_mode<EPI_WN_RES_SKIP>(p, n, row, col, v); break;
    case EPI_WN_SKIP_LAST: store_mode<EPI_WN_SKIP_LAST>(p, n, row, col, v); break;
    case EPI_CONVT: store_mode<EPI_CONVT>(p, n, row, col, v); break;
    case EPI_MRF_MEAN: store_mode<EPI_MRF_MEAN>(p, n, row, col, v); break;
  }
}

// Two-phase epilogue of the MFMA kernels. A wave's tile has NR·NT elements; with the loads of store_mode sitting inside
// per-element bounds tests, every element was its own load → wait → store round trip (≈ 0.5 µs each, 16-64 per wave).
// epi_load fetches what the mode ADDS to the accumulator from clamped, always-valid indices (so all of a tile's loads
// are in flight together, unconditionally); epi_finish does the arithmetic and the (masked) store. The arithmetic and
// its order are exactly store_mode's.
struct EpiIn {
  float a, b, c;
};
template <int MODE>
__device__ __forceinline__ EpiIn epi_load(const ConvArgs& p, int n, int row, int col) {
  EpiIn e{0.0f, 0.0f, 0.0f};
  if constexpr (MODE == EPI_STORE || MODE == EPI_RSUB) {
    const int idx = (p.out_ch_base + p.out_ch_sign * row) * p.y_len + col;
    if (MODE == EPI_RSUB || p.res) e.a = (p.res + (int64_t)n * p.y_batch_stride)[idx];
  } else if constexpr (MODE == EPI_WN_RES_SKIP) {
    if (row < p.wn_c) {
      e.a = (p.res + (int64_t)n * p.y_batch_stride)[row * p.y_len + col];
    } else if (p.skip) {
      e.a = (p.skip + (int64_t)n * p.y2_batch_stride)[(row - p.wn_c) * p.y_len + col];
    }
  } else if constexpr (MODE == EPI_WN_SKIP_LAST) {
    if (p.skip) e.a = (p.skip + (int64_t)n * p.y2_batch_stride)[row * p.y_len + col];
  } else if constexpr (MODE == EPI_MRF_MEAN) {
    const int idx = row * p.y_len + col;
    const int64_t bo = (int64_t)n * p.y_batch_stride;
    e.a = (p.res + bo)[idx];
    e.b = (p.mrf_a + bo)[idx];
    e.c = (p.mrf_b + bo)[idx];
  }
  return e;
}
template <int MODE>
__device__ __forceinline__ void epi_finish(const ConvArgs& p, int n, int row, int col, float v, const EpiIn& e) {
  if constexpr (MODE == EPI_STORE) {
    const int idx = (p.out_ch_base + p.out_ch_sign * row) * p.y_len + col;
    (p.y + (int64_t)n * p.y_batch_stride)[idx] = p.res ? v + e.a : v;
  } else if constexpr (MODE == EPI_RSUB) {
    const int idx = (p.out_ch_base + p.out_ch_sign * row) * p.y_len + col;
    (p.y + (int64_t)n * p.y_batch_stride)[idx] = e.a - v;
  } else if constexpr (MODE == EPI_WN_RES_SKIP) {
    if (row < p.wn_c) (p.y + (int64_t)n * p.y_batch_stride)[row * p.y_len + col] = e.a + v;
    else (p.y2 + (int64_t)n * p.y2_batch_stride)[(row - p.wn_c) * p.y_len + col] = e.a + v;
  } else if constexpr (MODE == EPI_WN_SKIP_LAST) {
    (p.y2 + (int64_t)n * p.y2_batch_stride)[row * p.y_len + col] = e.a + v;
  } else if constexpr (MODE == EPI_MRF_MEAN) {
    const float r2 = v + e.a;
    const float m = ((e.b + e.c) + r2) / 3.0f;
    (p.y + (int64_t)n * p.y_batch_stride)[row * p.y_len + col] = lrelu(m, p.alpha2);
  } else {
    store_mode<MODE>(p, n, row, col, v);  // RELU / TANH / CONVT add nothing from memory
  }
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// bias index of a GEMM row: the row itself, or row / stride for ConvTranspose's (co, phase) rows — without paying an
// integer division per accumulator register in every conv's prologue (16 per lane ≈ 1 µs of VALU work)
__device__ __forceinline__ int bias_index(const ConvArgs& p, int row) {
  if (p.ct_stride <= 0) return row;
  return p.ct_shift >= 0 ? (row >> p.ct_shift) : (row / p.ct_stride);
}

// Contraction steps fetched per prefetch group = G(K) channel units × K taps. Small groups on purpose (2–7 steps): the in-block
// K-split keeps ≥ 2 groups per slice, so the group size caps how finely a short-utterance conv can be spread over waves and
// blocks — and that, not the depth of the prefetch, is what these launches are bound by (DESIGN.md finding 10). r2: G(1) 8 → 4 → 2
// took the factor-8 utterance 0.905 → 0.877 → 0.870 ms (factor 1: 0.703 → 0.676 → 0.650), G(1) = 1 gave some of it back.
template <int K>
struct GroupOf {
  static constexpr int G = K == 1 ? 2 : K == 2 ? 4 : K == 3 ? 2 : K == 5 ? 1 : 1;
};

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// conv_stream_kernel<K taps, NT time tiles per wave, GATE, PRO, BT threads per block, TM tile edge>
//
// TM = 32: v_mfma_f32_32x32x2_f32 (2 channels per step); TM = 16: v_mfma_f32_16x16x4_f32 (4 channels per step).  The
// 16-wide geometry exists for short utterances: a [192 × 112] output is 24 tiles of 32² but 84 of 16², and tiles ×
// split-K slices is all the parallelism such a conv has.
//
// Everything that is the same for the 64 lanes of a wave — tile coordinates, contraction cursor, row bases, tap
// offsets — is kept in SGPRs (the wave id goes through readfirstlane, otherwise hipcc treats all of it as per-lane
// 64-bit VALU arithmetic and the kernel becomes VALU-issue-bound at ~20 instructions per MFMA).  Operands come in
// through buffer descriptors: lane part in a constant 32-bit voffset, hardware range checking makes every address
// legal (out of range ⇒ 0), so the fetch is branch-free: per B element one v_add + one buffer_load, per A fragment one
// buffer_load with an SGPR offset.  Zero padding inside a row is a per-lane bit mask computed once per tile and only
// on tiles that touch a row edge.  Groups of G·K steps are double-buffered in registers (loads of group g+1 are in
// flight while group g feeds the matrix pipe).
#ifdef PH_STREAM_TRACE
#define PH_SSTAMP(k) do { if (p.trace && lane == 0 && blockIdx.x < 512 && blockIdx.y == 0) p.trace[((size_t)blockIdx.x * 16 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_SSTAMP(k) do { } while (0)
#endif

template <int K, int NT, bool GATE, int PRO, int BT, int TM>
__global__ __launch_bounds__(BT) void conv_stream_kernel(const ConvArgs p, const int nchunks, const int mtiles, const int ks_log2,
                                                         const int ngroups) {
  constexpr int G = GroupOf<K>::G, S = G * K;
  constexpr int CPS = TM == 32 ? 2 : 4;   // input channels per contraction step
  constexpr int NR = TM == 32 ? 16 : 4;   // accumulator registers per tile
  using AccT = typename AccSel<TM>::type;
  constexpr int NA = GATE ? 2 : 1;
  constexpr int NX = (PRO == PRO_AVG3_LRELU) ? 3 : 1;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [KS-1][WT][NA][NT][NR][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  PH_SSTAMP(0);
  const int KS = 1 << ks_log2;
  const int WT = (BT / 64) >> ks_log2;
  const int tw = wave >> ks_log2, ks = wave & (KS - 1);
  const int n = blockIdx.y;
  const int mt_eff = GATE ? mtiles / 2 : mtiles;
  const int tile = (int)blockIdx.x * WT + tw;
  const bool in_grid = tile < mt_eff * nchunks;
  const int mt = in_grid ? tile % mt_eff : 0;
  const int chunk = in_grid ? tile / mt_eff : 0;
  const int t0 = chunk * TM * NT;
  // Bucketed / ragged batches: a tile whose every column lies at or past the item's TRUE length computes nothing anyone reads
  // (consumers mask by the same length) — 'same'-length convs and ConvTranspose (GEMM columns = input positions) only
  const bool past_len = p.len_ptr && (p.Lout == p.Lin || p.epilogue == EPI_CONVT) && t0 >= true_len(p, n);
  const bool active = in_grid && !past_len;
  const int j = lane & (TM - 1), kk = lane / TM;
  const int ncp = (p.Cin + CPS - 1) / CPS;  // channel units (pairs / quads)
  const int nsteps = ngroups * S;  // packed steps per row tile (zero-padded to whole groups)

  AccT acc[NA][NT];
#pragma unroll
  for (int a = 0; a < NA; a++) {
    const int mbase = (a == 0 ? mt : mt + mt_eff) * TM;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int row = mbase + acc_row_t<TM>(r, lane);
      const float b = (ks == 0 && p.bias && row < p.Cout) ? p.bias[bias_index(p, row)] : 0.0f;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[a][nt][r] = b;
    }
  }

  PH_SSTAMP(1);
  if (active) {
    const int g_begin = (int)(((int64_t)ngroups * ks) >> ks_log2), g_end = (int)(((int64_t)ngroups * (ks + 1)) >> ks_log2);
    const int xbytes = (int)(p.x_batch_stride * 4);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.x_batch_stride), 0, xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rx2 = rx, rx3 = rx;
    if constexpr (NX == 3) {
      rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 + (int64_t)n * p.x_batch_stride), 0, xbytes, 0x00020000);
      rx3 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x3 + (int64_t)n * p.x_batch_stride), 0, xbytes, 0x00020000);
    }
    const int wbytes = nsteps * 256;
    const __amdgpu_buffer_rsrc_t rwa = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)mt * nsteps * 64), 0, wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rwb = rwa;
    if constexpr (GATE) rwb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)(mt + mt_eff) * nsteps * 64), 0, wbytes, 0x00020000);
    // lane parts of the addresses (bytes). Rows of a channel pair: {r0, r0+1}; with a reversed channel map the pair is
    // stored in descending order, so lane half kk picks row (1−kk) and r0 is lowered by one.
    const int rowsel = p.in_ch_sign > 0 ? kk : CPS - 1 - kk;
    const int voffB = (rowsel * p.Lin + j) * 4;
    const int voffA = lane * 4;
    const int row_adj = p.in_ch_sign > 0 ? 0 : -(CPS - 1);
    const int tb = t0 - p.padL;
    // does any element of this tile's window fall outside [0, Lin)?  (wave-uniform)
    const int span_lo = tb + (p.dil < 0 ? (K - 1) * p.dil : 0), span_hi = tb + (p.dil > 0 ? (K - 1) * p.dil : 0) + TM * NT - 1;
    const int Lv = true_len(p, n);  // wave-uniform
    const bool interior = span_lo >= 0 && span_hi < Lv;

    auto body = [&](auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      // bit (k·NT + nt) of okbits: this lane's element of tap k / time tile nt lies inside the row
      unsigned long long okbits = ~0ull;
      if constexpr (EDGE) {
        okbits = 0;
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            const int pos = tb + k * p.dil + TM * nt + j;
            if (pos >= 0 && pos < Lv) okbits |= 1ull << (k * NT + nt);
          }
      }
      // D groups form a register ring: D−1 groups of loads are in flight while one group feeds the matrix pipe.  With
      // ~1–2 waves per SIMD (all a short utterance offers) this is what covers the 0.3–2 µs load latency.
      // PRO_LN: statistics of this lane's columns (one per tap and time tile), from the producer's per-slot partial sums
      float lnm[PRO == PRO_LN ? K : 1][NT], lns[PRO == PRO_LN ? K : 1][NT];
      auto load_ln_stats = [&]() {
        // The lanes that share a column (64 / TM of them: lane groups kk) split the producer's slots among themselves, every
        // load is independent (ONE memory round trip; a serial loop over the 12 slots cost 12 of them: r2g, +7 µs per conv),
        // then the group sums are exchanged by lane shuffles. Fixed association ⇒ deterministic.
        const int parts = (p.Cin + 15) >> 4;
        constexpr int NG = 64 / TM;          // lane groups per column
        constexpr int PPG = TM == 16 ? 4 : 8;  // slots per lane group: covers parts ≤ 16 (Cin ≤ 256; host-checked)
        const float* sb = p.ln_stats + (int64_t)n * parts * p.Lin * 2;
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            const int col = min(max(tb + k * p.dil + TM * nt + j, 0), p.Lin - 1);
            float2 pr[PPG];
#pragma unroll
            for (int q = 0; q < PPG; q++) {
              const int slot = kk * PPG + q;
              pr[q] = *(const float2*)(sb + ((int64_t)min(slot, parts - 1) * p.Lin + col) * 2);
              if (slot >= parts) pr[q] = make_float2(0.0f, 0.0f);
            }
            // slot i holds (Σ y, Σ (y − mean_i)²) of its n_i rows: Chan's combination — mean = ΣΣ / C, M2 = Σ [M2_i + n_i·(mean_i − mean)²] —
            // instead of Σ y² / C − mean², which loses (mean / sigma)² · 6e-8 of relative accuracy
            float s1 = 0.0f;
#pragma unroll
            for (int q = 0; q < PPG; q++) s1 += pr[q].x;
            s1 += __shfl_xor(s1, 32, 64);
            if constexpr (NG == 4) s1 += __shfl_xor(s1, 16, 64);
            const float mean = s1 / (float)p.Cin;
            float m2 = 0.0f;
#pragma unroll
            for (int q = 0; q < PPG; q++) {
              const int slot = kk * PPG + q;
              const float ni = (float)min(16, p.Cin - 16 * slot);
              if (slot < parts) {
                const float dm = pr[q].x / ni - mean;
                m2 += pr[q].y + ni * (dm * dm);
              }
            }
            m2 += __shfl_xor(m2, 32, 64);
            if constexpr (NG == 4) m2 += __shfl_xor(m2, 16, 64);
            const float var = m2 / (float)p.Cin;
            lnm[k][nt] = mean;
            lns[k][nt] = 1.0f / sqrtf(var + p.ln_eps);  // reciprocal once per column: a division per operand element in the K loop is ≈ 10 vector instructions next to every MFMA
          }
      };
      constexpr int regs_per_group = S * (NA + NX * NT) + (PRO == PRO_LN ? 2 * G : 0);
      // D = 2 above 256 threads. (A deeper ring there — so that a wave's 3–4 groups are all in flight before its first MFMA — was
      // measured: the flow's gated conv stayed at 13.4 µs, the FFN's second conv went 12.9 → 16.7 µs. These launches are bound by the
      // bytes their CU pulls in, not by dependent round trips. Re-measured with the small groups of round 2 (ring of 3–4 wherever ≤ 64–96
      // registers allow): factor 8 0.849 → 0.884 ms, factor 1 0.648 → 0.675. Capping the ring at 2 everywhere costs 8 % at factor 64, at 3 nothing.)
      constexpr int D = BT > 256 ? 2 : (regs_per_group * 4 <= 112 ? 4 : (regs_per_group * 3 <= 132 ? 3 : 2));
      float av[D][NA][S], bv[D][NX][S][NT];
      float lng[PRO == PRO_LN ? D : 1][G], lnb[PRO == PRO_LN ? D : 1][G];  // gamma / beta of this lane's channel per channel unit
      auto fetch = [&](auto slot_tag, const int g) {
        constexpr int sl = decltype(slot_tag)::value;
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
          const int cp = g * G + gi;
          const int cpc = cp < ncp ? cp : ncp - 1;  // padded steps carry zero weights; keep their rows legal
          const int rowbase = (p.in_ch_base + p.in_ch_sign * CPS * cpc + row_adj) * p.Lin + tb;
          if constexpr (PRO == PRO_LN) {
            const int ch = min(CPS * cpc + kk, p.Cin - 1);
            lng[sl][gi] = p.ln_gamma[ch];
            lnb[sl][gi] = p.ln_beta[ch];
          }
#pragma unroll
          for (int k = 0; k < K; k++) {
            const int st = gi * K + k;
            const int soffA = (g * S + st) * 256;
            av[sl][0][st] = bload(rwa, voffA, soffA);
            if constexpr (GATE) av[sl][1][st] = bload(rwb, voffA, soffA);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
              const int off = voffB + (rowbase + k * p.dil + TM * nt) * 4;
              bv[sl][0][st][nt] = bload(rx, off, 0);
              if constexpr (NX == 3) {
                bv[sl][1][st][nt] = bload(rx2, off, 0);
                bv[sl][2][st][nt] = bload(rx3, off, 0);
              }
            }
          }
        }
      };
      auto compute = [&](auto slot_tag, const int g) {
        constexpr int sl = decltype(slot_tag)::value;
#pragma unroll
        for (int st = 0; st < S; st++) {
          const int k = st % K;
#pragma unroll
          for (int nt = 0; nt < NT; nt++) {
            float v = bv[sl][0][st][nt];
            if constexpr (NX == 3) v = ((v + bv[sl][1][st][nt]) + bv[sl][2][st][nt]) / 3.0f;
            if constexpr (PRO == PRO_LN) {
              v = ((v - lnm[k][nt]) * lns[k][nt]) * lng[sl][st / K] + lnb[sl][st / K];
              // the row-tile-0 waves materialise the normalised tensor once (centre tap = the column itself)
              if (mt == 0 && p.ln_out && k * p.dil == p.padL) {
                const int ch = CPS * (g * G + st / K) + kk, col = t0 + TM * nt + j;
                if (ch < p.Cin && col < p.Lin) (p.ln_out + (int64_t)n * p.x_batch_stride)[ch * p.Lin + col] = v;
              }
            } else if constexpr (PRO != PRO_NONE) v = lrelu(v, p.alpha);
            if constexpr (EDGE) v = ((okbits >> (k * NT + nt)) & 1ull) ? v : 0.0f;
            acc[0][nt] = mfma_t<TM>(av[sl][0][st], v, acc[0][nt]);
            if constexpr (GATE) acc[1][nt] = mfma_t<TM>(av[sl][1][st], v, acc[1][nt]);
          }
        }
      };
      // The fetches are UNCONDITIONAL (past the slice end they re-fetch its last group, results unused): with a fetch
      // under an `if`, hipcc has to pick the s_waitcnt vmcnt(N) that is safe on the path where the fetch did not happen,
      // i.e. N ≈ 6 instead of ≈ 48 — which drains the whole ring before every group and serialises load and MFMA.
      if (g_begin < g_end) {
        const int g_last = g_end - 1;
        static_for<D - 1>([&](auto d) { fetch(d, min(g_begin + d.value, g_last)); });
        PH_SSTAMP(5);
        // the statistics are requested BEHIND the first operand groups: one memory round trip covers both
        if constexpr (PRO == PRO_LN) load_ln_stats();
        for (int g = g_begin; g < g_end; g += D) {
          static_for<D>([&](auto d) {
            const int gg = g + d.value;
            fetch(std::integral_constant<int, (d.value + D - 1) % D>{}, min(gg + D - 1, g_last));
            if (gg < g_end) compute(d, gg);
            if (d.value == 0 && g == g_begin) PH_SSTAMP(6);
          });
        }
      }
    };
    if (interior) body(std::false_type{});
    else body(std::true_type{});
  }

  PH_SSTAMP(2);
  if (KS > 1) {  // fixed-order reduction over the contraction slices: slice 0 + slice 1 + … (deterministic)
    constexpr int per_wave = NA * NT * NR * 64;
    if (ks > 0) {
      float* dst = red + (int64_t)((ks - 1) * WT + tw) * per_wave + lane;
#pragma unroll
      for (int a = 0; a < NA; a++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < NR; r++) dst[((a * NT + nt) * NR + r) * 64] = acc[a][nt][r];
    }
    __syncthreads();
    if (ks == 0) {
      for (int s2 = 1; s2 < KS; s2++) {
        const float* src = red + (int64_t)((s2 - 1) * WT + tw) * per_wave + lane;
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < NR; r++) acc[a][nt][r] += src[((a * NT + nt) * NR + r) * 64];
      }
    }
  }

  PH_SSTAMP(3);
  if (!active || ks != 0) return;
  const int rows_out = GATE ? p.Cout / 2 : p.Cout;
  auto emit = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int col = t0 + TM * nt + j;
      const int colc = min(col, p.Lout - 1);
      EpiIn e[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) e[r] = epi_load<MODE>(p, n, min(mt * TM + acc_row_t<TM>(r, lane), rows_out - 1), colc);
      float vals[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int row = mt * TM + acc_row_t<TM>(r, lane);
        float v = acc[0][nt][r];
        if constexpr (GATE) v = tanhf(v) * sigmoid_stable(acc[1][nt][r]);
        vals[r] = 0.0f;
        if (col < p.Lout && row < rows_out) {
          epi_finish<MODE>(p, n, row, col, v, e[r]);
          if constexpr (MODE == EPI_STORE) vals[r] = p.res ? v + e[r].a : v;  // exactly what epi_finish stored
        }
      }
      if constexpr (MODE == EPI_STORE && !GATE) {
        if (p.stats_out) {  // wave-uniform: LayerNorm statistics of this tile's rows per column and 16-row slot (see ConvArgs)
          const int parts = (p.Cout + 15) >> 4;
          float* sb = p.stats_out + (int64_t)n * parts * p.y_len * 2;
          constexpr int NSL = TM / 16;  // 16-row slots of the tile: 32-wide tiles hold two (registers 0–7 / 8–15 of a lane)
#pragma unroll
          for (int sl = 0; sl < NSL; sl++) {
            const int slot = NSL * mt + sl;
            const float cnt = (float)max(1, min(16, rows_out - 16 * slot));
            float s1 = 0.0f;
#pragma unroll
            for (int r = 0; r < NR / NSL; r++) s1 += vals[sl * (NR / NSL) + r];
            s1 += __shfl_xor(s1, 32, 64);
            if constexpr (TM == 16) s1 += __shfl_xor(s1, 16, 64);
            const float ms = s1 / cnt;
            float q2 = 0.0f;
#pragma unroll
            for (int r = 0; r < NR / NSL; r++) {
              const int row = mt * TM + acc_row_t<TM>(sl * (NR / NSL) + r, lane);
              const float dv = vals[sl * (NR / NSL) + r] - ms;
              if (row < rows_out) q2 += dv * dv;
            }
            q2 += __shfl_xor(q2, 32, 64);
            if constexpr (TM == 16) q2 += __shfl_xor(q2, 16, 64);
            if (lane < TM && col < p.Lout && slot < parts) *(float2*)(sb + ((int64_t)slot * p.y_len + col) * 2) = make_float2(s1, q2);
          }
        }
      }
    }
  };
  switch (p.epilogue) {  // wave-uniform
    case EPI_STORE: emit(std::integral_constant<int, EPI_STORE>{}); break;
    case EPI_RELU: emit(std::integral_constant<int, EPI_RELU>{}); break;
    case EPI_TANH: emit(std::integral_constant<int, EPI_TANH>{}); break;
    case EPI_RSUB: emit(std::integral_constant<int, EPI_RSUB>{}); break;
    case EPI_WN_RES_SKIP: emit(std::integral_constant<int, EPI_WN_RES_SKIP>{}); break;
    case EPI_WN_SKIP_LAST: emit(std::integral_constant<int, EPI_WN_SKIP_LAST>{}); break;
    case EPI_CONVT: emit(std::integral_constant<int, EPI_CONVT>{}); break;
    case EPI_MRF_MEAN: emit(std::integral_constant<int, EPI_MRF_MEAN>{}); break;
  }
  PH_SSTAMP(4);
}

// ======================================================================================================
// conv_tile_kernel — the bulk path (long rows: HiFi-GAN stages, long-form flow).
//
// PMC on the streaming kernel at L = 86 016 showed the matrix pipe ≈ 35 % busy with waves parked on VMEM: every MFMA was
// pulling a fresh 512-byte B fragment through L1/L2 (K-fold re-reads of the activation row, ≈ 8 TB/s of cache traffic).
// Here a 256-thread block owns BM = 32·MT output channels × BN = 128·NTW time steps; per chunk of CPC channel pairs it
// stages the activation window [2·CPC][BN + halo] (pre-activation and zero padding applied on the way in) and the
// matching weight fragments into LDS ONCE, and all K taps × MT row tiles × NTW time tiles are fed from LDS
// (ds_read_b32, conflict-free: a wave reads 2 × 32 consecutive floats).  The next chunk's global loads are issued
// before the current chunk's MFMAs and parked in registers, so HBM/L2 latency sits under the matrix work.
// Global traffic per MFMA drops from ≈ 770 B to ≈ 80 B.
template <int K, int MT, int NTW>
struct TileCfg {
  static constexpr int CPC = (128 / (MT * K)) >= 8 ? 8 : ((128 / (MT * K)) >= 4 ? 4 : ((128 / (MT * K)) >= 2 ? 2 : 1));
  static constexpr int BN = 128 * NTW;
  static constexpr int A_FLOATS = MT * CPC * K * 64;
  static constexpr int A_PER_THREAD = A_FLOATS / 256;  // multiples of 256 by construction (64·MT·CPC·K)
};

template <int K, int MT, int NTW, int PRO>
__global__ __launch_bounds__(256) void conv_tile_kernel(const ConvArgs p, const int nchunks_t, const int mgroups, const int nsteps,
                                                        const int ldx, const int xs_floats) {
  using Cfg = TileCfg<K, MT, NTW>;
  constexpr int CPC = Cfg::CPC, BN = Cfg::BN;
  constexpr int NX = (PRO == PRO_AVG3_LRELU) ? 3 : 1;
  constexpr int XMAX = 24;  // staged activation elements per thread per chunk (host guarantees 2·CPC·W ≤ 256·XMAX)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;               // [2·CPC][ldx]
  float* As = lds + xs_floats;   // [MT][CPC·K][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 31, kk = lane >> 5;
  const int chunk_t = blockIdx.x % nchunks_t, mg = blockIdx.x / nchunks_t;
  const int n = blockIdx.y;
  const int t0 = chunk_t * BN;
  const int mt0 = mg * MT;
  const int ncp = (p.Cin + 1) >> 1;
  const int halo_lo = p.dil < 0 ? (K - 1) * p.dil : 0;  // ≤ 0
  const int W = BN + (K - 1) * (p.dil < 0 ? -p.dil : p.dil);
  const int lo = t0 - p.padL + halo_lo;  // input position of window column 0
  const float* xb = p.x + (int64_t)n * p.x_batch_stride;
  const float* x2b = NX == 3 ? p.x2 + (int64_t)n * p.x_batch_stride : nullptr;
  const float* x3b = NX == 3 ? p.x3 + (int64_t)n * p.x_batch_stride : nullptr;
  const int Lv = true_len(p, n);

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (mt0 + m) * 32 + acc_row(r, lane);
      const float b = (p.bias && row < p.Cout) ? p.bias[bias_index(p, row)] : 0.0f;
#pragma unroll
      for (int nt = 0; nt < NTW; nt++) acc[m][nt][r] = b;
    }

  // staging registers.  The window is cut into 64-column segments; segment s = wave + 4·i of the [2·CPC][nseg] grid is
  // slot i of this wave, so (row, segment) are wave-uniform scalars and the only per-lane address part is lane·4
  // (no per-element integer division: the first version of this kernel spent more VALU time on e / W than on MFMAs).
  float xr[NX][XMAX];
  float ar[Cfg::A_PER_THREAD];
  const int nseg = (W + 63) >> 6;
  constexpr int nrows = 2 * CPC;
  const int xbytes = (int)(p.x_batch_stride * 4);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, xbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rx2 = rx, rx3 = rx;
  if constexpr (NX == 3) {
    rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)x2b, 0, xbytes, 0x00020000);
    rx3 = __builtin_amdgcn_make_buffer_rsrc((void*)x3b, 0, xbytes, 0x00020000);
  }
  const int lane4 = lane * 4;
  auto load_chunk = [&](const int c0 /*first channel pair*/) {
    int row = 0, seg = wave;
#pragma unroll
    for (int i = 0; i < XMAX; i++) {
      while (seg >= nseg) { seg -= nseg; row++; }
      if (row < nrows) {
        const int ch = 2 * c0 + row;
        const int chc = ch < p.Cin ? ch : p.Cin - 1;
        // element offset of (row, seg·64) in the tensor; negative / past-the-end offsets are range-checked to 0 by the
        // descriptor, positions that fall into a neighbouring row are masked when the value is written to LDS
        const int sbase = ((p.in_ch_base + p.in_ch_sign * chc) * p.Lin + lo + seg * 64) * 4;
        xr[0][i] = bload(rx, sbase + lane4, 0);
        if constexpr (NX == 3) {
          xr[1][i] = bload(rx2, sbase + lane4, 0);
          xr[2][i] = bload(rx3, sbase + lane4, 0);
        }
      }
      seg += 4;
    }
#pragma unroll
    for (int i = 0; i < Cfg::A_PER_THREAD; i++) {
      const int e = tid + 256 * i;                 // [m][step in chunk][64]
      const int m = e / (CPC * K * 64), r = e - m * (CPC * K * 64);
      const int st = c0 * K + (r >> 6);            // global step of this tile row
      const int mt = mt0 + m;
      const bool ok = st < nsteps && mt * 32 < p.Cout;
      ar[i] = ok ? p.w[((int64_t)mt * nsteps + st) * 64 + (r & 63)] : 0.0f;
    }
  };
  auto store_chunk = [&](const int c0) {
    int row = 0, seg = wave;
#pragma unroll
    for (int i = 0; i < XMAX; i++) {
      while (seg >= nseg) { seg -= nseg; row++; }
      if (row < nrows) {
        const int col = seg * 64 + lane;
        const int pos = lo + col;
        const bool ok = (2 * c0 + row) < p.Cin && pos >= 0 && pos < Lv;
        float v = xr[0][i];
        if constexpr (NX == 3) v = ((v + xr[1][i]) + xr[2][i]) / 3.0f;
        if constexpr (PRO != PRO_NONE) v = lrelu(v, p.alpha);
        if (col < W) Xs[row * ldx + col] = ok ? v : 0.0f;
      }
      seg += 4;
    }
#pragma unroll
    for (int i = 0; i < Cfg::A_PER_THREAD; i++) As[tid + 256 * i] = ar[i];
  };

  const int colw = wave * (32 * NTW) - halo_lo + j;  // this lane's window column for tap 0, time tile 0
  load_chunk(0);
  for (int c0 = 0; c0 < ncp; c0 += CPC) {
    __syncthreads();  // previous chunk fully consumed
    store_chunk(c0);
    __syncthreads();
    if (c0 + CPC < ncp) load_chunk(c0 + CPC);  // in flight during the MFMAs below
#pragma unroll
    for (int cp = 0; cp < CPC; cp++) {
      const float* xrow = Xs + (2 * cp + kk) * ldx + colw;
#pragma unroll
      for (int k = 0; k < K; k++) {
        float a[MT], b[NTW];
#pragma unroll
        for (int m = 0; m < MT; m++) a[m] = As[(m * CPC * K + cp * K + k) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NTW; nt++) b[nt] = xrow[k * p.dil + 32 * nt];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int nt = 0; nt < NTW; nt++) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[nt], acc[m][nt], 0, 0, 0);
      }
    }
  }

  auto emit = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NTW; nt++) {
      const int col = t0 + wave * (32 * NTW) + 32 * nt + j;
      const int colc = min(col, p.Lout - 1);
#pragma unroll
      for (int m = 0; m < MT; m++) {
        EpiIn e[16];
#pragma unroll
        for (int r = 0; r < 16; r++) e[r] = epi_load<MODE>(p, n, min((mt0 + m) * 32 + acc_row(r, lane), p.Cout - 1), colc);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = (mt0 + m) * 32 + acc_row(r, lane);
          if (col < p.Lout && row < p.Cout) epi_finish<MODE>(p, n, row, col, acc[m][nt][r], e[r]);
        }
      }
    }
  };
  switch (p.epilogue) {  // wave-uniform
    case EPI_STORE: emit(std::integral_constant<int, EPI_STORE>{}); break;
    case EPI_RELU: emit(std::integral_constant<int, EPI_RELU>{}); break;
    case EPI_TANH: emit(std::integral_constant<int, EPI_TANH>{}); break;
    case EPI_RSUB: emit(std::integral_constant<int, EPI_RSUB>{}); break;
    case EPI_WN_RES_SKIP: emit(std::integral_constant<int, EPI_WN_RES_SKIP>{}); break;
    case EPI_WN_SKIP_LAST: emit(std::integral_constant<int, EPI_WN_SKIP_LAST>{}); break;
    case EPI_CONVT: emit(std::integral_constant<int, EPI_CONVT>{}); break;
    case EPI_MRF_MEAN: emit(std::integral_constant<int, EPI_MRF_MEAN>{}); break;
  }
}

inline bool k_supported(int K) { return K == 1 || K == 2 || K == 3 || K == 5 || K == 7 || K == 11; }
inline int group_of(int K) { return K == 1 ? 2 : K == 2 ? 4 : K == 3 ? 2 : K == 5 ? 1 : 1; }
inline int padded_steps(int Cin, int K, int tm = 32) {
  const int cps = tm == 32 ? 2 : 4;
  const int ncu = (Cin + cps - 1) / cps, G = group_of(K);
  return (int)ceil_div(ncu, G) * G * K;
}

template <int K, int NT, bool GATE, int PRO, int BT, int TM>
void launch_one(hipStream_t s, const ConvArgs& a, int nchunks, int mtiles, int ks_log2, int ngroups, dim3 grid, size_t lds) {
  if (lds > 64 * 1024) {  // opt in to > 64 KiB of dynamic LDS once per instantiation
    static bool configured[kMaxDevices] = {};
    if (lds_optin_needed(configured))
      (void)hipFuncSetAttribute((const void*)conv_stream_kernel<K, NT, GATE, PRO, BT, TM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
  }
  hipLaunchKernelGGL((conv_stream_kernel<K, NT, GATE, PRO, BT, TM>), grid, dim3(BT), lds, s, a, nchunks, mtiles, ks_log2, ngroups);
}

template <int K, bool GATE, int PRO, int TM>
bool launch_shape(hipStream_t s, const ConvArgs& a, int NT, int BT, int nchunks, int mtiles, int ks_log2, int ngroups, dim3 grid,
                  size_t lds) {
  if (BT == 256) {
    if (NT == 1) { launch_one<K, 1, GATE, PRO, 256, TM>(s, a, nchunks, mtiles, ks_log2, ngroups, grid, lds); return true; }
    if constexpr (TM == 32) {
      if (NT == 2) { launch_one<K, 2, GATE, PRO, 256, TM>(s, a, nchunks, mtiles, ks_log2, ngroups, grid, lds); return true; }
      if constexpr (!GATE && PRO != PRO_AVG3_LRELU)
        if (NT == 4) { launch_one<K, 4, GATE, PRO, 256, TM>(s, a, nchunks, mtiles, ks_log2, ngroups, grid, lds); return true; }
    }
    return false;
  }
  if (NT != 1) return false;
  if (BT == 512) { launch_one<K, 1, GATE, PRO, 512, TM>(s, a, nchunks, mtiles, ks_log2, ngroups, grid, lds); return true; }
  if constexpr (!GATE || TM == 16)
    if (BT == 1024) { launch_one<K, 1, GATE, PRO, 1024, TM>(s, a, nchunks, mtiles, ks_log2, ngroups, grid, lds); return true; }
  return false;
}

// which (K, GATE, PRO, TM) combinations are compiled: every conv of the Piper graph + the op-level API's plain convs
template <int K, int TM>
bool launch_k(hipStream_t s, const ConvArgs& a, int NT, int BT, int nchunks, int mtiles, int ks_log2, int ngroups, dim3 grid,
              size_t lds) {
  if (a.gate) {
    if (a.prologue != PRO_NONE) return false;
    return launch_shape<K, true, PRO_NONE, TM>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds);
  }
  switch (a.prologue) {
    case PRO_NONE: return launch_shape<K, false, PRO_NONE, TM>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds);
    case PRO_LRELU: return launch_shape<K, false, PRO_LRELU, TM>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds);
    case PRO_AVG3_LRELU:
      if constexpr (K == 2 || K == 1) return launch_shape<K, false, PRO_AVG3_LRELU, TM>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds);
      return false;
    case PRO_LN:
      if constexpr (K == 1 || K == 3) return launch_shape<K, false, PRO_LN, TM>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds);
      return false;
  }
  return false;
}

template <int K, int MT, int NTW, int PRO>
bool launch_tile_one(hipStream_t s, const ConvArgs& a, int nsteps) {
  using Cfg = TileCfg<K, MT, NTW>;
  const int W = Cfg::BN + (K - 1) * (a.dil < 0 ? -a.dil : a.dil);
  if (2 * Cfg::CPC * ((W + 63) / 64) > 4 * 24) return false;  // staging slots per wave (XMAX)
  const int ldx = W + 1;
  const int xs_floats = ((2 * Cfg::CPC * ldx + 63) / 64) * 64;
  const size_t lds = (size_t)(xs_floats + Cfg::A_FLOATS) * sizeof(float);
  if (lds > 160 * 1024) return false;
  if (lds > 64 * 1024) {
    static bool configured[kMaxDevices] = {};
    if (lds_optin_needed(configured))
      (void)hipFuncSetAttribute((const void*)conv_tile_kernel<K, MT, NTW, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int mtiles = (int)ceil_div(a.Cout, 32);
  const int mgroups = (int)ceil_div(mtiles, MT);
  const int nchunks_t = (int)ceil_div(a.Lout, Cfg::BN);
  dim3 grid((unsigned)(nchunks_t * mgroups), (unsigned)a.N);
  hipLaunchKernelGGL((conv_tile_kernel<K, MT, NTW, PRO>), grid, dim3(256), lds, s, a, nchunks_t, mgroups, nsteps, ldx, xs_floats);
  return true;
}

template <int K, int PRO>
bool launch_tile_shape(hipStream_t s, const ConvArgs& a, int MT, int NTW, int nsteps) {
  if (MT == 1 && NTW == 2) return launch_tile_one<K, 1, 2, PRO>(s, a, nsteps);
  if (MT == 2 && NTW == 2) return launch_tile_one<K, 2, 2, PRO>(s, a, nsteps);
  if (MT == 4 && NTW == 1) return launch_tile_one<K, 4, 1, PRO>(s, a, nsteps);
  if (MT == 2 && NTW == 1) return launch_tile_one<K, 2, 1, PRO>(s, a, nsteps);
  if (MT == 1 && NTW == 1) return launch_tile_one<K, 1, 1, PRO>(s, a, nsteps);
  return false;
}

template <int K>
bool launch_tile_k(hipStream_t s, const ConvArgs& a, int MT, int NTW, int nsteps) {
  switch (a.prologue) {
    case PRO_NONE: return launch_tile_shape<K, PRO_NONE>(s, a, MT, NTW, nsteps);
    case PRO_LRELU: return launch_tile_shape<K, PRO_LRELU>(s, a, MT, NTW, nsteps);
    case PRO_AVG3_LRELU:
      if constexpr (K == 2 || K == 1) return launch_tile_shape<K, PRO_AVG3_LRELU>(s, a, MT, NTW, nsteps);
      return false;
  }
  return false;
}

// Bulk path eligibility + launch. Returns false when the streaming kernel should be used instead.
}  // namespace detail
}  // namespace ph
