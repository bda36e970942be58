// attention.hip — text-encoder relative-position self-attention core, one launch per layer.
//
//   scores[i][j] = (q_i/√d)·k_j + (q_i/√d)·E_k[j−i+w]   (second term only for |j−i| ≤ w)
//   p = softmax_j(scores);   out_i = Σ_j p[i][j]·v_j + Σ_{|δ|≤w} p[i][i+δ]·E_v[δ+w]
//
// This is exactly what the exported graph computes with MatMul + Pad/Reshape/Slice "skew" + Softmax
// (GraphExecutor.swift:1862-1929, 1180-1210, 1371-1426): rel→abs maps relative column m of row i to absolute
// column j = i + m − (T−1), abs→rel is its inverse, and the embedding table padded to 2T−1 rows is zero outside
// the ±w window — so the skew is pure index arithmetic and the [N,H,T,2T−1] tensors never exist.
// A block owns R query rows of one head; the R×T score strip lives in LDS (T ≤ 4096: the reference's own
// --max-phonemes cap, PiperCLI.swift:394).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "conv.h"

namespace {

constexpr int kBlock = 256;
constexpr int kKeyTile = 128;  // keys staged per LDS tile

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// K and V tiles are brought into LDS with one burst of independent, coalesced loads per tile (row stride TK+1 so that
// both the per-key and the per-channel access patterns are bank-conflict-free); the dot products then run out of LDS.
// With T ≈ 100 the op is a chain of memory round trips, so the structure minimises dependent global accesses:
// q strip → K tile(s) → V tile(s) → store.
template <int R>
__global__ __launch_bounds__(kBlock) void rel_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ ek,
                                                               const float* __restrict__ ev, float* __restrict__ out, int H, int d,
                                                               int T, int w, int64_t in_batch_stride, int64_t out_batch_stride,
                                                               int G, int TK
#ifdef PH_ATT_STAMPS
                                                               , unsigned long long* stamps
#endif
                                                               ) {
#ifdef PH_ATT_STAMPS
#define PH_STAMP(i) do { __syncthreads(); if (stamps && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PH_STAMP(i)
#endif
  PH_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int W = 2 * w + 1;
  const int ld = TK + 1;
  float* qs = smem;               // [R][d]
  float* qe = qs + R * d;         // [R][W]
  float* part = qe + R * W;       // [G][R][d]
  float* kv = part + G * R * d;   // [d][TK+1]
  float* sc = kv + d * ld;        // [R][T]
  float* eks = sc + R * T;        // [W][d] relative-key embeddings
  float* evs = eks + W * d;       // [W][d] relative-value embeddings
  float* psc = evs + W * d;       // [slices-1][R][TKP] partial scores of the channel slices (≤ 256·R floats)
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * R, h = blockIdx.y, n = blockIdx.z;
  const float* qb = q + (int64_t)n * in_batch_stride + (int64_t)h * d * T;
  const float* kb = k + (int64_t)n * in_batch_stride + (int64_t)h * d * T;
  const float* vb = v + (int64_t)n * in_batch_stride + (int64_t)h * d * T;
  const float scale = sqrtf((float)d);

  // 0. the two small embedding tables go to LDS with the q strip (one memory round trip for all three)
  for (int idx = tid; idx < W * d; idx += kBlock) {
    eks[idx] = ek[idx];
    evs[idx] = ev[idx];
  }
  PH_STAMP(1);
  // 1. q strip, scaled by Div like the graph (query / sqrt(k_channels))
  for (int idx = tid; idx < R * d; idx += kBlock) {
    const int c = idx / R, r = idx - c * R;
    const int i = i0 + r;
    qs[r * d + c] = i < T ? qb[(int64_t)c * T + i] / scale : 0.0f;
  }
  __syncthreads();
  PH_STAMP(2);
  // 1b. relative-key logits for the ±w window
  for (int idx = tid; idx < R * W; idx += kBlock) {
    const int r = idx / W, m = idx - r * W;
    float s = 0.0f;
#pragma unroll 8
    for (int c = 0; c < d; c++) s += qs[r * d + c] * eks[m * d + c];
    qe[r * W + m] = s;
  }
  PH_STAMP(3);
  // 2. score strip, one key tile at a time
  for (int j0 = 0; j0 < T; j0 += TK) {
    const int tk = min(TK, T - j0);
    __syncthreads();
    // rows of the tile are dealt to the 256 threads in passes of (256 / TKP) rows, TKP = next power of two ≥ tk: index
    // math is shifts, and the 8-way unrolled body keeps 8 independent loads per thread in flight
    {
      const int sh = 32 - __clz(tk - 1 > 0 ? tk - 1 : 1);  // log2(TKP) for tk ≥ 2
      const int shc = tk <= 1 ? 0 : sh;
      const int jj = tid & ((1 << shc) - 1), rpp = kBlock >> shc, cbase = tid >> shc;
      if (jj < tk) {
#pragma unroll 8
        for (int c = cbase; c < d; c += rpp) kv[c * ld + jj] = kb[(int64_t)c * T + j0 + jj];
      }
    }
    __syncthreads();
    // keys along the low bits of the thread id, channel slices along the high bits (256 / TKP slices): all 256 threads
    // work and the dependent FMA chain per thread is d / slices long instead of d
    {
      const int sh = tk <= 1 ? 0 : 32 - __clz(tk - 1);
      const int jj = tid & ((1 << sh) - 1), cs = tid >> sh, ns = kBlock >> sh;
      const int cb = (int)((int64_t)d * cs / ns), ce = (int)((int64_t)d * (cs + 1) / ns);
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; r++) acc[r] = 0.0f;
      if (jj < tk) {
#pragma unroll 4
        for (int c = cb; c < ce; c++) {
          const float kval = kv[c * ld + jj];
#pragma unroll
          for (int r = 0; r < R; r++) acc[r] = fmaf(qs[r * d + c], kval, acc[r]);
        }
      }
      if (ns > 1) {
        __syncthreads();  // every thread is done reading part[] of an earlier phase (none) / kv stays valid
        if (jj < tk && cs > 0) {
#pragma unroll
          for (int r = 0; r < R; r++) psc[(((cs - 1) * R + r) << sh) + jj] = acc[r];
        }
        __syncthreads();
      }
      if (jj < tk && cs == 0) {
        for (int s2 = 1; s2 < ns; s2++) {
#pragma unroll
          for (int r = 0; r < R; r++) acc[r] += psc[(((s2 - 1) * R + r) << sh) + jj];
        }
        const int j = j0 + jj;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int delta = j - (i0 + r);
          float sv = acc[r];
          if (delta >= -w && delta <= w) sv += qe[r * W + delta + w];
          sc[r * T + j] = sv;
        }
      }
    }
  }
  __syncthreads();
  PH_STAMP(4);
  // 3. row softmax (softmax.metal:13-41: max, exp, sum, multiply by 1/sum), one wave per row
  {
    const int lane = tid & 63, wv = tid >> 6;
    for (int r = wv; r < R; r += kBlock / 64) {
      float* row = sc + r * T;
      float m = -INFINITY;
      for (int j = lane; j < T; j += 64) m = fmaxf(m, row[j]);
      m = wave_max(m);
      float s = 0.0f;
      for (int j = lane; j < T; j += 64) {
        const float e = expf(row[j] - m);
        row[j] = e;
        s += e;
      }
      s = wave_sum(s);
      const float inv = 1.0f / s;
      for (int j = lane; j < T; j += 64) row[j] *= inv;
    }
  }
  PH_STAMP(5);
  // 4. P·V: thread = (channel c, key slice g of the tile); p comes from LDS as a broadcast
  const int c4 = tid % d, g4 = tid / d;
  float pv[R];
#pragma unroll
  for (int r = 0; r < R; r++) pv[r] = 0.0f;
  for (int j0 = 0; j0 < T; j0 += TK) {
    const int tk = min(TK, T - j0);
    __syncthreads();
    // rows of the tile are dealt to the 256 threads in passes of (256 / TKP) rows, TKP = next power of two ≥ tk: index
    // math is shifts, and the 8-way unrolled body keeps 8 independent loads per thread in flight
    {
      const int sh = 32 - __clz(tk - 1 > 0 ? tk - 1 : 1);  // log2(TKP) for tk ≥ 2
      const int shc = tk <= 1 ? 0 : sh;
      const int jj = tid & ((1 << shc) - 1), rpp = kBlock >> shc, cbase = tid >> shc;
      if (jj < tk) {
#pragma unroll 8
        for (int c = cbase; c < d; c += rpp) kv[c * ld + jj] = vb[(int64_t)c * T + j0 + jj];
      }
    }
    __syncthreads();
    if (g4 < G) {
      const int ja = (int)((int64_t)tk * g4 / G), jb = (int)((int64_t)tk * (g4 + 1) / G);
#pragma unroll 4
      for (int jj = ja; jj < jb; jj++) {
        const float vv = kv[c4 * ld + jj];
#pragma unroll
        for (int r = 0; r < R; r++) pv[r] = fmaf(sc[r * T + j0 + jj], vv, pv[r]);
      }
    }
  }
  if (g4 < G) {
#pragma unroll
    for (int r = 0; r < R; r++) part[(g4 * R + r) * d + c4] = pv[r];
  }
  __syncthreads();
  PH_STAMP(6);
  // 5. combine slices, add the relative-value term, store [H·d, T]
  float* ob = out + (int64_t)n * out_batch_stride + (int64_t)h * d * T;
  for (int idx = tid; idx < R * d; idx += kBlock) {
    const int c = idx / R, r = idx - c * R;
    const int i = i0 + r;
    if (i >= T) continue;
    float o = 0.0f;
    for (int g = 0; g < G; g++) o += part[(g * R + r) * d + c];
    float rel = 0.0f;
    for (int m = 0; m < W; m++) {
      const int j = i + m - w;
      if (j >= 0 && j < T) rel = fmaf(sc[r * T + j], evs[m * d + c], rel);
    }
    ob[(int64_t)c * T + i] = o + rel;
  }
  PH_STAMP(7);
}


// ---------------------------------------------------------------------------------------------------------------------
// MFMA formulation (exact fp32, v_mfma_f32_16x16x4_f32), operands straight from global memory into fragments.
//
// A block owns RV query rows of one head (RV = 16, or 8 valid rows of a 16-row tile when the score strip of 16 rows would
// not fit the LDS) and 8 waves:
//   1. scores: wave w takes key tiles w, w+8, … (16 keys each): S[i][j] = Σ_c (q[c][i]/√d)·k[c][j] — A = the q strip
//      (d/4 fragments per lane, loaded once), B = k read in place ([c][j] is already the B layout: 16 consecutive keys of 4
//      consecutive channels per instruction); the tile after the last is the relative-key logits q·E_kᵀ (B = E_k). No K tile
//      is ever staged: nothing is shared between the waves of a block, so LDS would only add a round trip.
//   2. softmax over the strip in LDS (rel→abs skew = index arithmetic on the 16×(2w+1) logits), one wave per row.
//   3. out[c][i] = Σ_j v[c][j]·P[i][j] + Σ_m E_v[m][c]·P[i][i+m−w]: wave w takes channel tile w (16 channels), A = v read in
//      place, B = Pᵀ from LDS; the relative-value term is ⌈(2w+1)/4⌉ more steps on the same accumulator (abs→rel skew again
//      as index arithmetic).
// A first MFMA attempt in round 1 staged K and V through LDS tiles and was slower than the scalar kernel below; the cost
// was the staging chain, not the arithmetic. `len_ptr` (optional) gives the true length of each batch item inside a
// bucketed schedule: keys ≥ len are excluded exactly (they are not part of the softmax), rows ≥ len are not computed.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kAttWaves = 8;

// One head of the strip, run by NW waves (`wv` = this wave's index among them): scores → softmax → P·V (+ relative values).
// `dst` receives out[c][i] with row stride `dst_ld` (the [H·d, T] tensor in global memory, or the block's [H·d][16] tile in
// LDS when the o-projection is fused). Barriers are block-wide: every wave of the block calls this in lockstep.
// Everything the later phases read from global memory that does not depend on the softmax — the first two key chunks of this
// wave's V rows and its E_v column — is requested together with q and the first K tile, so the head costs ONE exposed memory
// round trip instead of four (the phases are short; at T ≈ 100 the head is a latency chain, not arithmetic).
template <int D, int RV, int NW>
__device__ __forceinline__ void att_head(const float* __restrict__ qb, const float* __restrict__ kb, const float* __restrict__ vb,
                                         const float* __restrict__ ek, const float* __restrict__ ev, float* __restrict__ dst, int dst_ld,
                                         int dst_col0, int T, int Tv, int w, int i0, float* sc, float* qe, int Tp, int lane, int wv) {
  constexpr int NS = D / 4;   // contraction steps over the head dim
  constexpr int NCT = D / 16; // channel tiles of the output
  constexpr int CH = 8;       // P·V steps per prefetch chunk
  constexpr int MAXW4 = 4;    // relative-value steps (2w+1 ≤ 16)
  const int r16 = lane & 15, kq = lane >> 4;
  const int W = 2 * w + 1;
  const float scale = sqrtf((float)D);
  const int nkt = (Tv + 15) >> 4;  // key tiles; tile index nkt is the relative-key tile
  const bool pi_ok = r16 < RV && i0 + r16 < Tv;
  const int nsteps = (Tv + 3) >> 2;

  static_assert(NS % 2 == 0, "head dim must be a multiple of 8");
  constexpr int NH = NS / 2;  // a key tile is fetched in two halves of the contraction: one half in flight, one feeding the MFMAs
  float qa[NS], b0[NH], b1[NH];
  float a0[CH], a1[CH], arel[MAXW4];
  auto load_half = [&](float (&b)[NH], int jt, int half) {
    if (jt < nkt) {
      const int j = jt * 16 + r16;
      const int jc = min(j, Tv - 1);
#pragma unroll
      for (int s = 0; s < NH; s++) b[s] = kb[(int64_t)(4 * (half * NH + s) + kq) * T + jc];
    } else {  // E_k[m][c] as B[k = c][n = m]
      const int mc = min(r16, W - 1);
#pragma unroll
      for (int s = 0; s < NH; s++) b[s] = ek[mc * D + 4 * (half * NH + s) + kq];
    }
  };
  auto load_v = [&](float (&a)[CH], const float* vrow, int s0) {
#pragma unroll
    for (int u = 0; u < CH; u++) {
      const int j = 4 * (s0 + u) + kq;
      a[u] = vrow[min(j, Tv - 1)];
    }
  };
  // ---- one burst of loads: q strip, first K tile, first V chunks and E_v column of this wave's first channel tile
  {
    const int qi = min(i0 + r16, Tv - 1);
#pragma unroll
    for (int s = 0; s < NS; s++) qa[s] = qb[(int64_t)(4 * s + kq) * T + qi];
    if (wv <= nkt) load_half(b0, wv, 0);
    if (wv < NCT) {
      const float* vrow = vb + (int64_t)(wv * 16 + r16) * T;
      load_v(a0, vrow, 0);
      load_v(a1, vrow, CH);
#pragma unroll
      for (int s = 0; s < MAXW4; s++) arel[s] = ev[min(4 * s + kq, W - 1) * D + wv * 16 + r16];
    }
  }

  // ---- 1. scores
  {
    const bool row_ok = pi_ok;
#pragma unroll
    for (int s = 0; s < NS; s++) qa[s] = row_ok ? qa[s] / scale : 0.0f;  // Div of the graph: query / sqrt(k_channels)
    // two accumulators (even / odd steps): the 16x16x4 form needs 40 cycles between dependent issues, 32 between independent
    for (int jt = wv; jt <= nkt; jt += NW) {  // wave-uniform
      const bool col_ok = jt < nkt ? (jt * 16 + r16 < Tv) : (r16 < W);
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      load_half(b1, jt, 1);
#pragma unroll
      for (int s = 0; s < NH; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], col_ok ? b0[s] : 0.0f, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s + 1], col_ok ? b0[s + 1] : 0.0f, acc1, 0, 0, 0);
      }
      if (jt + NW <= nkt) load_half(b0, jt + NW, 0);
#pragma unroll
      for (int s = 0; s < NH; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[NH + s], col_ok ? b1[s] : 0.0f, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[NH + s + 1], col_ok ? b1[s + 1] : 0.0f, acc1, 0, 0, 0);
      }
      // D: row = 4·kq + r (query), col = r16 (key / relative position)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 4 * kq + r;
        const float val = acc0[r] + acc1[r];
        if (row < RV) {
          if (jt < nkt) sc[row * Tp + jt * 16 + r16] = val;
          else qe[row * 17 + r16] = val;
        }
      }
    }
  }
  __syncthreads();

  // ---- 2. softmax (softmax.metal:13-41: max, exp and sum, multiply by 1/sum), relative-key logits added on the way in
  for (int r = wv; r < RV; r += NW) {
    const int ia = i0 + r;
    if (ia >= Tv) break;  // wave-uniform
    float* row = sc + r * Tp;
    const float* qr = qe + r * 17;
    float m = -INFINITY;
    for (int j = lane; j < Tv; j += 64) {
      const int delta = j - ia;
      float sv = row[j];
      if (delta >= -w && delta <= w) sv += qr[delta + w];
      row[j] = sv;
      m = fmaxf(m, sv);
    }
    m = wave_max(m);
    float sum = 0.0f;
    for (int j = lane; j < Tv; j += 64) {
      const float e = expf(row[j] - m);
      row[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Tv; j += 64) row[j] *= inv;
  }
  __syncthreads();

  // ---- 3. P·V + relative-value term: channel tiles wv, wv + NW, …
  for (int ct = wv; ct < NCT; ct += NW) {
    const int c0 = ct * 16;
    const float* vrow = vb + (int64_t)(c0 + r16) * T;        // A[m = channel][k = key]
    const float* prow = sc + min(r16, RV - 1) * Tp;          // B[k = key][n = query row]
    if (ct != wv) {  // later tiles of this wave: their first chunks were not part of the opening burst
      load_v(a0, vrow, 0);
      load_v(a1, vrow, CH);
#pragma unroll
      for (int s = 0; s < MAXW4; s++) arel[s] = ev[min(4 * s + kq, W - 1) * D + c0 + r16];
    }
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, accb = {0.0f, 0.0f, 0.0f, 0.0f};
    auto run = [&](const float (&a)[CH], int s0) {
      float b[CH];
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int j = 4 * (s0 + u) + kq;
        b[u] = prow[min(j, Tv - 1)];
      }
#pragma unroll
      for (int u = 0; u < CH; u += 2) {
        const int j = 4 * (s0 + u) + kq;
        const bool ok = j < Tv, ok2 = j + 4 < Tv;  // also false for the steps past nsteps of the last chunk
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? a[u] : 0.0f, (ok && pi_ok) ? b[u] : 0.0f, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x4f32(ok2 ? a[u + 1] : 0.0f, (ok2 && pi_ok) ? b[u + 1] : 0.0f, accb, 0, 0, 0);
      }
    };
    for (int s0 = 0; s0 < nsteps; s0 += 2 * CH) {
      run(a0, s0);
      if (s0 + CH >= nsteps) break;
      if (s0 + 2 * CH < nsteps) load_v(a0, vrow, s0 + 2 * CH);
      run(a1, s0 + CH);
      if (s0 + 3 * CH < nsteps) load_v(a1, vrow, s0 + 3 * CH);
    }
    // relative values: A = E_v[m][c], B = P[i][i + m − w] (abs→rel skew as an index)
    const int ia = i0 + r16;
#pragma unroll
    for (int s = 0; s < MAXW4; s++) {
      const int mrel = 4 * s + kq;
      const bool mok = mrel < W;
      const int j = ia + mrel - w;
      const bool jok = mok && pi_ok && j >= 0 && j < Tv;
      const float b = prow[jok ? j : 0];
      if (4 * s < W) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mok ? arel[s] : 0.0f, jok ? b : 0.0f, acc, 0, 0, 0);
    }
    // D: row = channel 4·kq + r, col = query row r16
    if (pi_ok) {
#pragma unroll
      for (int r = 0; r < 4; r++) dst[(int64_t)(c0 + 4 * kq + r) * dst_ld + dst_col0 + r16] = acc[r] + accb[r];
    }
  }
}

template <int D, int RV>
__global__ __launch_bounds__(64 * kAttWaves) void rel_attention_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                           const float* __restrict__ v, const float* __restrict__ ek,
                                                                           const float* __restrict__ ev, float* __restrict__ out, int T,
                                                                           int w, int64_t in_batch_stride, int64_t out_batch_stride,
                                                                           const int* __restrict__ len_ptr, int Tp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sc = smem;            // [RV][Tp] scores → probabilities
  float* qe = smem + RV * Tp;  // [16][17] relative-key logits of the strip
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i0 = blockIdx.x * RV, h = blockIdx.y, n = blockIdx.z;
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;  // block-uniform true length
  if (i0 >= Tv) return;
  const int64_t hoff = (int64_t)n * in_batch_stride + (int64_t)h * D * T;
  att_head<D, RV, kAttWaves>(q + hoff, k + hoff, v + hoff, ek, ev, out + (int64_t)n * out_batch_stride + (int64_t)h * D * T, T, i0, T, Tv, w, i0,
                             sc, qe, Tp, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention BLOCK of an encoder layer in one launch: y = o_proj(rel_attention(q, k, v)); out = LN_c(x + y)·γ + β.
// (GraphExecutor.swift: MatMul/Softmax/skew arms + Conv :1739-1810 + Add :741-779 + the LayerNorm chain :2071-2125.)
// A block owns 16 columns (query rows) and ALL channels: it runs the heads one after the other into an [H·d][16] tile in
// LDS, then the k = 1 output projection on that tile (v_mfma_f32_16x16x4_f32, 16-row tiles of the packed 16-wide weight
// image over the 8 waves, bias first), adds the residual and normalises over the channels — which it can do exactly
// (two passes: mean, then mean of squared deviations, like the graph's ReduceMean / Sub / Pow / ReduceMean) because every
// channel of its columns is in the block. Replaces three launches (attention, conv_o, add+LayerNorm) and two round trips of
// the [H·d, T] tensor through memory.
template <int D>
__global__ __launch_bounds__(64 * kAttWaves) void attention_block_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                        const float* __restrict__ v, const float* __restrict__ ek,
                                                                        const float* __restrict__ ev, const float* __restrict__ wo16,
                                                                        const float* __restrict__ bo, const float* __restrict__ xres,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        float* __restrict__ out, int H, int T, int w, int64_t in_batch_stride,
                                                                        int64_t x_batch_stride, const int* __restrict__ len_ptr, int Tp,
                                                                        int o_nsteps, float eps) {
  constexpr int RV = 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sc = smem;                 // [2][16][Tp]: one strip per head of the pair in flight
  float* qe = sc + 2 * RV * Tp;     // [2][16][17]
  float* att = qe + 2 * 16 * 17;    // [H·D][16]: B operand of the projection (stride 16: conflict-free fragment reads)
  const int C = H * D;
  float* red = att + C * 16;        // [kAttWaves][16] column partials of the LayerNorm
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r16 = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.x * RV, n = blockIdx.z;
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;
  if (i0 >= Tv) return;
  // two heads at a time, four waves each (Piper: H = 2, one pass): half the dependent phases of a head-after-head loop
  for (int hp = 0; hp < H; hp += 2) {
    const int hg = wave >> 2, wv = wave & 3;
    const int h = min(hp + hg, H - 1);  // odd H: the second group repeats the last head (same values, written twice)
    const int64_t hoff = (int64_t)n * in_batch_stride + (int64_t)h * D * T;
    att_head<D, RV, 4>(q + hoff, k + hoff, v + hoff, ek, ev, att + h * D * 16, 16, 0, T, Tv, w, i0, sc + hg * RV * Tp, qe + hg * 16 * 17, Tp, lane,
                       wv);
    __syncthreads();  // the strips are rewritten by the next pair; after the last pair: the tile is complete
  }
  // ---- output projection: row tile mt = 16 channels; waves take tiles wave, wave + 8, …  (C/16 = 12 tiles for Piper)
  const int ntiles = C >> 4;
  const bool col_ok = i0 + r16 < Tv;
  const int colc = min(i0 + r16, Tv - 1);
  const float* xb = xres + (int64_t)n * x_batch_stride;
  constexpr int MAXT = 2;  // tiles per wave held in registers (C ≤ 256)
  float val[MAXT][4];
  float s1 = 0.0f;
#pragma unroll
  for (int ti = 0; ti < MAXT; ti++) {
    const int mt = wave + kAttWaves * ti;
    if (mt < ntiles) {  // wave-uniform
      f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = bo[16 * mt + 4 * kq + r];  // bias first (CPUBackend.conv1d)
      float res[4];
#pragma unroll
      for (int r = 0; r < 4; r++) res[r] = xb[(int64_t)(16 * mt + 4 * kq + r) * T + colc];
      const float* wa = wo16 + (int64_t)mt * o_nsteps * 64 + lane;
      const int nst = C >> 2;
      for (int s0 = 0; s0 < nst; s0 += 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = wa[(s0 + u) * 64];
#pragma unroll
        for (int u = 0; u < 8; u++) b[u] = att[(4 * (s0 + u) + kq) * 16 + r16];
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        val[ti][r] = res[r] + acc[r];  // Add(x, y)
        s1 += val[ti][r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) val[ti][r] = 0.0f;
    }
  }
  // ---- LayerNorm over the channels of each column: D layout has the column on lane & 15, channels on (lane >> 4, register)
  s1 += __shfl_xor(s1, 16, 64);
  s1 += __shfl_xor(s1, 32, 64);
  if (lane < 16) red[wave * 16 + lane] = s1;
  __syncthreads();
  float mean = 0.0f;
#pragma unroll
  for (int wv = 0; wv < kAttWaves; wv++) mean += red[wv * 16 + r16];
  mean = mean / (float)C;
  __syncthreads();
  float s2 = 0.0f;
#pragma unroll
  for (int ti = 0; ti < MAXT; ti++) {
    const int mt = wave + kAttWaves * ti;
    if (mt < ntiles) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        val[ti][r] -= mean;
        s2 += val[ti][r] * val[ti][r];
      }
    }
  }
  s2 += __shfl_xor(s2, 16, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (lane < 16) red[wave * 16 + lane] = s2;
  __syncthreads();
  float var = 0.0f;
#pragma unroll
  for (int wv = 0; wv < kAttWaves; wv++) var += red[wv * 16 + r16];
  var = var / (float)C;
  const float sd = sqrtf(var + eps);
  float* ob = out + (int64_t)n * x_batch_stride;
#pragma unroll
  for (int ti = 0; ti < MAXT; ti++) {
    const int mt = wave + kAttWaves * ti;
    if (mt < ntiles && col_ok) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int c = 16 * mt + 4 * kq + r;
        ob[(int64_t)c * T + i0 + r16] = (val[ti][r] / sd) * gamma[c] + beta[c];
      }
    }
  }
}

template <int D, int RV>
int launch_att_mfma(hipStream_t s, const float* q, const float* k, const float* v, const float* ek, const float* ev, float* out, int N, int H,
                    int T, int w, int64_t in_bs, int64_t out_bs, const int* len_ptr) {
  const int Tp = ((T + 31) / 32) * 32 + 2;  // ≡ 2 mod 32: the Pᵀ fragment reads (16 rows × 2 keys per half wave) hit 32 distinct banks
  const size_t lds = ((size_t)RV * Tp + 16 * 17) * sizeof(float);
  if (lds > 160 * 1024) return -1;
  static bool raised[ph::kMaxDevices] = {};
  if (lds > 64 * 1024 && ph::lds_optin_needed(raised))
    (void)hipFuncSetAttribute((const void*)rel_attention_mfma_kernel<D, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid((unsigned)ph::ceil_div(T, RV), (unsigned)H, (unsigned)N);
  hipLaunchKernelGGL((rel_attention_mfma_kernel<D, RV>), grid, dim3(64 * kAttWaves), lds, s, q, k, v, ek, ev, out, T, w, in_bs, out_bs, len_ptr, Tp);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// v3: the same MFMA formulation with K and V staged through LDS in full rows.
// r2d measurement: the register-fragment kernel above spends its time in the texture path — every fragment load is a
// wave-instruction that touches 4 × 64-byte segments (16 keys × 4 channels), 68 of them per wave, and the block's 8 waves
// share one address unit — 10 µs per head at T = 112 for 0.03 GFLOP. Here a K (then V) tile of 128 keys × all D channels is
// fetched ONCE per block with 16-byte loads along the key axis (rows of 512 contiguous bytes), written to LDS with row
// strides chosen so that both fragment shapes read conflict-free (K rows ≡ 16 floats mod 32: B[k = channel][n = key];
// V rows ≡ 2 mod 32: A[m = channel][k = key]), and all fragments come from LDS. V's first tile is requested before the score
// phase and parked in registers, so the softmax hides its latency. Needs T % 4 == 0 (16-byte rows; every bucketed plan has
// T % 16 == 0) and T ≤ 1024 (score strip + tiles in 160 KiB); other shapes take the register-fragment kernel.
constexpr int kTK = 128;               // keys per staged tile
constexpr int kLdK = kTK + 16;         // ≡ 16 (mod 32)
constexpr int kLdV = kTK + 2;          // ≡ 2 (mod 32)

// SPLIT (long rows): blockIdx.y = head · nsplit + part; the block covers only the key tiles of its part, leaves the probabilities
// unnormalised (e^{s − m} with the part's own row maximum m) and writes its share of P·V plus (m, Σ e) per query row to scratch;
// rel_attention_merge_kernel combines the parts (out = Σ_p e^{m_p − M} o_p / Σ_p e^{m_p − M} l_p). A block's time is what its CU
// pulls in — all of K and V of its head, 688 KB at T = 896, at ≈ 8 B/clk (DESIGN.md finding 9; two tiles in flight did not help) —
// so halving the bytes per block is what shortens the launch.
#ifdef PH_ATT_LDS_TRACE
__device__ unsigned long long* ph_att_trace_buf;
#define PH_ASTAMP(k) do { if ((threadIdx.x & 63) == 0 && ph_att_trace_buf) ph_att_trace_buf[((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * kAttWaves + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_ASTAMP(k) do { } while (0)
#endif
template <int D, bool SPLIT>
__global__ __launch_bounds__(64 * kAttWaves) void rel_attention_lds_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                          const float* __restrict__ v, const float* __restrict__ ek,
                                                                          const float* __restrict__ ev, float* __restrict__ out, int T,
                                                                          int w, int64_t in_batch_stride, int64_t out_batch_stride,
                                                                          const int* __restrict__ len_ptr, int Tp, int nsplit,
                                                                          float* __restrict__ part_o, float* __restrict__ part_ml) {
  constexpr int RV = 16, NS = D / 4, NCT = D / 16;
  constexpr int NT = 64 * kAttWaves;                 // threads
  constexpr int F4 = D * (kTK / 4);                  // float4 slots of a staged tile
  constexpr int SLOTS = (F4 + NT - 1) / NT;          // per thread (6 for D = 96)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                       // [D][kLdK] (K) or [D][kLdV] (V)
  float* sc = tile + D * kLdK;              // [16][Tp]
  float* qs = sc + RV * Tp;                 // [D][16]  A fragments of the q strip
  float* qe = qs + D * 16;                  // [16][17]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int i0 = blockIdx.x * RV, n = blockIdx.z;
  const int h = SPLIT ? (int)blockIdx.y / nsplit : (int)blockIdx.y;
  const int part = SPLIT ? (int)blockIdx.y - h * nsplit : 0;
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;
  if (i0 >= Tv) return;
  const int W = 2 * w + 1;
  const int64_t hoff = (int64_t)n * in_batch_stride + (int64_t)h * D * T;
  const float* qb = q + hoff;
  const float* kb = k + hoff;
  const float* vb = v + hoff;
  const float scale = sqrtf((float)D);
  const int ntile_all = (Tv + kTK - 1) / kTK;
  // this block's key tiles [tb, te) and keys [kbeg, kend): parts of ceil(ntile / nsplit) tiles (a part past the end is empty: the
  // merge kernel applies the same rule and skips it)
  const int tpp = SPLIT ? (ntile_all + nsplit - 1) / nsplit : ntile_all;
  const int tb = SPLIT ? part * tpp : 0;
  const int te = SPLIT ? min(tb + tpp, ntile_all) : ntile_all;
  if (SPLIT && tb >= te) return;
  const int kbeg = tb * kTK, kend = min(te * kTK, Tv);
  const int ntile = te - tb;

  float4 stg[SLOTS];
  // tile `t0` of `src` → registers: slot e = tid + NT·i ↔ (row = e / 32, 16-byte column = e % 32); rows are T floats apart and
  // T % 4 == 0, so every load is aligned; columns past the row end are clamped (their keys are masked at use)
  auto fetch = [&](const float* src, int t0) {
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
      const int e = tid + NT * i;
      const int row = min(e >> 5, D - 1), c4 = e & 31;
      const int col = min(t0 + 4 * c4, T - 4);
      stg[i] = *(const float4*)(src + (int64_t)row * T + col);
    }
  };
  auto commit = [&](int ld, int t0) {  // registers → LDS tile with row stride ld (8-byte stores: ld is even)
#pragma unroll
    for (int i = 0; i < SLOTS; i++) {
      const int e = tid + NT * i;
      const int row = e >> 5, c4 = e & 31;
      if (row < D) {
        float4 val = stg[i];
        if (t0 + 4 * c4 > T - 4) val = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // clamped load: not this tile's columns
        float2* dst = (float2*)(tile + row * ld + 4 * c4);
        dst[0] = make_float2(val.x, val.y);
        dst[1] = make_float2(val.z, val.w);
      }
    }
  };

  PH_ASTAMP(0);
  // ---- 0. q strip → LDS (scaled: Div of the graph), first K tile in flight
  fetch(kb, kbeg);
  for (int e = tid; e < D * 16; e += NT) {
    const int c = e >> 4, i = e & 15;
    const bool ok = i0 + i < Tv;
    const float qv = qb[(int64_t)c * T + min(i0 + i, Tv - 1)];
    qs[c * 16 + i] = ok ? qv / scale : 0.0f;
  }
  // relative-value fragments of phase 3: requested here, with everything else of the block's first round trip (asked for where they are
  // used they cost the launch a cold round trip of their own after the softmax)
  float arel[4];
  if (wave < NCT) {
#pragma unroll
    for (int s = 0; s < 4; s++) arel[s] = ev[min(4 * s + kq, W - 1) * D + wave * 16 + r16];
  }
  // relative-key logits: wave 7 straight from global (E_k is tiny), while the others wait for the K tile
  float ekf[NS];
  if (wave == kAttWaves - 1) {
    const int mc = min(r16, W - 1);
#pragma unroll
    for (int s = 0; s < NS; s++) ekf[s] = ek[mc * D + 4 * s + kq];
  }
  commit(kLdK, kbeg);
  PH_ASTAMP(1);
  __syncthreads();
  PH_ASTAMP(2);
  if (wave == kAttWaves - 1) {
    f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < NS; s += 2) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[(4 * s + kq) * 16 + r16], r16 < W ? ekf[s] : 0.0f, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[(4 * (s + 1) + kq) * 16 + r16], r16 < W ? ekf[s + 1] : 0.0f, a1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) qe[(4 * kq + r) * 17 + r16] = a0[r] + a1[r];
  }

  // ---- 1. scores, one staged K tile (8 key tiles of 16, one per wave) at a time
  for (int t = 0; t < ntile; t++) {
    const int t0 = kbeg + t * kTK;
    if (t + 1 < ntile) fetch(kb, t0 + kTK);  // next K tile in flight
    else fetch(vb, kbeg);                    // … or V's first tile: lands under the softmax
    const int j0 = t0 + wave * 16;
    if (j0 < Tv) {  // wave-uniform
      const bool col_ok = j0 + r16 < Tv;
      f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < NS; s += 2) {
        const float b0 = tile[(4 * s + kq) * kLdK + wave * 16 + r16], b1 = tile[(4 * (s + 1) + kq) * kLdK + wave * 16 + r16];
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[(4 * s + kq) * 16 + r16], col_ok ? b0 : 0.0f, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[(4 * (s + 1) + kq) * 16 + r16], col_ok ? b1 : 0.0f, a1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) sc[(4 * kq + r) * Tp + j0 + r16] = a0[r] + a1[r];
    }
    __syncthreads();  // every wave is done with this K tile
    if (t + 1 < ntile) {
      commit(kLdK, t0 + kTK);
      __syncthreads();
    }
  }

  PH_ASTAMP(3);
  // ---- 2. softmax (softmax.metal:13-41), relative-key logits added on the way in; V tile 0 sits in registers meanwhile
  // One key tile (T ≤ 128: every utterance up to factor 9) — a lane holds its two keys of a row in registers and the wave's two rows go
  // through together: one LDS read and one write per element instead of three of each in three dependent passes (r3 phase trace,
  // tools/probe/attprobe2: the softmax was 5.6 k of the kernel's 22 k cycles at T = 112). Same operations in the same order as the loop below.
  if (!SPLIT && ntile == 1 && RV == 2 * kAttWaves) {
    const int ra = wave, rb = wave + kAttWaves;
    const int ia = i0 + ra, ib = i0 + rb;
    const bool la = ia < Tv, lb = ib < Tv;  // wave-uniform
    float* rowa = sc + ra * Tp;
    float* rowb = sc + rb * Tp;
    const int j0 = kbeg + lane, j1 = kbeg + lane + 64;
    const bool k0 = j0 < kend, k1 = j1 < kend;
    float a0 = -INFINITY, a1 = -INFINITY, b0 = -INFINITY, b1 = -INFINITY;
    if (la) {
      if (k0) { a0 = rowa[j0]; const int dl = j0 - ia; if (dl >= -w && dl <= w) a0 += qe[ra * 17 + dl + w]; }
      if (k1) { a1 = rowa[j1]; const int dl = j1 - ia; if (dl >= -w && dl <= w) a1 += qe[ra * 17 + dl + w]; }
    }
    if (lb) {
      if (k0) { b0 = rowb[j0]; const int dl = j0 - ib; if (dl >= -w && dl <= w) b0 += qe[rb * 17 + dl + w]; }
      if (k1) { b1 = rowb[j1]; const int dl = j1 - ib; if (dl >= -w && dl <= w) b1 += qe[rb * 17 + dl + w]; }
    }
    const float ma = wave_max(fmaxf(a0, a1)), mb = wave_max(fmaxf(b0, b1));
    const float ea0 = k0 ? expf(a0 - ma) : 0.0f, ea1 = k1 ? expf(a1 - ma) : 0.0f;
    const float eb0 = k0 ? expf(b0 - mb) : 0.0f, eb1 = k1 ? expf(b1 - mb) : 0.0f;
    const float inva = 1.0f / wave_sum(ea0 + ea1), invb = 1.0f / wave_sum(eb0 + eb1);
    if (la) { if (k0) rowa[j0] = ea0 * inva; if (k1) rowa[j1] = ea1 * inva; }
    if (lb) { if (k0) rowb[j0] = eb0 * invb; if (k1) rowb[j1] = eb1 * invb; }
  } else
  for (int r = wave; r < RV; r += kAttWaves) {
    const int ia = i0 + r;
    if (ia >= Tv) break;  // wave-uniform
    float* row = sc + r * Tp;
    const float* qr = qe + r * 17;
    float m = -INFINITY;
    for (int j = kbeg + lane; j < kend; j += 64) {
      const int delta = j - ia;
      float sv = row[j];
      if (delta >= -w && delta <= w) sv += qr[delta + w];
      row[j] = sv;
      m = fmaxf(m, sv);
    }
    m = wave_max(m);
    float sum = 0.0f;
    for (int j = kbeg + lane; j < kend; j += 64) {
      const float e = expf(row[j] - m);
      row[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if constexpr (SPLIT) {  // unnormalised: (m, Σ e) go to the merge
      if (lane == 0) {
        float* ml = part_ml + (((int64_t)n * gridDim.y + blockIdx.y) * 2) * T;
        ml[ia] = m;
        ml[T + ia] = sum;
      }
    } else {
      const float inv = 1.0f / sum;
      for (int j = lane; j < Tv; j += 64) row[j] *= inv;
    }
  }
  PH_ASTAMP(4);
  commit(kLdV, kbeg);
  __syncthreads();
  PH_ASTAMP(5);

  // ---- 3. P·V over the staged V tiles (waves 0 … NCT−1: one 16-channel tile each), then the relative-value steps
  const bool pi_ok = i0 + r16 < Tv;
  const float* prow = sc + r16 * Tp;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, accb = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int t = 0; t < ntile; t++) {
    const int t0 = kbeg + t * kTK;
    if (t + 1 < ntile) fetch(vb, t0 + kTK);
    if (wave < NCT) {
      const float* vrow = tile + (wave * 16 + r16) * kLdV;
      const int nst = min(kTK, Tv - t0 + 3) >> 2;  // steps of 4 keys in this tile
      for (int s = 0; s < nst; s += 2) {
        const int j = t0 + 4 * s + kq;
        const bool ok = j < Tv, ok2 = j + 4 < Tv && s + 1 < nst;
        const float a0v = vrow[4 * s + kq], a1v = vrow[min(4 * (s + 1) + kq, kTK - 1)];
        const float b0v = prow[min(j, Tv - 1)], b1v = prow[min(j + 4, Tv - 1)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? a0v : 0.0f, (ok && pi_ok) ? b0v : 0.0f, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_16x16x4f32(ok2 ? a1v : 0.0f, (ok2 && pi_ok) ? b1v : 0.0f, accb, 0, 0, 0);
      }
    }
    if (t + 1 < ntile) {
      __syncthreads();
      commit(kLdV, t0 + kTK);
      __syncthreads();
    }
  }
  if (wave < NCT) {
    const int ia = i0 + r16;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int mrel = 4 * s + kq;
      const bool mok = mrel < W;
      const int j = ia + mrel - w;
      const bool jok = mok && pi_ok && j >= kbeg && j < kend;
      const float b = prow[jok ? j : kbeg];
      if (4 * s < W) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mok ? arel[s] : 0.0f, jok ? b : 0.0f, acc, 0, 0, 0);
    }
    float* ob = SPLIT ? part_o + ((int64_t)n * gridDim.y + blockIdx.y) * D * T : out + (int64_t)n * out_batch_stride + (int64_t)h * D * T;
    if (pi_ok) {
#pragma unroll
      for (int r = 0; r < 4; r++) ob[(int64_t)(wave * 16 + 4 * kq + r) * T + i0 + r16] = acc[r] + accb[r];
    }
  }
  PH_ASTAMP(6);
}

// out[n][h·D + c][i] = Σ_p e^{m_p − M} o_p[c][i] / Σ_p e^{m_p − M} l_p over the parts that hold keys (the kernel's own rule)
template <int D>
__global__ __launch_bounds__(256) void rel_attention_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                  float* __restrict__ out, int H, int T, int nsplit, int64_t out_batch_stride,
                                                                  const int* __restrict__ len_ptr) {
  const int n = blockIdx.z, h = blockIdx.y;
  const int Tv = len_ptr ? min(len_ptr[n], T) : T;
  const int ntile_all = (Tv + kTK - 1) / kTK;
  const int tpp = (ntile_all + nsplit - 1) / nsplit;
  const int live = min(nsplit, (ntile_all + tpp - 1) / tpp);  // parts with at least one key tile
  const int64_t pbase = ((int64_t)n * H + h) * nsplit;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < D * T; e += gridDim.x * 256) {
    const int c = e / T, i = e - c * T;
    if (i >= Tv) continue;
    float M = -INFINITY;
    for (int p = 0; p < live; p++) M = fmaxf(M, part_ml[(pbase + p) * 2 * T + i]);
    float num = 0.0f, den = 0.0f;
    for (int p = 0; p < live; p++) {
      const float sc = expf(part_ml[(pbase + p) * 2 * T + i] - M);
      num = fmaf(sc, part_o[(pbase + p) * D * T + (int64_t)c * T + i], num);
      den = fmaf(sc, part_ml[(pbase + p) * 2 * T + T + i], den);
    }
    out[(int64_t)n * out_batch_stride + ((int64_t)h * D + c) * T + i] = num / den;
  }
}

template <int D>
int launch_att_lds(hipStream_t s, const float* q, const float* k, const float* v, const float* ek, const float* ev, float* out, int N, int H,
                   int T, int w, int64_t in_bs, int64_t out_bs, const int* len_ptr, int nsplit, float* part_o, float* part_ml) {
  const int Tp = ((T + 31) / 32) * 32 + 2;
  const size_t lds = ((size_t)D * kLdK + (size_t)16 * Tp + (size_t)D * 16 + 16 * 17) * sizeof(float);
  if (lds > 160 * 1024) return -1;
  static bool raised[ph::kMaxDevices] = {};
  static bool raised_split[ph::kMaxDevices] = {};
  if (nsplit > 1) {
    if (H * nsplit > 65535) return -1;
    if (lds > 64 * 1024 && ph::lds_optin_needed(raised_split))
      (void)hipFuncSetAttribute((const void*)rel_attention_lds_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    dim3 grid((unsigned)ph::ceil_div(T, 16), (unsigned)(H * nsplit), (unsigned)N);
    hipLaunchKernelGGL((rel_attention_lds_kernel<D, true>), grid, dim3(64 * kAttWaves), lds, s, q, k, v, ek, ev, out, T, w, in_bs, out_bs, len_ptr, Tp,
                       nsplit, part_o, part_ml);
    dim3 mgrid((unsigned)std::min<int64_t>(ph::ceil_div((int64_t)D * T, 256), 64), (unsigned)H, (unsigned)N);
    hipLaunchKernelGGL((rel_attention_merge_kernel<D>), mgrid, dim3(256), 0, s, part_o, part_ml, out, H, T, nsplit, out_bs, len_ptr);
    return 0;
  }
  if (lds > 64 * 1024 && ph::lds_optin_needed(raised))
    (void)hipFuncSetAttribute((const void*)rel_attention_lds_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid((unsigned)ph::ceil_div(T, 16), (unsigned)H, (unsigned)N);
  hipLaunchKernelGGL((rel_attention_lds_kernel<D, false>), grid, dim3(64 * kAttWaves), lds, s, q, k, v, ek, ev, out, T, w, in_bs, out_bs, len_ptr, Tp,
                     1, nullptr, nullptr);
  return 0;
}

template <int D>
int launch_att_block(hipStream_t s, const float* q, const float* k, const float* v, const float* ek, const float* ev, const float* wo16,
                     const float* bo, const float* xres, const float* gamma, const float* beta, float* out, int N, int H, int T, int w,
                     int64_t in_bs, int64_t x_bs, const int* len_ptr, int o_nsteps, float eps) {
  const int Tp = ((T + 31) / 32) * 32 + 2;
  const size_t lds = ((size_t)2 * 16 * Tp + 2 * 16 * 17 + (size_t)H * D * 16 + kAttWaves * 16) * sizeof(float);
  if (lds > 160 * 1024) return -1;
  static bool raised[ph::kMaxDevices] = {};
  if (lds > 64 * 1024 && ph::lds_optin_needed(raised))
    (void)hipFuncSetAttribute((const void*)attention_block_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  dim3 grid((unsigned)ph::ceil_div(T, 16), 1u, (unsigned)N);
  hipLaunchKernelGGL((attention_block_kernel<D>), grid, dim3(64 * kAttWaves), lds, s, q, k, v, ek, ev, wo16, bo, xres, gamma, beta, out, H, T, w,
                     in_bs, x_bs, len_ptr, Tp, o_nsteps, eps);
  return 0;
}

}  // namespace

namespace ph {
int launch_rel_attention(piper_hip_ctx* ctx, hipStream_t s, const float* q, const float* k, const float* v, const float* ek, const float* ev, float* out,
                         int N, int H, int d, int T, int w, int64_t in_batch_stride, int64_t out_batch_stride, const int* len_ptr);

// Key-split form: as many parts as keep the grid within one block per CU (more only queue up behind each other — r2 at T = 896:
// 2 parts 40 µs, 3 parts 46, 4 parts 48, unsplit 48; at T = 560: 32 / 29 / 29, unsplit 35; T = 336: 25 → 20, T = 224: 19.6 → 18.6), at
// most one part per key tile (so nothing changes up to T = 128). part_o: N·H·parts·d·T floats, part_ml: N·H·parts·2·T floats of scratch.
int rel_attention_split_parts(piper_hip_ctx* ctx, int N, int H, int d, int T, int w) {
  static const bool no_split = getenv("PIPER_HIP_ATT_NO_SPLIT") != nullptr;
  static const int forced = [] { const char* e = getenv("PIPER_HIP_ATT_SPLIT"); return e ? std::min(std::max(atoi(e), 2), 8) : 0; }();  // tuning
  static const int min_t = [] { const char* e = getenv("PIPER_HIP_ATT_SPLIT_MIN_T"); return e ? atoi(e) : 129; }();  // two key tiles or more
  if (no_split || d != 96 || T < min_t || T > 1024 || (T & 3) != 0 || w < 0 || 2 * w + 1 > 16 || N < 1 || H < 1) return 1;
  const int ntile = (T + kTK - 1) / kTK;
  const int64_t blocks = (int64_t)ceil_div(T, 16) * H * N;
  const int parts = forced ? forced : (int)std::min<int64_t>(ntile, ctx->num_cus / std::max<int64_t>(1, blocks));
  return std::max(1, std::min(parts, 8));
}

int launch_rel_attention_split(piper_hip_ctx* ctx, hipStream_t s, const float* q, const float* k, const float* v, const float* ek,
                               const float* ev, float* out, int N, int H, int d, int T, int w, int64_t in_batch_stride,
                               int64_t out_batch_stride, const int* len_ptr, int nsplit, float* part_o, float* part_ml) {
  if (N <= 0 || T <= 0) return PIPER_HIP_OK;
  if (nsplit < 2 || !part_o || !part_ml || d != 96 || H > 65535 || N > 65535 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) != 0 ||
      (in_batch_stride & 3) != 0 || nsplit > (T + kTK - 1) / kTK || nsplit > 8 || T > 1024 || (T & 3) != 0 || w < 0 || 2 * w + 1 > 16)
    return launch_rel_attention(ctx, s, q, k, v, ek, ev, out, N, H, d, T, w, in_batch_stride, out_batch_stride, len_ptr);
  if (launch_att_lds<96>(s, q, k, v, ek, ev, out, N, H, T, w, in_batch_stride, out_batch_stride, len_ptr, nsplit, part_o, part_ml) != 0)
    return launch_rel_attention(ctx, s, q, k, v, ek, ev, out, N, H, d, T, w, in_batch_stride, out_batch_stride, len_ptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rel_attention (split) launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

int launch_rel_attention(piper_hip_ctx* ctx, hipStream_t s, const float* q, const float* k, const float* v, const float* ek,
                         const float* ev, float* out, int N, int H, int d, int T, int w, int64_t in_batch_stride,
                         int64_t out_batch_stride, const int* len_ptr) {
  if (N <= 0 || T <= 0) return PIPER_HIP_OK;
  static const bool no_mfma = getenv("PIPER_HIP_ATT_SCALAR") != nullptr;  // A/B switch: the round-1 scalar kernel
  static const bool no_lds = getenv("PIPER_HIP_ATT_NO_LDS") != nullptr;  // A/B switch: register-fragment MFMA kernel
  if (!no_mfma && !no_lds && d == 96 && w >= 0 && 2 * w + 1 <= 16 && H <= 65535 && N <= 65535 && T <= 1024 && T >= 4 && (T & 3) == 0 &&
      (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && (in_batch_stride & 3) == 0) {
    if (launch_att_lds<96>(s, q, k, v, ek, ev, out, N, H, T, w, in_batch_stride, out_batch_stride, len_ptr, 1, nullptr, nullptr) == 0) {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rel_attention (lds) launch failed: %s", hipGetErrorString(e));
      return PIPER_HIP_OK;
    }
  }
  if (!no_mfma && d == 96 && w >= 0 && 2 * w + 1 <= 16 && H <= 65535 && N <= 65535 && T <= 4096) {
    int rc = T <= 2048 ? launch_att_mfma<96, 16>(s, q, k, v, ek, ev, out, N, H, T, w, in_batch_stride, out_batch_stride, len_ptr)
                       : launch_att_mfma<96, 8>(s, q, k, v, ek, ev, out, N, H, T, w, in_batch_stride, out_batch_stride, len_ptr);
    if (rc == 0) {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rel_attention (mfma) launch failed: %s", hipGetErrorString(e));
      return PIPER_HIP_OK;
    }
  }
  if (len_ptr) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rel_attention: per-item lengths need the MFMA kernel (head_dim 96, window ≤ 7)");
  if (d > kBlock) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rel_attention: head_dim %d > %d", d, kBlock);
  if (T > 4096) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rel_attention: T=%d exceeds 4096 (reference max-phonemes cap)", T);
  if (H > 65535 || N > 65535) PH_FAIL(PIPER_HIP_ERR_SHAPE, "rel_attention: heads/batch too large");
  // (An MFMA formulation — 16 query rows per block on v_mfma_f32_16x16x4 — was built and measured slower at every T:
  //  47 vs 30 µs at T = 112, 233 vs 161 µs at T = 896.  The op is a chain of LDS/HBM round trips, not arithmetic, and 4×
  //  fewer blocks each re-staging K and V lengthen the chain.)
  const int G = kBlock / d;
  const int TK = T < kKeyTile ? T : kKeyTile;
  // query rows per block: fewer rows → more blocks (short utterances have only T/R·H of them); 4 rows also keeps the score
  // strip of the longest utterances inside the 160 KiB of LDS
  const int R = (T > 2048 || (int64_t)ceil_div(T, 8) * H * N < ctx->num_cus) ? 4 : 8;
  const size_t lds = (size_t)(R * d + R * (2 * w + 1) + G * R * d + (size_t)d * (TK + 1) + (size_t)R * T + 2 * (2 * w + 1) * d + (size_t)kBlock * R) * sizeof(float);
  if (lds > 160 * 1024) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rel_attention: needs %zu B of LDS", lds);
  const void* fn = R == 4 ? (const void*)rel_attention_kernel<4> : (const void*)rel_attention_kernel<8>;
  static size_t configured[2] = {0, 0};
  if (lds > 64 * 1024 && lds > configured[R == 4]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rel_attention: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    configured[R == 4] = 160 * 1024;
  }
  dim3 grid((unsigned)ceil_div(T, R), (unsigned)H, (unsigned)N);
  if (R == 4)
    hipLaunchKernelGGL(rel_attention_kernel<4>, grid, dim3(kBlock), lds, s, q, k, v, ek, ev, out, H, d, T, w, in_batch_stride,
#ifdef PH_ATT_STAMPS
                       out_batch_stride, G, TK, nullptr);
#else
                       out_batch_stride, G, TK);
#endif
  else
    hipLaunchKernelGGL(rel_attention_kernel<8>, grid, dim3(kBlock), lds, s, q, k, v, ek, ev, out, H, d, T, w, in_batch_stride,
#ifdef PH_ATT_STAMPS
                       out_batch_stride, G, TK, nullptr);
#else
                       out_batch_stride, G, TK);
#endif
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rel_attention launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}
// Measured at T = 112 (r2d): the fused launch takes 29 µs against 27 µs for the three launches it replaces — the heads'
// fragment-shaped loads keep the texture path busy — so the VOICE schedule uses it only on request (PIPER_HIP_ATT_BLOCK=1);
// the op-level entry point always runs it.
bool attention_block_wanted() {
  static const bool on = getenv("PIPER_HIP_ATT_BLOCK") != nullptr && getenv("PIPER_HIP_ATT_SCALAR") == nullptr;
  return on;
}

bool attention_block_eligible(int H, int d, int w, int T) {
  if (d != 96 || H * d > 256 || (H * d) % 32 || 2 * w + 1 > 16 || w < 0 || T > 2048 || T < 1) return false;
  const int Tp = ((T + 31) / 32) * 32 + 2;
  return ((size_t)2 * 16 * Tp + 2 * 16 * 17 + (size_t)H * d * 16 + kAttWaves * 16) * sizeof(float) <= 160 * 1024;
}

// attention + output projection + residual + LayerNorm in one launch (attention_block_kernel). wo16 = the 16-wide packed
// fragment image of conv_o (pack_conv_weights(..., tm = 16)), o_nsteps its padded step count. Returns UNSUPPORTED (nothing
// launched) outside the kernel's geometry: the caller schedules the three separate launches instead.
int launch_attention_block(piper_hip_ctx* ctx, hipStream_t s, const float* q, const float* k, const float* v, const float* ek, const float* ev,
                           const float* wo16, const float* bo, const float* xres, const float* gamma, const float* beta, float* out, int N,
                           int H, int d, int T, int w, int64_t in_batch_stride, int64_t x_batch_stride, const int* len_ptr, int o_nsteps,
                           float eps) {
  if (N <= 0 || T <= 0) return PIPER_HIP_OK;
  if (!attention_block_eligible(H, d, w, T) || N > 65535) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "attention_block: geometry not covered");
  if (launch_att_block<96>(s, q, k, v, ek, ev, wo16, bo, xres, gamma, beta, out, N, H, T, w, in_batch_stride, x_batch_stride, len_ptr, o_nsteps,
                           eps) != 0)
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "attention_block: needs more than 160 KiB of LDS");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "attention_block launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}
}  // namespace ph

PH_EXPORT int piper_hip_rel_attention_f32(piper_hip_ctx* ctx, const float* q, const float* k, const float* v,
                                          const float* emb_rel_k, const float* emb_rel_v, int64_t n, int64_t heads,
                                          int64_t head_dim, int64_t t, int64_t window, float** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (n < 0 || heads <= 0 || head_dim <= 0 || t < 0 || window < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "rel_attention: bad shape");
  if (n * heads * head_dim * t > 0x7fffffff) PH_FAIL(PIPER_HIP_ERR_SHAPE, "rel_attention: tensor too large");
  const size_t cnt = (size_t)(n * heads * head_dim * t);
  int rc = ph::ensure_out(ctx, out, cnt, 0);
  if (rc) return rc;
  if (cnt == 0) return PIPER_HIP_OK;
  if (!q || !k || !v || !emb_rel_k || !emb_rel_v) PH_FAIL(PIPER_HIP_ERR_ARG, "rel_attention: null input");
  ph::StreamScope ss(ctx, stream);
  const int64_t bs = heads * head_dim * t;
  const int nsplit = ph::rel_attention_split_parts(ctx, (int)n, (int)heads, (int)head_dim, (int)t, (int)window);
  if (nsplit > 1) {
    void *po = nullptr, *pml = nullptr;
    rc = ctx->pool.alloc((size_t)(n * heads * nsplit * head_dim * t) * sizeof(float), &po);
    if (rc) return rc;
    ph::defer_free(ctx, po);
    rc = ctx->pool.alloc((size_t)(n * heads * nsplit * 2 * t) * sizeof(float), &pml);
    if (rc) return rc;
    ph::defer_free(ctx, pml);
    rc = ph::launch_rel_attention_split(ctx, ss.s, q, k, v, emb_rel_k, emb_rel_v, *out, (int)n, (int)heads, (int)head_dim, (int)t, (int)window, bs,
                                        bs, nullptr, nsplit, (float*)po, (float*)pml);
  } else
    rc = ph::launch_rel_attention(ctx, ss.s, q, k, v, emb_rel_k, emb_rel_v, *out, (int)n, (int)heads, (int)head_dim, (int)t,
                                  (int)window, bs, bs, nullptr);
  if (rc) return rc;
  return ss.finish("rel_attention_f32");
}

PH_EXPORT int piper_hip_attention_block_f32(piper_hip_ctx* ctx, const float* q, const float* k, const float* v, const float* emb_rel_k,
                                            const float* emb_rel_v, const float* w_o, const float* b_o, const float* x,
                                            const float* gamma, const float* beta, int64_t n, int64_t heads, int64_t head_dim, int64_t t,
                                            int64_t window, float eps, float** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (n < 0 || heads <= 0 || head_dim <= 0 || t < 0 || window < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "attention_block: bad shape");
  if (n * heads * head_dim * t > 0x7fffffff) PH_FAIL(PIPER_HIP_ERR_SHAPE, "attention_block: tensor too large");
  const int C = (int)(heads * head_dim);
  const size_t cnt = (size_t)(n * C * t);
  int rc = ph::ensure_out(ctx, out, cnt, 0);
  if (rc) return rc;
  if (cnt == 0) return PIPER_HIP_OK;
  if (!q || !k || !v || !emb_rel_k || !emb_rel_v || !w_o || !b_o || !x || !gamma || !beta) PH_FAIL(PIPER_HIP_ERR_ARG, "attention_block: null input");
  if (!ph::attention_block_eligible((int)heads, (int)head_dim, (int)window, (int)t))
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "attention_block: covered geometry is head_dim 96, heads·head_dim ≤ 256, window ≤ 7, T ≤ 2048 — compose "
                                       "rel_attention + conv1d + add_layernorm instead");
  ph::StreamScope ss(ctx, stream);
  void* pw = nullptr;
  rc = ctx->pool.alloc(ph::packed_conv_floats(C, C, 1, 16) * sizeof(float), &pw);
  if (rc) return rc;
  ph::defer_free(ctx, pw);
  ph::pack_conv_weights(ss.s, w_o, C, C, 1, (float*)pw, 16);
  const int o_nsteps = (int)(ph::packed_conv_floats(C, C, 1, 16) / ((size_t)ph::ceil_div(C, 16) * 64));
  const int64_t bs = (int64_t)C * t;
  rc = ph::launch_attention_block(ctx, ss.s, q, k, v, emb_rel_k, emb_rel_v, (const float*)pw, b_o, x, gamma, beta, *out, (int)n, (int)heads,
                                  (int)head_dim, (int)t, (int)window, bs, bs, nullptr, o_nsteps, eps);
  if (rc) return rc;
  return ss.finish("attention_block_f32");
}
namespace { PH_WARM(attention, (rel_attention_lds_kernel<96, false>)); }
#ifdef PH_ATT_LDS_TRACE
void ph_att_set_trace(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(ph_att_trace_buf), &buf, sizeof buf); }
#endif
