// conv_short.hip — Conv1d on SHORT rows (one utterance: T = 14 … 900 phoneme columns, F = 42 … 2 700 frame columns): every conv of the
// text encoder and of the flow, and the generator's conv_pre.
//
// Reference: conv1d_f32 (Kernels/conv1d.metal:28-71) behind the Conv arm (GraphExecutor.swift:1739-1810); the gate is the
// Tanh / Sigmoid / Mul arms (:2017-2045, :741-779) of the WaveNet layer, LayerNorm the ReduceMean … Div chain (:2071-2125).
//
// Why a second kernel next to conv_stream_kernel (round 3). Those launches hold 0.01–0.25 GFLOP: the matrix pipe needs < 1 µs, and
// what a launch costs is the bytes every CU has to pull through its vector L1 (DESIGN.md findings 9, 10; r3 probes
// `tools/probe/sharedprobe.hip`: a CU streams data that other CUs read too at ≈ 60 B/clk when every wave load is one contiguous
// 256-byte run, and at a third of that when it is four 64-byte row segments — which is what the streaming kernel's B fragment is,
// once per TAP). The streaming kernel's 16×16 tile of the flow's gated conv pulls 2 × 61 KB of weight fragments (a tanh row tile and
// its sigmoid partner) and 240 segment loads of activations per block. Here:
//   * the activation window of a wave's contraction slice is fetched ONCE ([channels of the slice] × [16·NT + K − 1] columns, one
//     dword per element, all taps and both column tiles served from it) and laid down in a wave-private piece of LDS — no block
//     barrier: a wave stages what only it reads; LayerNorm-on-load, the true-length mask and the Flip/Split channel map are
//     applied on the way in, so the K-loop is ds_read_b32 + MFMA with no vector-ALU work;
//   * a gated conv packs 8 tanh rows and their 8 sigmoid rows into ONE 16-row tile (fragment image `w16g`); the two halves meet in
//     the epilogue through one lane exchange (lane ^ 32). A block pulls 61 KB of weights for 32 columns instead of 123 KB for 16;
//   * NT = 2 column tiles per wave share every weight fragment whenever that still leaves about one block per CU.
// Weight fragments stream as before: one coalesced 256-byte load per contraction step, the wave's whole slice requested up front.
// The contraction is split over the waves of the block (KS ≤ 16) and summed in fixed order through LDS (deterministic).
#include "conv.h"
#include "conv_kernels.hpp"

namespace ph {
namespace detail {

// Channel quads per staged chunk (compile time: it sizes the register arrays). A wave's slice of the contraction is
// ceil(Cin / 4 / KS) quads; the voice's convs have Cin = 192 (3 / 6 / 12 quads at KS = 16 / 8 / 4), 96 or 768 (12 at KS = 16).
constexpr int short_qc(int K, int PRO, int BT) {
  return K == 1 ? 8 : K == 3 ? (PRO == PRO_LN ? 6 : 12) : K == 5 ? (BT == 1024 ? 3 : 6) : 6;
}
constexpr int short_pitch(int W) {
  int p = (W + 3) & ~3;
  return (p % 32) == 0 && p > 16 ? p + 4 : p;
}

// Piece-wise staging geometry: the window of a wave is [rows][W = 16·NT + K − 1]; its first 16·NT columns ("main": the output
// columns' own positions shifted by −padL) are fetched 64 / (16·NT) rows per load, the K − 1 halo columns 8 rows per load (8 lanes per
// row), so every index is a shift or a mask of the lane id and a row step is a constant added to ONE per-lane offset.
template <int K, int NT, int PRO, bool GATE, int BT>
__global__ __launch_bounds__(BT) void conv_short_kernel(const ConvArgs p, const int nspans, const int mtiles, const int ks_log2, const int nsteps,
                                                        const int rows_alloc) {
  constexpr int W = 16 * NT + K - 1;
  constexpr int PITCH = short_pitch(W);
  constexpr int QC = short_qc(K, PRO, BT), ROWS = 4 * QC, NS = QC * K;
  constexpr int WA = 16 * NT, RPA = 64 / WA;        // main piece: columns, rows per load (WA = 48: one row per load, 48 lanes of it)
  constexpr int NLA = (ROWS + RPA - 1) / RPA;       // main-piece loads per chunk
  constexpr int NLB = K > 1 ? (ROWS + 7) / 8 : 0;   // halo-piece loads per chunk (8 rows × 8 lanes)
  constexpr int NE = 4 * NT;                        // accumulator registers of a tile
  constexpr int OOB = 0x7fffffff;  // byte offset no buffer of this library reaches (host-checked): the load returns 0 without touching memory
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  PH_SSTAMP(0);
  // The argument block is read lazily by default — a dependent scalar round trip (cold: ≈ 0.3 µs) wherever a field is first needed,
  // ten of them in a row in the first version of this kernel. Naming every field here makes the compiler fetch the block in one burst.
  const float *px = p.x, *pw = p.w, *pbias = p.bias, *pres = p.res, *pskip = p.skip;
  float *py = p.y, *py2 = p.y2;
  const int Cin = p.Cin, Cout = p.Cout, padL = p.padL, Lin = p.Lin, Lout = p.Lout, in_base = p.in_ch_base, in_sign = p.in_ch_sign;
  const int* len_ptr = p.len_ptr;
  const int len_mul = p.len_mul;
  const int64_t xbs = p.x_batch_stride;
  asm volatile("" ::"s"(px), "s"(pw), "s"(pbias), "s"(pres), "s"(pskip), "s"(py), "s"(py2));
  asm volatile("" ::"s"(Cin), "s"(Cout), "s"(padL), "s"(Lin), "s"(Lout), "s"(in_base), "s"(in_sign), "s"(len_ptr), "s"(len_mul), "s"(xbs));
  const int KS = 1 << ks_log2;
  const int WT = (BT / 64) >> ks_log2;
  const int tw = wave >> ks_log2, ks = wave & (KS - 1);
  const int n = blockIdx.y;
  const int tile = (int)blockIdx.x * WT + tw;
  const bool in_grid = tile < mtiles * nspans;
  // a runtime division lands in vector registers and would make every descriptor below "divergent" (waterfall loops): pin the quotient
  const int span = __builtin_amdgcn_readfirstlane(in_grid ? tile / mtiles : 0);
  const int mt = in_grid ? tile - span * mtiles : 0;
  const int t0 = span * 16 * NT;
  const int j = lane & 15, kk = lane >> 4;
  // LDS: per wave [rows_alloc][PITCH] window (+ LayerNorm operands), then the split-K exchange [tile][slice][NE][64]
  const int wave_floats = rows_alloc * PITCH + (PRO == PRO_LN ? 2 * 48 + 2 * ROWS : 0);
  float* xs = lds + wave * wave_floats;
  float* st = xs + rows_alloc * PITCH;
  float* red = lds + (BT / 64) * wave_floats + tw * (KS * NE * 64);
  const int rows_out = GATE ? Cout / 2 : Cout;
  const int nquads = (Cin + 3) >> 2;
  const int q_begin = (int)(((int64_t)nquads * ks) >> ks_log2), q_end = (int)(((int64_t)nquads * (ks + 1)) >> ks_log2);
  const int nq = q_end - q_begin;
  // the item's true length: a global load that nothing below waits for until the window is written to LDS
  // (branch-free: a load under `if (len_ptr)` is waited for at the end of that branch — a whole memory round trip before the first operand)
  // … and kept in a VECTOR register on purpose (the opaque zero hides its uniformity): a uniform value is moved to a scalar register
  // with v_readfirstlane right after the load, which again waits for it ahead of the operand loads
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const int len_raw = (len_ptr ? len_ptr + n : (const int*)pw)[vzero];
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  PH_SSTAMP(1);
  if (in_grid && nq > 0) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(px + (int64_t)n * xbs), 0, (int)(xbs * 4), 0x00020000);
    // the wave's slice of its row tile's fragment image: steps [q_begin·K, q_end·K) — a load past it returns 0
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc((void*)(pw + ((int64_t)mt * nsteps + (int64_t)q_begin * K) * 64), 0, nq * K * 256, 0x00020000);
    // per-lane staging constants
    const int colA = (WA & (WA - 1)) == 0 ? (lane & (WA - 1)) : lane, rA = (WA & (WA - 1)) == 0 ? lane / WA : 0;  // main piece
    const int posA = t0 - padL + colA;
    const bool okA = colA < WA && posA >= 0 && posA < Lin;
    const int offA = (in_sign * rA * Lin + posA) * 4;         // + chunk base + i · stepA
    const int stepA = in_sign * RPA * Lin * 4;
    const int colB = WA + (lane & 7), rB = lane >> 3;         // halo piece
    const int posB = t0 - padL + colB;
    const bool okB = K > 1 && (lane & 7) < K - 1 && posB >= 0 && posB < Lin;
    const int offB = (in_sign * rB * Lin + posB) * 4;
    const int stepB = in_sign * 8 * Lin * 4;
    const int nchunks = (nq + QC - 1) / QC;
    // (the body is a lambda so that chunk 0 — the only one in the common case — is straight-line code: at a loop header the compiler
    // merges the wait counts of both entries and would wait for the length load before the first operand load is issued)
    auto do_chunk = [&](const int c) __attribute__((always_inline)) {
      const int qc0 = q_begin + c * QC;              // first quad of the chunk
      const int rv = min(min(ROWS, 4 * (nq - c * QC)), Cin - 4 * qc0);  // valid channel rows in this chunk (wave-uniform)
      const int cbase = (in_base + in_sign * 4 * qc0) * Lin * 4;
      float av[NS], xa[NLA], xb2[NLB > 0 ? NLB : 1];
      // ---- every load of the chunk in one burst: weights (contiguous 256-byte fragments), window, LayerNorm operands ----
      const int voffA = lane * 4 + c * NS * 256;
#pragma unroll
      for (int s = 0; s < NS; s++) av[s] = bload(rw, voffA + s * 256, 0);
#pragma unroll
      for (int i = 0; i < NLA; i++) xa[i] = bload(rx, (okA && rA + RPA * i < rv) ? offA + cbase + i * stepA : OOB, 0);
#pragma unroll
      for (int i = 0; i < NLB; i++) xb2[i] = bload(rx, (okB && rB + 8 * i < rv) ? offB + cbase + i * stepB : OOB, 0);
      if constexpr (PRO == PRO_LN) {
        {  // gamma / beta of the chunk's channel rows: one row per lane, handed to the staging loop through LDS
          const int ch = min(4 * qc0 + min(lane, ROWS - 1), Cin - 1);
          const float g = p.ln_gamma[ch], b = p.ln_beta[ch];
          if (lane < ROWS) {
            st[96 + lane] = g;
            st[96 + ROWS + lane] = b;
          }
        }
        // statistics of the window's columns from the producer's per-slot partial sums (ConvArgs::stats_out), one column per lane;
        // slots added in fixed order ⇒ deterministic. (A wave recomputes what its block mates compute too: L1 hits, one round trip.)
        if (c == 0) {
          const int parts = (Cin + 15) >> 4;
          const float* sb = p.ln_stats + (int64_t)n * parts * Lin * 2;
          const int col = min(max(t0 - padL + min(lane, W - 1), 0), Lin - 1);
          float2 pr[16];
#pragma unroll
          for (int q = 0; q < 16; q++) {
            pr[q] = *(const float2*)(sb + ((int64_t)min(q, parts - 1) * Lin + col) * 2);
            if (q >= parts) pr[q] = make_float2(0.0f, 0.0f);
          }
          // slot q holds (Σ y, Σ (y − mean_q)²) of its n_q rows: Chan's combination (see conv_kernels.hpp load_ln_stats)
          float s1 = 0.0f;
#pragma unroll
          for (int q = 0; q < 16; q++) s1 += pr[q].x;
          const float mean = s1 / (float)Cin;
          float m2 = 0.0f;
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const float nq_ = (float)min(16, Cin - 16 * q);
            if (q < parts) {
              const float dm = pr[q].x / nq_ - mean;
              m2 += pr[q].y + nq_ * (dm * dm);
            }
          }
          const float var = m2 / (float)Cin;
          if (lane < W) {
            st[lane] = mean;
            st[48 + lane] = 1.0f / sqrtf(var + p.ln_eps);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      PH_SSTAMP(5);
      int lr = len_raw;
      asm volatile("" : "+v"(lr));  // the length is first LOOKED AT here, behind the chunk's loads (the compiler would hoist the arithmetic, and its wait, above them)
      const int Lv = len_ptr ? min(lr * len_mul, Lin) : Lin;
      // ---- window → LDS (wave-private): prologue + zero padding applied once per element ----
      auto put = [&](const float x, const int row, const int col, const int pos, const bool okp) {
        const bool ok = okp && row < rv && pos < Lv;
        float v = x;
        if constexpr (PRO == PRO_LN) {
          v = ((v - st[col]) * st[48 + col]) * st[96 + row] + st[96 + ROWS + row];
          // the row-tile-0 blocks materialise the normalised tensor once (their waves cover all channels, the spans all columns)
          if (mt == 0 && p.ln_out && ok && col >= padL && col < padL + 16 * NT) (p.ln_out + (int64_t)n * xbs)[(int64_t)(4 * qc0 + row) * Lin + pos] = v;
        }
        if (row < rows_alloc) xs[row * PITCH + col] = ok ? v : 0.0f;
      };
#pragma unroll
      for (int i = 0; i < NLA; i++)
        if (colA < WA) put(xa[i], rA + RPA * i, colA, posA, okA);
#pragma unroll
      for (int i = 0; i < NLB; i++)
        if ((lane & 7) < K - 1) put(xb2[i], rB + 8 * i, colB, posB, okB);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      PH_SSTAMP(6);
      // ---- K loop: one ds_read_b32 per column tile + MFMA per (channel quad, tap) ----
      const float* xw = xs + kk * PITCH + j;
      const int nqc = min(QC, nq - c * QC);
#pragma unroll
      for (int qi = 0; qi < QC; qi++) {
        if (qi < nqc) {  // wave-uniform
#pragma unroll
          for (int k = 0; k < K; k++) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
              const float b = xw[4 * qi * PITCH + k + 16 * nt];
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[qi * K + k], b, acc[nt], 0, 0, 0);
            }
          }
        }
      }
      if (c + 1 < nchunks) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    };
    do_chunk(0);
    for (int c = 1; c < nchunks; c++) do_chunk(c);
  }

  PH_SSTAMP(2);
  // ---- split-K exchange: every slice leaves its partial tile in LDS; after the barrier the tile's NE accumulator registers are
  // shared out among the block's waves (one register = 64 outputs per wave), each summed in fixed slice order 0, 1, … on top of the
  // bias (bias first, like CPUBackend.swift:46-63) and finished by that wave — the reduction and the epilogue (tanh / sigmoid,
  // residual loads, stores) of a 16-slice tile run on NE waves side by side instead of on one wave after the other.
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) red[(ks * NE + nt * 4 + r) * 64 + lane] = acc[nt][r];
  __syncthreads();
  PH_SSTAMP(3);
  int lr_t = len_raw;
  asm volatile("" : "+v"(lr_t));
  const int Lv_t = len_ptr ? min(lr_t * len_mul, Lin) : Lin;
  const bool tile_live = in_grid && !(len_ptr && Lout == Lin && t0 >= Lv_t);  // a tile past the item's true length stores nothing anyone reads
  if (!tile_live) return;
  // stats_out needs the 16 rows of a column together: then a wave takes a whole column tile (4 registers) instead of one register
  const bool by_tile = !GATE && p.stats_out != nullptr;
  const int units = by_tile ? NT : NE;
  for (int u = ks; u < units; u += KS) {  // wave-uniform trip count
    const int nt = by_tile ? u : u >> 2;
    const int r_lo = by_tile ? 0 : (u & 3), r_hi = by_tile ? 4 : (u & 3) + 1;
    const int col = t0 + 16 * nt + j;
    float ps1 = 0.0f, pv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (r < r_lo || r >= r_hi) continue;  // wave-uniform
      const int i = 4 * kk + r;  // row of the 16-row tile
      int co;
      bool row_ok;
      if constexpr (GATE) {
        const int h = 8 * mt + (i & 7);
        row_ok = h < rows_out;
        co = (i < 8 ? 0 : rows_out) + h;
      } else {
        co = 16 * mt + i;
        row_ok = co < Cout;
      }
      float v = pbias ? pbias[row_ok ? co : 0] : 0.0f;
      if (!row_ok) v = 0.0f;
      const float* src = red + (nt * 4 + r) * 64 + lane;
      float part[16];
#pragma unroll
      for (int s2 = 0; s2 < 16; s2++) part[s2] = s2 < KS ? src[s2 * NE * 64] : 0.0f;
#pragma unroll
      for (int s2 = 0; s2 < 16; s2++)
        if (s2 < KS) v += part[s2];
      if constexpr (GATE) {
        // rows 0–7 of the tile (lanes 0–31) are tanh rows, rows 8–15 (lanes 32–63) their sigmoid partners: one exchange brings them together
        const float sg = __shfl_xor(v, 32, 64);
        const float g = tanhf(v) * sigmoid_stable(sg);
        const int row = 8 * mt + (i & 7);
        if (lane < 32 && row < rows_out && col < Lout) (py + (int64_t)n * p.y_batch_stride)[(p.out_ch_base + p.out_ch_sign * row) * p.y_len + col] = g;
      } else {
        const int row = co;
        const bool okp = row_ok && col < Lout;
        const int rowc = min(row, rows_out - 1), colc = min(col, Lout - 1);
        switch (p.epilogue) {  // wave-uniform
          case EPI_STORE: {
            const EpiIn e = epi_load<EPI_STORE>(p, n, rowc, colc);
            if (okp) {
              epi_finish<EPI_STORE>(p, n, row, col, v, e);
              const float val = pres ? v + e.a : v;  // exactly what epi_finish stored
              ps1 += val;
              pv[r] = val;
            }
          } break;
          case EPI_RELU: if (okp) epi_finish<EPI_RELU>(p, n, row, col, v, EpiIn{}); break;
          case EPI_TANH: if (okp) epi_finish<EPI_TANH>(p, n, row, col, v, EpiIn{}); break;
          case EPI_RSUB: { const EpiIn e = epi_load<EPI_RSUB>(p, n, rowc, colc); if (okp) epi_finish<EPI_RSUB>(p, n, row, col, v, e); } break;
          case EPI_WN_RES_SKIP: { const EpiIn e = epi_load<EPI_WN_RES_SKIP>(p, n, rowc, colc); if (okp) epi_finish<EPI_WN_RES_SKIP>(p, n, row, col, v, e); } break;
          case EPI_WN_SKIP_LAST: { const EpiIn e = epi_load<EPI_WN_SKIP_LAST>(p, n, rowc, colc); if (okp) epi_finish<EPI_WN_SKIP_LAST>(p, n, row, col, v, e); } break;
          default: break;  // host-checked
        }
      }
    }
    if constexpr (!GATE) {
      if (by_tile) {  // LayerNorm statistics of this tile's 16 rows per column: (Σ y, Σ (y − mean_tile)²) (ConvArgs::stats_out)
        const float cnt = (float)max(1, min(16, Cout - 16 * mt));
        ps1 += __shfl_xor(ps1, 32, 64);
        ps1 += __shfl_xor(ps1, 16, 64);
        const float ms = ps1 / cnt;
        float q2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float dv = pv[r] - ms;
          if (16 * mt + 4 * kk + r < Cout) q2 += dv * dv;
        }
        q2 += __shfl_xor(q2, 32, 64);
        q2 += __shfl_xor(q2, 16, 64);
        if (lane < 16 && col < Lout) {
          const int parts = (Cout + 15) >> 4;
          float* sb = p.stats_out + (int64_t)n * parts * p.y_len * 2;
          *(float2*)(sb + ((int64_t)mt * p.y_len + col) * 2) = make_float2(ps1, q2);
        }
      }
    }
  }
  PH_SSTAMP(4);
}

template <int K, int NT, int PRO, bool GATE, int BT>
void launch_short_one(hipStream_t s, const ConvArgs& a, int nspans, int mtiles, int ks_log2, int nsteps, int rows_alloc, dim3 grid, size_t lds) {
  if (lds > 64 * 1024) {
    static bool configured[kMaxDevices] = {};
    if (lds_optin_needed(configured))
      (void)hipFuncSetAttribute((const void*)conv_short_kernel<K, NT, PRO, GATE, BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((conv_short_kernel<K, NT, PRO, GATE, BT>), grid, dim3(BT), lds, s, a, nspans, mtiles, ks_log2, nsteps, rows_alloc);
}

template <int K, int NT, int PRO, bool GATE>
bool launch_short_bt(hipStream_t s, const ConvArgs& a, int BT, int nspans, int mtiles, int ks_log2, int nsteps, int rows_alloc, dim3 grid, size_t lds) {
  switch (BT) {
    case 256: launch_short_one<K, NT, PRO, GATE, 256>(s, a, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds); return true;
    case 512: launch_short_one<K, NT, PRO, GATE, 512>(s, a, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds); return true;
    case 1024: launch_short_one<K, NT, PRO, GATE, 1024>(s, a, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds); return true;
  }
  return false;
}

template <int K>
bool launch_short_k(hipStream_t s, const ConvArgs& a, int NT, int BT, int nspans, int mtiles, int ks_log2, int nsteps, int rows_alloc, dim3 grid, size_t lds) {
  if (a.gate) {
    if constexpr (K == 5) {
      if (NT == 3) return launch_short_bt<K, 3, PRO_NONE, true>(s, a, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds);
      if (NT == 2) return launch_short_bt<K, 2, PRO_NONE, true>(s, a, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds);
      return launch_short_bt<K, 1, PRO_NONE, true>(s, a, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds);
    }
    return false;
  }
  if (a.prologue == PRO_LN) {
    if constexpr (K == 1 || K == 3) {
      if (NT == 2) return launch_short_bt<K, 2, PRO_LN, false>(s, a, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds);
      return launch_short_bt<K, 1, PRO_LN, false>(s, a, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds);
    }
    return false;
  }
  if (NT == 2) return launch_short_bt<K, 2, PRO_NONE, false>(s, a, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds);
  return launch_short_bt<K, 1, PRO_NONE, false>(s, a, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds);
}

}  // namespace detail

using namespace detail;

static int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  if (!e) return dflt;
  const int v = atoi(e);
  return v < lo || v > hi ? dflt : v;
}

// Returns 1 when the conv was enqueued on the short-row kernel, 0 when the caller should use the streaming kernel, < 0 on error.
int try_launch_conv_short(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a_in) {
  static const bool off = getenv("PIPER_HIP_NO_SHORT") != nullptr;
  if (off) return 0;
  ConvArgs a = a_in;
  if (a.dil != 1 || (a.K != 1 && a.K != 3 && a.K != 5 && a.K != 7)) return 0;
  if (a.prologue != PRO_NONE && a.prologue != PRO_LN) return 0;
  if (a.gate) {
    if (!a.w16g || a.epilogue != EPI_STORE || a.res || a.stats_out || a.K != 5 || (a.Cout % 16)) return 0;
  } else {
    if (!a.w16) return 0;
    if (a.epilogue != EPI_STORE && a.epilogue != EPI_RELU && a.epilogue != EPI_TANH && a.epilogue != EPI_RSUB && a.epilogue != EPI_WN_RES_SKIP &&
        a.epilogue != EPI_WN_SKIP_LAST)
      return 0;
  }
  if (a.prologue == PRO_LN && (a.Cin > 256 || (a.K != 1 && a.K != 3))) return 0;
  // Which convs take this kernel (r3 rocprofv3 A/B on the factor-8 utterance, in-graph kernel time, this kernel vs the streaming one):
  // gated k5 7.4 vs 8.8 µs, k7 8.4 vs 9.9, k3 behind LayerNorm 9.7 vs 9.9 — taken; plain k1 5.2–5.9 vs 5.25, k1 behind LayerNorm 6.9–8.3
  // vs 5.7, plain k3 on 768 channels 12.6 vs 10.6 — left to the streaming kernel. PIPER_HIP_SHORT_MASK re-opens the choice.
  static const int mask = env_int("PIPER_HIP_SHORT_MASK", 1 | 2 | 4, 0, 63);
  const int cls = a.gate ? 1 : a.K == 7 ? 2 : (a.K == 3 && a.prologue == PRO_LN) ? 4 : a.K == 3 ? 8 : (a.K == 1 && a.prologue == PRO_NONE) ? 16 : a.K == 1 ? 32 : 0;
  if (!(mask & cls)) return 0;
  if (a.x_batch_stride * 4 >= 0x7fffffffLL) return 0;
  const int mtiles = a.gate ? a.Cout / 16 : (int)ceil_div(a.Cout, 16);
  const int nquads = (a.Cin + 3) / 4;
  // Column tiles per wave (NT). Measured (r3, `tools/probe/short_ab.sh`, factor-8 gated conv, in-graph kernel time): NT = 1 → 504 blocks of
  // 8 waves 7.5 µs; NT = 2 → 264 blocks 10.1 µs (eight CUs get two); NT = 3 → 168 blocks of 16 waves 8.4 µs although its busiest CU
  // pulls a third fewer bytes — small work units win (DESIGN.md finding 10), so one column tile per wave unless forced.
  const int nt_max = a.gate ? 3 : 2;
  int NT = 1;
  static const int force_nt = env_int("PIPER_HIP_SHORT_NT", 0, 1, 3);
  if (force_nt && force_nt <= nt_max) NT = force_nt;
  const int nspans = (int)ceil_div(a.Lout, 16 * NT);
  const int64_t tiles = (int64_t)mtiles * nspans;
  if (tiles * a.N > 8 * (int64_t)ctx->num_cus) return 0;  // enough tiles for the bulk kernels (same threshold as the 16-wide streaming tiles)
  // contraction slices per tile: fill the CUs' wave slots (want ≈ 16 waves per CU) while a slice keeps ≥ 2 quads
  static const int want_waves = env_int("PIPER_HIP_SHORT_WAVES_PER_CU", 16, 1, 64);
  static const int ks_max_log2 = env_int("PIPER_HIP_SHORT_KS_MAX_LOG2", 3, 0, 4);  // 8 slices = 512-thread blocks: the 1024-thread variants measured 0.3–0.8 µs slower
  int ks_log2 = 0;
  while (ks_log2 < ks_max_log2 && tiles * a.N * (2 << ks_log2) <= (int64_t)ctx->num_cus * want_waves * 11 / 10 && nquads / (2 << ks_log2) >= 2) ks_log2++;
  const int KS = 1 << ks_log2;
  const int BT = KS <= 4 ? 256 : 64 * KS;
  const int QC = short_qc(a.K, a.prologue, BT);
  if (ceil_div(nquads, KS) > 2 * QC) return 0;  // long slices: the streaming kernel's operand ring overlaps their round trips, this kernel does not
  const int WT = (BT / 64) / KS;
  const int W = 16 * NT + a.K - 1;
  const int pitch = short_pitch(W);
  const int rows_alloc = 4 * (int)std::min<int64_t>(QC, ceil_div(nquads, KS));  // channel rows a wave ever stages at once
  const int wave_floats = rows_alloc * pitch + (a.prologue == PRO_LN ? 96 + 8 * QC : 0);
  const size_t lds = ((size_t)(BT / 64) * wave_floats + (size_t)KS * WT * NT * 4 * 64) * sizeof(float);
  if (lds > 160 * 1024) return 0;
  if (a.N > 65535) return 0;
  const int nsteps = padded_steps(a.Cin, a.K, 16);
  if (a.gate) a.w = a.w16g; else a.w = a.w16;
  dim3 grid((unsigned)ceil_div(tiles, WT), (unsigned)a.N);
  bool ok = false;
  switch (a.K) {
    case 1: ok = launch_short_k<1>(s, a, NT, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds); break;
    case 3: ok = launch_short_k<3>(s, a, NT, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds); break;
    case 5: ok = launch_short_k<5>(s, a, NT, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds); break;
    case 7: ok = launch_short_k<7>(s, a, NT, BT, nspans, mtiles, ks_log2, nsteps, rows_alloc, grid, lds); break;
  }
  if (!ok) return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_short launch failed: %s", hipGetErrorString(e));
  return 1;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(conv_short, (conv_short_kernel<7, 1, 0, false, 512>)); } }
