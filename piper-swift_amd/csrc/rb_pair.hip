// rb_pair.hip — two chained ResBlock convs of the HiFi-GAN generator in ONE kernel, the intermediate kept in LDS.
//
// Reference: the generator's ResBlocks are chains of "x ← x + conv_d(lrelu(x))" (ResBlock2, Piper medium) or
// "x ← x + conv_1(lrelu(conv_d(lrelu(x))))" (ResBlock1, Piper high) — GraphExecutor dispatches every LeakyRelu / Conv / Add of
// them on its own (GraphExecutor.swift:2047-2069 LeakyRelu, :1739-1810 Conv, :861-899 Add; conv1d.metal:28-71 is the conv). conv_win_kernel runs
// one conv per launch: per conv it reads x (window) + the residual and writes the result — three fp32 passes over a
// [C × L] tensor — and each 32×32 output tile pays a full block prologue (window staging, weight-ring start) and epilogue for
// as little as 48 MFMAs. Measured on medium/factor 8 (r2 PMC, tools/probe/run_pmc.sh): 13 vector + 13 scalar instructions
// per MFMA, matrix pipe 37 % busy.
//
// Here a block computes BOTH convs of a pair for a column tile:
//   stage   lrelu(x) window  [C × (256 + 2·pa)]  → LDS (aligned float4, zero outside [0, len))
//   conv a  x1 = [x +] conv_a(lrelu(x)) + bias on 8 column tiles of 32 (= the output tiles + a halo each side: 16 columns
//           when conv b reaches ≤ 16 positions ⇒ 7 output tiles, else 48 ⇒ 5); lrelu(x1) → LDS (zero outside [0, len): it is
//           the NEXT conv's zero-padded input), raw x1 → the dead x window (ResBlock2's residual)
//   conv b  y = (x1 | x) + conv_b(lrelu(x1)) + bias on the output tiles → global (buffer stores)
// One read of x and one write of y per pair instead of three reads and two writes, one launch instead of two, and a wave
// runs 2 × (2 tiles × 16·K·C/32) MFMAs between its prologue and epilogue. The halo recompute costs 8/7 (8/5) on conv a.
// The K-loops hold no vector ALU work beyond one address add per LDS read pair (see lrelu_max below for why).
// Exact fp32 (v_mfma_f32_32x32x2_f32), the same contraction order as conv_win_kernel (tap-major, channel pairs ascending).
#include <algorithm>
#include <type_traits>

#include "conv_win.h"

namespace ph {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRA = 8;        // float4 weight groups in flight (32 steps)
constexpr int kStepPad = 16;  // conv_win's fragment image pads steps to a multiple of 16
constexpr int kWN = 4, kNTW = 2, kT1 = kWN * kNTW;
constexpr int kColsA = 32 * kT1;        // 256 columns of x1 per block
// x1 columns each side of the block's output columns: 16 when conv b reaches ≤ 16 positions (7 output tiles per block),
// otherwise 48 (5 output tiles; Piper medium's third ResBlock: kernel 7, dilation 12 ⇒ reach 36)
__host__ __device__ constexpr int halo_of(int pb) { return pb <= 16 ? 16 : 48; }
constexpr int kMaxReachB = 48;
constexpr int kStageU = 10;            // float4 window loads per thread, all in flight: C·Wx/4 ≤ kStageU·threads

struct RbPairMulti {
  RbPairArgs c[kWinMulti];
};

// tools/probe/pairprobe -DPH_PAIR_TRACE: per wave and tile, s_memtime stamps at the phase boundaries (written to *ph_pair_trace)
#ifdef PH_PAIR_TRACE
__device__ unsigned long long* ph_pair_trace_buf;
#define PH_STAMP(k) do { if (lane == 0 && round < 4 && ph_pair_trace_buf && blockIdx.x < 512) ph_pair_trace_buf[(((size_t)blockIdx.x * (MT * kWN) + wave) * 4 + round) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_STAMP(k) do { } while (0)
#endif

// LeakyReLU for 0 < α < 1 is max(v, αv). It is applied where a value is WRITTEN to LDS (staging, x1 epilogue), never in the
// K-loops: on gfx950 a vector instruction does not overlap the matrix pipe of its SIMD — tools/probe/issueprobe, r2o: every
// VALU op next to an MFMA costs ≈ 4 of the pipe's cycles (8 per MFMA: 150 → 107 TFLOP/s), scalar ops are free — so the loops
// carry one v_add per ds_read2 and nothing else. (An inline-asm v_max_f32 in place of fmaxf's canonicalise + max gave wrong
// sums next to the MFMAs: pairprobe variants 0/2, r2m.)
__device__ __forceinline__ float lrelu_max(float v, float alpha) { return fmaxf(v, v * alpha); }

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, float v, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}

// Tile schedule of a launch: the (pair, batch item, column block) tiles in ONE list, heaviest pair first; block b of G walks
// it in snake order (b, 2G−1−b, 2G+b, …) so that every block gets a similar mix of heavy and light tiles.
struct PairSched {
  int ntx[kWinMulti];  // column blocks per row, by position in `order`
  int cnt[kWinMulti];  // tiles of that pair (ntx · batch)
  int total, batch, order;
};

// MT row tiles (C = 32·MT); the block has MT·kWN waves: wave = (wm, wn), one row tile and kNTW column tiles each.
// PERSISTENT: a block loops over its tiles; while it multiplies tile i, the window of tile i+1 is already on its way from
// memory into registers (issued at the start of conv a, committed to LDS after conv b), and the weight ring of the next conv
// is always started before the epilogue in front of it.
template <int MT>
__global__ __launch_bounds__(MT * kWN * 64, 2) void rb_pair_kernel(const RbPairMulti multi, const PairSched sch, const int Wx, const int W1,
                                                               const unsigned inv_w4) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BT = MT * kWN * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave % MT, wn = wave / MT;
  const int r = lane & 31, h = lane >> 5;
  constexpr int C = 32 * MT, C2 = C >> 1;
  float* xs = lds;                 // lrelu(x) window [C][Wx]; after conv a: raw x1 [C][W1] (ResBlock2's residual)
  float* x1s = lds + C * Wx + 4;   // lrelu(x1) [C][W1]   (+4: the staging dump slot)
  // biases of every conv of the launch, [pair position z][a | b][C]: accumulators START at the bias (bias first, like
  // CPUBackend.conv1d), read here with four ds_read_b128 per conv. (r2r trace: taking them through scalar loads inside the
  // epilogues serialised every element behind an lgkmcnt(0) — 12 k + 9 k cycles per tile, 30 % of it.)
  float* biasS = x1s + C * W1;
  for (int i = threadIdx.x; i < kWinMulti * 2 * C; i += BT) {
    const int z = i / (2 * C), rem = i - z * 2 * C;
    const RbPairArgs& p = multi.c[(sch.order >> (4 * z)) & 15];
    biasS[i] = rem < C ? p.ba[rem] : p.bb[rem - C];
  }
  const int rowlane = wm * 32 + 4 * h;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int G = (int)gridDim.x, bid = (int)blockIdx.x;

  struct Tile {  // wave-uniform description of one tile
    const float *x, *wa4, *wb4;
    float* y;
    int bias_off;  // of conv a in biasS; conv b: + C
    int Ka, dila, Kb, dilb, res_a, res_b_x, L, Lv, pa, pb, halo, c0, ga, shift, Sa, Spa, Sb, Spb, ntb;
    float alpha;
  };
  auto tile_index = [&](int round) {
    const int idx = round * G + ((round & 1) ? G - 1 - bid : bid);
    return idx < sch.total ? idx : -1;
  };
  auto setup = [&](int idx) {
    int z = 0, rem = idx;
    if (rem >= sch.cnt[0]) { rem -= sch.cnt[0]; z = 1; }
    if (z == 1 && rem >= sch.cnt[1]) { rem -= sch.cnt[1]; z = 2; }
    const RbPairArgs& p = multi.c[(sch.order >> (4 * z)) & 15];
    const int ntx = sch.ntx[z];
    const int n = __builtin_amdgcn_readfirstlane(rem / ntx);  // the division runs on the vector ALU: tell the compiler the result is uniform
    const int bx = rem - n * ntx;
    Tile t;
    t.x = p.x + (int64_t)n * C * p.L;
    t.y = p.y + (int64_t)n * C * p.L;
    t.wa4 = p.wa4; t.wb4 = p.wb4; t.bias_off = z * 2 * C;
    t.Ka = p.Ka; t.dila = p.dila; t.Kb = p.Kb; t.dilb = p.dilb; t.res_a = p.res_a; t.res_b_x = p.res_b_x; t.L = p.L; t.alpha = p.alpha;
    t.Lv = p.len_ptr ? min(p.len_ptr[n] * p.len_mul, p.L) : p.L;
    t.pa = (p.Ka - 1) * p.dila / 2; t.pb = (p.Kb - 1) * p.dilb / 2;
    t.halo = halo_of(t.pb);
    const int ncb = kColsA - 2 * t.halo;            // output columns per tile: 224 or 160
    t.c0 = bx * ncb;                                 // first output column
    const int g0 = t.c0 - t.halo - t.pa;             // input position of window column `shift`
    t.ga = g0 & ~3;
    t.shift = g0 - t.ga;
    t.Sa = p.Ka * C2; t.Spa = (t.Sa + kStepPad - 1) / kStepPad * kStepPad;
    t.Sb = p.Kb * C2; t.Spb = (t.Sb + kStepPad - 1) / kStepPad * kStepPad;
    t.ntb = min(max((ncb >> 5) - wn * kNTW, 0), kNTW);  // this wave's conv-b tiles
    return t;
  };

  // ---- weight ring
  float4 a[kRA];
  const char* wa = nullptr;
  auto load_a = [&](int slot, int ahead) { a[slot] = *(const float4*)(wa + ahead * 1024 + lane16); };
  auto ring_start = [&](const float* w4, int Sp) {
    wa = (const char*)w4 + (int64_t)wm * Sp * 256;
#pragma unroll
    for (int d = 0; d < kRA - 1; d++) load_a(d, d);
  };

  // ---- window staging in two halves: issue (global → registers, ALL of a thread's loads in flight), commit (→ LDS with
  // LeakyReLU and the zero padding). flat float4 index → (row, i4) by a multiply-high (exact for these sizes)
  float4 pre[kStageU];
  const int W4 = Wx >> 2, total4 = C * W4;
  auto stage_issue = [&](const Tile& t, const int tx) {
#pragma unroll
    for (int u = 0; u < kStageU; u++) {
      const int ic = min(tx + u * BT, total4 - 1);
      const int row = (int)__umulhi((unsigned)ic, inv_w4);
      const int pos = t.ga + 4 * (ic - row * W4);
      pre[u] = *(const float4*)(t.x + (int64_t)row * t.L + ((pos >= 0 && pos < t.L) ? pos : 0));
    }
  };
  auto stage_commit = [&](const Tile& t, const int tx) {
#pragma unroll
    for (int u = 0; u < kStageU; u++) {
      const int i = tx + u * BT;
      const int ic = min(i, total4 - 1);
      const int row = (int)__umulhi((unsigned)ic, inv_w4);
      const int i4 = ic - row * W4;
      const int pos = t.ga + 4 * i4;
      const int nv = (pos >= 0 && pos < t.L) ? t.Lv - pos : 0;
      float4 v = pre[u];
      v.x = nv > 0 ? lrelu_max(v.x, t.alpha) : 0.0f; v.y = nv > 1 ? lrelu_max(v.y, t.alpha) : 0.0f;
      v.z = nv > 2 ? lrelu_max(v.z, t.alpha) : 0.0f; v.w = nv > 3 ? lrelu_max(v.w, t.alpha) : 0.0f;
      *(float4*)(xs + (i < total4 ? row * Wx + 4 * i4 : C * Wx)) = v;
    }
  };

  // ---- one conv over this wave's column tiles: B straight from an LDS image [C][Wrow], A through the ring
  f32x16 acc[kNTW];
  auto run_conv = [&](auto nt_tag, const float* img, const int Wrow, const int S, const int Sp, const int dil, const int col0, const int bias_off) {
    constexpr int NT = decltype(nt_tag)::value;
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {  // register q of lane (r,h): row (q&3) + 8·(q>>2) + 4·h
      const float4 bv = *(const float4*)(biasS + bias_off + rowlane + 8 * g4);
#pragma unroll
      for (int j = 0; j < kNTW; j++) { acc[j][4 * g4] = bv.x; acc[j][4 * g4 + 1] = bv.y; acc[j][4 * g4 + 2] = bv.z; acc[j][4 * g4 + 3] = bv.w; }
    }
    const int lbase = h * Wrow + col0 + r;
    int sidx = 0, c_n = 0, left = S - 1;  // the index stops at the last real step (padded steps carry zero weights)
    const int wrap_delta = dil - 2 * Wrow * (C2 - 1);
    float b[2][NT][4];
    auto read_b4 = [&](int slot) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
#pragma unroll
        for (int j = 0; j < NT; j++) b[slot][j][e] = img[lbase + sidx + 32 * j];
        c_n++;
        const bool wrap = c_n == C2;
        c_n = wrap ? 0 : c_n;
        const int delta = wrap ? wrap_delta : 2 * Wrow;
        sidx += left > 0 ? delta : 0;
        left--;
      }
    };
    auto group = [&](int u) {
      load_a((u + kRA - 1) % kRA, kRA - 1 + u);
      read_b4((u + 1) & 1);
      // keep the ring load and the NEXT group's LDS reads here, ahead of this group's MFMAs: left alone the scheduler sinks
      // both next to their first use and every group then waits out a full LDS / L2 round trip (r2k ISA)
      __builtin_amdgcn_sched_barrier(0);
      const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b[u & 1][j][e], acc[j], 0, 0, 0);
    };
    read_b4(0);
    const int Gr = Sp >> 2, full = Gr / kRA;
    // The first ring turn is straight-line code: loads issued just before this conv (the next tile's window, the residual)
    // are younger than the ring's first groups, and only straight-line code lets the compiler count them precisely — inside
    // the loop every wait is vmcnt(7), which on the first turn would wait for those fresh loads at once.
    if (full > 0) {
#pragma unroll
      for (int u = 0; u < kRA; u++) group(u);
      wa += kRA * 1024;
    }
    for (int g = 1; g < full; g++) {
#pragma unroll
      for (int u = 0; u < kRA; u++) group(u);
      wa += kRA * 1024;
    }
    const int rem = Gr - full * kRA;
#pragma unroll
    for (int u = 0; u < kRA - 1; u++)
      if (u < rem) group(u);
  };

  // ================= cold start: first tile's ring and window
  Tile cur = setup(tile_index(0));  // the grid never exceeds the tile count
  ring_start(cur.wa4, cur.Spa);
  stage_issue(cur, (int)threadIdx.x);
  stage_commit(cur, (int)threadIdx.x);
  __syncthreads();

  for (int round = 0;; round++) {
    const int nidx = tile_index(round + 1);
    const bool has_next = nidx >= 0;          // block-uniform
    const Tile nxt = has_next ? setup(nidx) : cur;
    // Lane-dependent offsets (staging rows, epilogue addresses) are the same for every tile; left alone the compiler hoists
    // all ≈ 90 of them out of the tile loop and the kernel no longer fits 256 registers. Opaque copies of the lane ids per
    // tile keep them where they are used.
    int tx = (int)threadIdx.x, rl = rowlane, rr = r;
    asm volatile("" : "+v"(tx), "+v"(rl), "+v"(rr));
    // the RAW x this wave adds as a residual (ResBlock2: to x1 on its conv-a tiles; ResBlock1: to y on its conv-b tiles) is
    // requested right after the conv in front of the epilogue that uses it. Buffer loads: lane part of the address in one
    // register, row part a scalar offset; out of range ⇒ 0.
    float resx[kNTW][16];
    auto load_res = [&](int gres0) {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)cur.x, 0, C * cur.L * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < kNTW; j++) {
        const int g = gres0 + (wn * kNTW + j) * 32 + rr;
        const int voff = (g >= 0 && g < cur.L) ? (rl * cur.L + g) * 4 : -4;
#pragma unroll
        for (int q = 0; q < 16; q++) resx[j][q] = bload(rx, voff, ((q & 3) + 8 * (q >> 2)) * cur.L * 4);
      }
    };

    PH_STAMP(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): ring a landed (issued before the previous epilogue): clean state
    PH_STAMP(1);
    if (has_next) stage_issue(nxt, tx);  // next tile's window: in flight during conv a, held in registers through conv b

    // bucketed / ragged batches: a tile at or past the item's true length is skipped — its barriers and the next tile's
    // prefetch / commit stay (block-uniform)
    const bool live = cur.c0 < cur.Lv;
    // ======== conv a: x1 columns [c0 − halo, c0 − halo + 256) = tiles wn·2, wn·2 + 1 of the tile's 8
    if (live) run_conv(std::integral_constant<int, kNTW>{}, xs, Wx, cur.Sa, cur.Spa, cur.dila, cur.shift + wn * kNTW * 32, cur.bias_off);
    PH_STAMP(2);
    if (live) ring_start(cur.wb4, cur.Spb);  // conv b's ring: its round trip hides behind the x1 epilogue and the barriers
    if (live && cur.res_a) load_res(cur.c0 - cur.halo);
    __syncthreads();               // every wave is done reading the x window: its memory now takes the raw x1
    PH_STAMP(3);

    if (live) {  // x1 = [x +] acc (bias inside), zero outside [0, len): lrelu(x1) → x1s (conv b's operand), raw x1 → xs region (ResBlock2's
       // residual). register q of lane (r,h): row (q&3) + 8·(q>>2) + 4·h, column r
#pragma unroll
      for (int j = 0; j < kNTW; j++) {
        const int colw = (wn * kNTW + j) * 32 + rr;       // x1 column
        const int g = cur.c0 - cur.halo + colw;           // its position in the row
        const bool in = g >= 0 && g < cur.Lv;
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int rq = (q & 3) + 8 * (q >> 2);
          const int o = (rl + rq) * W1 + colw;
          float v = acc[j][q];
          if (cur.res_a) v += resx[j][q];
          v = in ? v : 0.0f;
          x1s[o] = lrelu_max(v, cur.alpha);
          if (!cur.res_b_x) xs[o] = v;
        }
      }
    }
    __syncthreads();

    PH_STAMP(4);
    // ======== conv b: the tile's output tiles, two per wave column (the last ones get one or none)
    if (live && cur.ntb > 0) {
      const int col0b = cur.halo - cur.pb + wn * kNTW * 32;
      __builtin_amdgcn_s_waitcnt(0x0F70);
      if (cur.ntb == 2) run_conv(std::integral_constant<int, 2>{}, x1s, W1, cur.Sb, cur.Spb, cur.dilb, col0b, cur.bias_off + C);
      else run_conv(std::integral_constant<int, 1>{}, x1s, W1, cur.Sb, cur.Spb, cur.dilb, col0b, cur.bias_off + C);
    }
    PH_STAMP(5);
    if (has_next) ring_start(nxt.wa4, nxt.Spa);  // next tile's conv a ring: lands during the epilogue and the barriers
    if (live && cur.res_b_x && cur.ntb > 0) load_res(cur.c0);
    if (live && cur.ntb > 0) {  // y = acc (bias inside) + (x | x1) → global; 2 rows × 32 consecutive columns per store instruction
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)cur.y, 0, C * cur.L * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < kNTW; j++) {
        if (j >= cur.ntb) break;
        const int colo = (wn * kNTW + j) * 32 + rr;        // output column within the tile
        const int g = cur.c0 + colo;
        const int voff = g < cur.L ? (rl * cur.L + g) * 4 : -4;  // −4: out of range ⇒ the store is dropped
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int rq = (q & 3) + 8 * (q >> 2);
          float v = acc[j][q];
          v += cur.res_b_x ? resx[j][q] : xs[(rl + rq) * W1 + cur.halo + colo];
          bstore(ry, v, voff, rq * cur.L * 4);
        }
      }
    }
    PH_STAMP(6);
    if (!has_next) break;
    __syncthreads();      // every wave is done with both LDS images
    stage_commit(nxt, tx);
    __syncthreads();
    PH_STAMP(7);
    cur = nxt;
  }
}

struct PairGeom {
  int Wx, W1;
  size_t lds;
};

// Row lengths: Wx covers shift (≤ 3) + 256 + 2·pa_max, W1 the 256 x1 columns (tiles a wave column does not own are neither
// read nor multiplied); both rounded to ≡ 32 (mod 64) floats when that fits
// (lanes 32–63 read the next channel row: a row stride of 32 mod 64 banks keeps the two halves on disjoint banks).
PairGeom pair_geom(int C, int pa_max, int pb_max) {
  auto pad = [](int w, bool odd32) {
    w = (w + 3) & ~3;
    if (odd32) while ((w & 63) != 32) w += 4;
    return w;
  };
  PairGeom g;
  for (int pass = 0; pass < 2; pass++) {
    g.Wx = pad(kColsA + 2 * pa_max + 3, pass == 0);
    g.W1 = pad(kColsA, pass == 0);
    (void)pb_max;
    g.lds = ((size_t)C * g.Wx + 4 + (size_t)C * g.W1 + (size_t)kWinMulti * 2 * C) * sizeof(float);
    if (g.lds <= 160 * 1024 && (C * g.Wx) / 4 <= kStageU * (C / 32) * kWN * 64) break;
  }
  return g;
}

template <int MT>
void launch_pair_inst(piper_hip_ctx* ctx, hipStream_t s, const RbPairMulti& m, const PairSched& sch, const PairGeom& g) {
  static bool raised[kMaxDevices] = {};
  if (g.lds > 64 * 1024 && lds_optin_needed(raised))
    (void)hipFuncSetAttribute((const void*)rb_pair_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const unsigned inv = (unsigned)(0x100000000ull / (unsigned)(g.Wx >> 2)) + 1u;
  const int per_cu = std::max(1, std::min(2, (int)((size_t)160 * 1024 / g.lds)));  // resident blocks per CU (LDS; ≤ 256 VGPRs ⇒ ≤ 2 waves/SIMD)
  const int grid = std::min(sch.total, per_cu * ctx->num_cus);
  hipLaunchKernelGGL((rb_pair_kernel<MT>), dim3(grid), dim3(MT * kWN * 64), g.lds, s, m, sch, g.Wx, g.W1, inv);
}

}  // namespace

#ifdef PH_PAIR_TRACE
void rb_pair_set_trace(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(ph_pair_trace_buf), &buf, sizeof buf); }
#endif

bool rb_pair_eligible(int C, int Ka, int dila, int Kb, int dilb, int L) {
  if (C != 32 && C != 64) return false;
  if (Ka < 1 || Kb < 1 || !(Ka & 1) || !(Kb & 1) || dila < 1 || dilb < 1) return false;
  if (L < 4 || (L & 3)) return false;
  const int pa = (Ka - 1) * dila / 2, pb = (Kb - 1) * dilb / 2;
  if (pb > kMaxReachB) return false;
  if ((int64_t)C * (kColsA + 2 * pa + 70) >= (1 << 20)) return false;  // multiply-high row split: exact far beyond this
  const PairGeom g = pair_geom(C, pa, pb);
  return g.lds <= 160 * 1024 && (C * g.Wx) / 4 <= kStageU * (C / 32) * kWN * 64;
}

int launch_rb_pair_multi(piper_hip_ctx* ctx, hipStream_t s, const RbPairArgs* pairs, int count) {
  if (count < 1 || count > kWinMulti) PH_FAIL(PIPER_HIP_ERR_ARG, "rb_pair: %d pairs in one launch (1..%d)", count, kWinMulti);
  const RbPairArgs& a = pairs[0];
  if (a.N <= 0 || a.L <= 0) return PIPER_HIP_OK;
  int pa_max = 0, pb_max = 0;
  int idx[kWinMulti] = {0, 1, 2};
  for (int i = 0; i < count; i++) {
    const RbPairArgs& b = pairs[i];
    if (b.N != a.N || b.C != a.C || b.L != a.L) PH_FAIL(PIPER_HIP_ERR_SHAPE, "rb_pair: pairs of one launch must share N, C and L");
    if (!rb_pair_eligible(b.C, b.Ka, b.dila, b.Kb, b.dilb, b.L)) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rb_pair: geometry not covered (C=%d K=%d,%d d=%d,%d L=%d)", b.C, b.Ka, b.Kb, b.dila, b.dilb, b.L);
    if (!(b.alpha > 0.0f && b.alpha < 1.0f)) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rb_pair: LeakyReLU slope %g outside (0,1)", (double)b.alpha);
    if (!b.x || !b.y || !b.wa4 || !b.wb4 || !b.ba || !b.bb) PH_FAIL(PIPER_HIP_ERR_ARG, "rb_pair: null operand");
    pa_max = std::max(pa_max, (b.Ka - 1) * b.dila / 2);
    pb_max = std::max(pb_max, (b.Kb - 1) * b.dilb / 2);
  }
  const PairGeom g = pair_geom(a.C, pa_max, pb_max);
  if (g.lds > 160 * 1024 || (a.C * g.Wx) / 4 > kStageU * (a.C / 32) * kWN * 64)
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rb_pair: window of %zu bytes (row %d) exceeds LDS / the staging registers", g.lds, g.Wx);
  std::sort(idx, idx + count, [&](int l, int r2) { return pairs[l].Ka + pairs[l].Kb > pairs[r2].Ka + pairs[r2].Kb; });
  PairSched sch = {};
  sch.batch = a.N;
  for (int i = 0; i < count; i++) {
    const RbPairArgs& b = pairs[idx[i]];
    sch.order |= idx[i] << (4 * i);
    sch.ntx[i] = (int)ceil_div(a.L, kColsA - 2 * halo_of((b.Kb - 1) * b.dilb / 2));
    sch.cnt[i] = sch.ntx[i] * a.N;
    sch.total += sch.cnt[i];
  }
  for (int i = count; i < kWinMulti; i++) { sch.ntx[i] = 1; sch.cnt[i] = 0; }
  RbPairMulti m;
  for (int i = 0; i < kWinMulti; i++) m.c[i] = pairs[i < count ? i : 0];
  if (a.C == 32) launch_pair_inst<1>(ctx, s, m, sch, g);
  else launch_pair_inst<2>(ctx, s, m, sch, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rb_pair launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(rb_pair, (rb_pair_kernel<1>)); } }
