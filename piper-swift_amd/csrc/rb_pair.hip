// rb_pair.hip — two chained ResBlock convs of the HiFi-GAN generator in ONE kernel, the intermediate kept in LDS.
//
// Reference: the generator's ResBlocks are chains of "x ← x + conv_d(lrelu(x))" (ResBlock2, Piper medium) or
// "x ← x + conv_1(lrelu(conv_d(lrelu(x))))" (ResBlock1, Piper high) — GraphExecutor dispatches every LeakyRelu / Conv / Add of
// them on its own (PiperMetalGraph.swift: the `dec.resblocks.*` nodes; conv1d.metal:28-71 is the conv). conv_win_kernel runs
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

// MT row tiles (C = 32·MT); the block has MT·kWN waves: wave = (wm, wn), one row tile and kNTW column tiles each.
template <int MT>
__global__ __launch_bounds__(MT * kWN * 64) void rb_pair_kernel(const RbPairMulti multi, const int batch, const int order, const int Wx, const int W1,
                                                               const unsigned inv_w4) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BT = MT * kWN * 64;
  const int jz = blockIdx.y / batch;
  const RbPairArgs& p = multi.c[(order >> (4 * jz)) & 15];  // heaviest pair first: blocks are handed out in grid order
  const int n = blockIdx.y - jz * batch;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave % MT, wn = wave / MT;
  const int r = lane & 31, h = lane >> 5;
  const int C = 32 * MT, C2 = C >> 1;
  const int pa = (p.Ka - 1) * p.dila / 2, pb = (p.Kb - 1) * p.dilb / 2;
  const int halo = halo_of(pb), ncb = kColsA - 2 * halo;  // output columns per block: 224 or 160
  const int c0 = blockIdx.x * ncb;                   // first output column of the block
  if (c0 >= p.L) return;                             // the grid is sized for the launch's narrowest blocks
  const int g0 = c0 - halo - pa;                     // input position of window column `shift`
  const int ga = g0 & ~3;
  const int shift = g0 - ga;
  float* xs = lds;                                   // lrelu(x) window [C][Wx]; after conv a: raw x1 [C][W1] (ResBlock2's residual)
  float* x1s = lds + C * Wx + 4;                     // lrelu(x1) [C][W1]   (+4: the staging dump slot)
  const int Lv = p.len_ptr ? min(p.len_ptr[n] * p.len_mul, p.L) : p.L;
  const float alpha = p.alpha;
  const int ntb = min(max((ncb >> 5) - wn * kNTW, 0), kNTW);  // this wave's conv-b tiles (wave-uniform)

  // ---- weight rings: conv a's first groups are requested before the window exists
  const int Sa = p.Ka * C2, Spa = (Sa + kStepPad - 1) / kStepPad * kStepPad;
  const int Sb = p.Kb * C2, Spb = (Sb + kStepPad - 1) / kStepPad * kStepPad;
  const unsigned lane16 = (unsigned)lane * 16u;
  float4 a[kRA];
  const char* wa = (const char*)p.wa4 + (int64_t)wm * Spa * 256;
  auto load_a = [&](int slot, int ahead) { a[slot] = *(const float4*)(wa + ahead * 1024 + lane16); };
#pragma unroll
  for (int d = 0; d < kRA - 1; d++) load_a(d, d);

  // ---- biases of this lane's 16 rows, both convs, and the RAW x this wave adds as a residual (ResBlock2: to x1 on its
  // conv-a tiles; ResBlock1: to y on its conv-b tiles) — requested now, used in the epilogues. Buffer loads: the lane part
  // of the address is one register, the row part a scalar offset; columns left of the row (negative offset) read as 0.
  float biasa[16], biasb[16], resx[kNTW][16];
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * C * p.L), 0, C * p.L * 4, 0x00020000);
  const int rowlane = wm * 32 + 4 * h;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int row = rowlane + (q & 3) + 8 * (q >> 2);
    biasa[q] = p.ba[row];
    biasb[q] = p.bb[row];
  }
  {
    const int gres0 = p.res_a ? c0 - halo : c0;  // position of column 0 of tile 0 of the residual's tile grid
#pragma unroll
    for (int j = 0; j < kNTW; j++) {
      const int g = gres0 + (wn * kNTW + j) * 32 + r;
      const int voff = (g >= 0 && g < p.L) ? (rowlane * p.L + g) * 4 : -4;  // −4: out of range ⇒ 0
#pragma unroll
      for (int q = 0; q < 16; q++) resx[j][q] = bload(rx, voff, ((q & 3) + 8 * (q >> 2)) * p.L * 4);
    }
  }

  // ---- stage lrelu(x): flat float4 index → (row, i4) by a multiply-high (exact for these sizes). ALL of a thread's loads
  // are in flight at once (≤ kStageU·BT float4 per block, host-checked): one memory round trip
  {
    const float* xb = p.x + (int64_t)n * C * p.L;
    const int W4 = Wx >> 2, total = C * W4, dump = C * Wx;
    float4 t[kStageU];
    int dst[kStageU], nv[kStageU];
#pragma unroll
    for (int u = 0; u < kStageU; u++) {
      const int i = (int)threadIdx.x + u * BT;
      const int ic = min(i, total - 1);
      const int row = (int)__umulhi((unsigned)ic, inv_w4);
      const int i4 = ic - row * W4;
      const int pos = ga + 4 * i4;
      const bool inb = pos >= 0 && pos < p.L;
      t[u] = *(const float4*)(xb + (int64_t)row * p.L + (inb ? pos : 0));
      nv[u] = inb ? Lv - pos : 0;
      dst[u] = i < total ? row * Wx + 4 * i4 : dump;
    }
#pragma unroll
    for (int u = 0; u < kStageU; u++) {
      float4 v = t[u];
      v.x = nv[u] > 0 ? lrelu_max(v.x, alpha) : 0.0f; v.y = nv[u] > 1 ? lrelu_max(v.y, alpha) : 0.0f;
      v.z = nv[u] > 2 ? lrelu_max(v.z, alpha) : 0.0f; v.w = nv[u] > 3 ? lrelu_max(v.w, alpha) : 0.0f;
      *(float4*)(xs + dst[u]) = v;
    }
  }
  __syncthreads();

  // ---- one conv over this wave's column tiles: B straight from an LDS image [C][Wrow], A through the ring
  f32x16 acc[kNTW];
  auto run_conv = [&](auto nt_tag, const float* img, const int Wrow, const int S, const int Sp, const int dil, const int col0) {
    constexpr int NT = decltype(nt_tag)::value;
#pragma unroll
    for (int j = 0; j < kNTW; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[j][q] = 0.0f;
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
    const int G = Sp >> 2, full = G / kRA;
    for (int g = 0; g < full; g++) {
#pragma unroll
      for (int u = 0; u < kRA; u++) group(u);
      wa += kRA * 1024;
    }
    const int rem = G - full * kRA;
#pragma unroll
    for (int u = 0; u < kRA - 1; u++)
      if (u < rem) group(u);
  };

  // ======== conv a: x1 columns [c0 − halo, c0 − halo + 256) = tiles wn·2, wn·2 + 1 of the block's 8
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the ring's first groups have landed long ago (clean state for the loop)
  run_conv(std::integral_constant<int, kNTW>{}, xs, Wx, Sa, Spa, p.dila, shift + wn * kNTW * 32);

  // conv b's ring starts now: its round trip hides behind the x1 epilogue and the barriers
  wa = (const char*)p.wb4 + (int64_t)wm * Spb * 256;
#pragma unroll
  for (int d = 0; d < kRA - 1; d++) load_a(d, d);
  __syncthreads();  // every wave is done reading the x window: its memory now takes the raw x1

  {  // x1 = [x +] acc + bias, zero outside [0, len): lrelu(x1) → x1s (conv b's operand), raw x1 → xs region (ResBlock2's
     // residual). register q of lane (r,h): row (q&3) + 8·(q>>2) + 4·h, column r
#pragma unroll
    for (int j = 0; j < kNTW; j++) {
      const int colw = (wn * kNTW + j) * 32 + r;      // x1 column
      const int g = c0 - halo + colw;                  // its position in the row
      const bool in = g >= 0 && g < Lv;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int o = (rowlane + (q & 3) + 8 * (q >> 2)) * W1 + colw;
        float v = acc[j][q] + biasa[q];
        if (p.res_a) v += resx[j][q];
        v = in ? v : 0.0f;
        x1s[o] = lrelu_max(v, alpha);
        if (!p.res_b_x) xs[o] = v;
      }
    }
  }
  __syncthreads();

  // ======== conv b: the block's ncb / 32 output tiles, two per wave column (the last ones get one or none)
  if (ntb == 0) return;
  const int col0b = halo - pb + wn * kNTW * 32;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if (ntb == 2) run_conv(std::integral_constant<int, 2>{}, x1s, W1, Sb, Spb, p.dilb, col0b);
  else run_conv(std::integral_constant<int, 1>{}, x1s, W1, Sb, Spb, p.dilb, col0b);

  {  // y = acc + bias + (x | x1) → global; 2 rows × 32 consecutive columns per store instruction
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (int64_t)n * C * p.L), 0, C * p.L * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < kNTW; j++) {
      if (j >= ntb) break;
      const int colo = (wn * kNTW + j) * 32 + r;       // output column within the block
      const int g = c0 + colo;
      const int voff = g < p.L ? (rowlane * p.L + g) * 4 : -4;  // −4: out of range ⇒ the store is dropped
#pragma unroll
      for (int q = 0; q < 16; q++) {
        float v = acc[j][q] + biasb[q];
        v += p.res_b_x ? resx[j][q] : xs[(rowlane + (q & 3) + 8 * (q >> 2)) * W1 + halo + colo];
        bstore(ry, v, voff, ((q & 3) + 8 * (q >> 2)) * p.L * 4);
      }
    }
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
    g.lds = ((size_t)C * g.Wx + 4 + (size_t)C * g.W1) * sizeof(float);
    if (g.lds <= 160 * 1024 && (C * g.Wx) / 4 <= kStageU * (C / 32) * kWN * 64) break;
  }
  return g;
}

template <int MT>
void launch_pair_inst(hipStream_t s, const RbPairMulti& m, int batch, int order, const PairGeom& g, dim3 grid) {
  static bool raised[kMaxDevices] = {};
  if (g.lds > 64 * 1024 && lds_optin_needed(raised))
    (void)hipFuncSetAttribute((const void*)rb_pair_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const unsigned inv = (unsigned)(0x100000000ull / (unsigned)(g.Wx >> 2)) + 1u;
  hipLaunchKernelGGL((rb_pair_kernel<MT>), grid, dim3(MT * kWN * 64), g.lds, s, m, batch, order, g.Wx, g.W1, inv);
}

}  // namespace

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
  int order = 0;
  for (int i = 0; i < count; i++) order |= idx[i] << (4 * i);
  RbPairMulti m;
  for (int i = 0; i < kWinMulti; i++) m.c[i] = pairs[i < count ? i : 0];
  const dim3 grid((unsigned)ceil_div(a.L, kColsA - 2 * halo_of(pb_max)), (unsigned)(a.N * count));
  if (a.C == 32) launch_pair_inst<1>(s, m, a.N, order, g, grid);
  else launch_pair_inst<2>(s, m, a.N, order, g, grid);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rb_pair launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace ph
