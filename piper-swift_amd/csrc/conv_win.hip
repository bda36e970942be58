// conv_win.hip — fp32 Conv1d / ConvTranspose1d for long rows: input window in LDS, exact-fp32 MFMA (32x32x2).
//
// Role: the ResBlock convs and ConvTranspose upsamplers of the HiFi-GAN generator (conv1d.metal:28-71, 97-142 in the
// reference) in the fp32 voice path. conv_stream_kernel feeds every MFMA's B operand from global memory with the
// LeakyReLU prologue in the loop; on rows of 2 688 … 86 016 steps that left the matrix pipe ≈ 35 % busy. Here
//   1. a block (4 waves) stages its whole input window — all Cin rows × (columns + dilation reach) — into LDS once,
//      as aligned float4 loads (rows are multiples of 4 long, so a float4 is entirely inside or outside [0, Lin): the
//      zero padding costs one select), LeakyReLU applied on the way in, 16 loads in flight per lane;
//   2. the K-loop is one ds_read_b32 + one MFMA per (channel pair, tap) step; weight fragments are float4 = 4 steps per
//      lane, streamed from L2 through an 8-deep ring (32 steps ahead) that is started before the staging round trip;
//      the window index advances by scalar selects (no branches, no multiplies);
//   3. when the problem has fewer 32×32 tiles than the chip has SIMDs, the block's waves split the contraction (KS = 2
//      or 4 contiguous step ranges) and reduce in fixed order through LDS — deterministic, like conv_stream_kernel;
//   4. epilogue: bias, residual, MRF mean, LeakyReLU, ConvTranspose scatter — loads first, masked stores after.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_win.h"

namespace ph {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBT = 256;
constexpr int kRA = 8;            // float4 weight groups in flight (32 steps)
constexpr int kStepPad = 16;      // steps are padded to a multiple of 16 so every K-split (1, 2, 4) is whole float4 groups
constexpr int kTailFloats = 4096; // readable floats behind the image: the ring runs up to 14 groups past a wave's range

__device__ __forceinline__ float lrelu1(float v, float alpha) { return v >= 0.0f ? v : v * alpha; }

inline int padded_steps_win(int Cin, int taps) { return (taps * (Cin / 2) + kStepPad - 1) / kStepPad * kStepPad; }

// element e of lane l of group s4 of row tile mt: step = 4·s4 + e = tap·(Cin/2) + c2 → w[row mt·32 + (l&31)][2·c2 + (l>>5)][tap]
__global__ __launch_bounds__(kBT) void pack_conv_win_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int S, int Sp,
                                                           float* __restrict__ out, int64_t total) {
  const int C2 = Cin >> 1;
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBT) {
    const int e = (int)(i & 3), l = (int)((i >> 2) & 63);
    const int64_t g = i >> 8;  // (mt, s4)
    const int s4 = (int)(g % (Sp >> 2)), mt = (int)(g / (Sp >> 2));
    const int step = 4 * s4 + e;
    float v = 0.0f;
    const int row = mt * 32 + (l & 31);
    if (step < S && row < Cout) {
      const int tap = step / C2, c2 = step - tap * C2;
      v = w[((int64_t)row * Cin + 2 * c2 + (l >> 5)) * K + tap];
    }
    out[i] = v;
  }
}

// ConvTranspose: GEMM row R = ρ·Cout + co, tap j ⇒ w[ci][co][(ρ+pad) mod s + s·j]
__global__ __launch_bounds__(kBT) void pack_convt_win_kernel(const float* __restrict__ w, int Cin, int Cout, int K, int stride, int pad,
                                                            int S, int Sp, float* __restrict__ out, int64_t total) {
  const int C2 = Cin >> 1;
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBT) {
    const int e = (int)(i & 3), l = (int)((i >> 2) & 63);
    const int64_t g = i >> 8;
    const int s4 = (int)(g % (Sp >> 2)), mt = (int)(g / (Sp >> 2));
    const int step = 4 * s4 + e;
    float v = 0.0f;
    const int R = mt * 32 + (l & 31);
    const int rho = R / Cout, co = R - rho * Cout;
    if (step < S && rho < stride) {
      const int j = step / C2, c2 = step - j * C2;
      v = w[((int64_t)(2 * c2 + (l >> 5)) * Cout + co) * K + (rho + pad) % stride + stride * j];
    }
    out[i] = v;
  }
}

// tools/probe/winprobe -DPH_WIN_TRACE: s_memtime stamps at the phase boundaries (per wave of the first 1024 blocks)
#ifdef PH_WIN_TRACE
__device__ unsigned long long* ph_win_trace_buf;
#define PH_WSTAMP(k) do { const int bl_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); if ((threadIdx.x & 63) == 0 && ph_win_trace_buf && bl_ < 1024) ph_win_trace_buf[((size_t)bl_ * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_WSTAMP(k) do { } while (0)
#endif

// WM waves along rows × WN along columns × KS along the contraction; WM·WN·KS = 4 or 8 waves per block. One 32×32 tile per wave.
// The 8-wave block (round 3: 4 row tiles × 2 contraction slices of one column tile) is for plain convs with few tiles and a wide window
// (stage 0 of the generator: 128 → 128 channels on 2 688 columns, 1 008 tiles): with 4 waves a block was ONE tile split four ways and
// staged the whole 128-row window for itself — four times per column tile. Sharing the window is worth 1.4 … 3.6 µs of 27 … 30 there
// (r3 A/B, tools/probe/gen_positions.py); 16 waves gave the same, and on the ConvTransposes (two taps: tiny windows) both are slower.
// What bounds these launches is not the window: ≈ 10 µs of MFMA time when perfectly spread + one un-overlapped prologue / staging /
// epilogue per block (every block of the single round is in the same phase at the same time).
struct ConvWinMulti {
  ConvWinArgs c[kWinMulti];
};

template <int WM, int WN, int KS>
__global__ __launch_bounds__(64 * WM * WN * KS) void conv_win_kernel(const ConvWinMulti multi, int batch, int xcd_rows) {
  constexpr int NW = WM * WN * KS, BT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float win[];  // [Cin][Wp] + dump float4; reused for the K-split reduction
  const ConvWinArgs& p = multi.c[blockIdx.z / batch];  // wave-uniform: which of the launch's convs this block works on
  // Workgroups go to the 8 XCDs round robin by their linear id, and each XCD has its own L2. With the column tile as the fastest grid
  // index every XCD ends up multiplying every row tile, i.e. pulls the WHOLE weight image through its L2 (the stage-0 ConvTranspose: 8 × 2 MB
  // for 0.35 GFLOP). xcd_rows (launcher: set when the weights outweigh the activations and gridDim.y % 8 == 0) renumbers the blocks so
  // that row-tile group y runs on XCD y mod 8: each L2 then sees one eighth of the weights and all of the (smaller) activations.
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_rows) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int q = lin >> 3;
    const int yh = __builtin_amdgcn_readfirstlane(q / (int)gridDim.x);
    bx = q - yh * (int)gridDim.x;
    by = (lin & 7) + 8 * yh;
  }
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
  constexpr int NBC = WN * 32;
  PH_WSTAMP(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ks = wave / (WM * WN), wt = wave % (WM * WN);
  const int wm = wt % WM, wn = wt / WM;
  const int r = lane & 31, h = lane >> 5;
  const bool ct = p.ct_stride > 0;
  const int taps = ct ? p.K / p.ct_stride : p.K;
  const int C2 = p.Cin >> 1;
  const int n = blockIdx.z % batch;
  const int nb0 = bx * NBC;
  // bucketed / ragged batches: every column of this block at or past the item's true length ⇒ nothing anyone reads (block-uniform)
  if (p.len_ptr && nb0 >= min(p.len_ptr[n] * p.len_mul, p.Lin)) return;
  const int mt = by * WM + wm;
  const int off_min = ct ? -(taps - 1) : -p.padL;
  const int off_max = ct ? (p.ct_stride - 1 + p.ct_pad) / p.ct_stride : (p.K - 1) * p.dil - p.padL;
  const int W = NBC + off_max - off_min;
  const int g0 = nb0 + off_min;          // input position of window column 0
  const int ga = g0 & ~3;                // staged from the aligned position below it
  const int shift = g0 - ga;
  const int Wp = (W + 3 + 3) & ~3;       // LDS row length (floats), holds shift + W
  const int S = taps * C2;
  const int Sp = (S + kStepPad - 1) / kStepPad * kStepPad;
  const int Sw = Sp / KS;                // steps of this wave: [ks·Sw, (ks+1)·Sw)
  const int s_begin = ks * Sw;

  // ---- weight ring: uniform base + 32-bit lane offset (saddr loads); started before the window exists
  const char* wa = (const char*)p.w4 + ((int64_t)mt * Sp + s_begin) * 256;
  const unsigned lane16 = (unsigned)lane * 16u;
  float4 a[kRA];
  auto load_a = [&](int slot, int ahead) { a[slot] = *(const float4*)(wa + ahead * 1024 + lane16); };
#pragma unroll
  for (int d = 0; d < kRA - 1; d++) load_a(d, d);

  // ---- stage the window: rows by wave, 256 positions (64 lanes × float4) per wave instruction
  auto stage = [&](auto avg_tag) {
    constexpr bool AVG = decltype(avg_tag)::value;
    constexpr int kStage = (AVG ? 5 : 16) * 4 / NW + (AVG && NW > 4 ? 1 : 0);  // float4 loads in flight per lane: 16, or 5 × 3 sources (4 waves)
    const int64_t xoff = (int64_t)n * p.Cin * p.Lin;
    const float* xb = p.x + xoff;
    const float* xb2 = AVG ? p.x2 + xoff : nullptr;
    const float* xb3 = AVG ? p.x3 + xoff : nullptr;
    const int W4 = Wp >> 2;              // float4s per row
    const int total4 = p.Cin * W4;
    const int dump = p.Cin * Wp;         // slots past the window store here (keeps the loads unconditional)
    const int Lv = p.len_ptr ? min(p.len_ptr[n] * p.len_mul, p.Lin) : p.Lin;  // true input length of this batch item
    // FLAT slot index over (row, float4 column): slot i ↔ row = i / W4 by a multiply-high (exact: i·W4 < 2^32). The window of a
    // ConvTranspose tile or of a short-dilation conv is 10 … 20 float4 wide — walking it row by row with 64 lanes per row left
    // 10/64 of every load instruction useful and cost 4 … 13 dependent round trips (r3s trace: 23 k … 70 k cycles of staging
    // against 9 k … 22 k of MFMAs); flat, a thread's kStage slots cover the block's window in one or two.
    const unsigned inv = 0xFFFFFFFFu / (unsigned)W4 + 1u;
    for (int base = threadIdx.x; base < total4; base += BT * kStage) {
      float4 t[kStage], t2[AVG ? kStage : 1], t3[AVG ? kStage : 1];
      int dst[kStage];
      int nvalid[kStage];
#pragma unroll
      for (int q = 0; q < kStage; q++) {
        const int i = base + q * BT;
        const int ic = min(i, total4 - 1);
        const int row = (int)__umulhi((unsigned)ic, inv);
        const int i4 = ic - row * W4;
        const int pos = ga + 4 * i4;     // multiple of 4; the row stride is a multiple of 4, the true length need not be
        nvalid[q] = pos < 0 ? 0 : Lv - pos;  // leading components inside [0, Lv) (≥ 4: all of them)
        const int64_t off = (int64_t)row * p.Lin + ((pos >= 0 && pos < p.Lin) ? pos : 0);
        t[q] = *(const float4*)(xb + off);
        if constexpr (AVG) { t2[q] = *(const float4*)(xb2 + off); t3[q] = *(const float4*)(xb3 + off); }
        dst[q] = i < total4 ? row * Wp + 4 * i4 : dump;
      }
#pragma unroll
      for (int q = 0; q < kStage; q++) {
        float4 v = t[q];
        if constexpr (AVG) {  // ((x + x2) + x3) / 3: the association of the graph's Add, Add, Div
          v.x = ((v.x + t2[q].x) + t3[q].x) / 3.0f; v.y = ((v.y + t2[q].y) + t3[q].y) / 3.0f;
          v.z = ((v.z + t2[q].z) + t3[q].z) / 3.0f; v.w = ((v.w + t2[q].w) + t3[q].w) / 3.0f;
        }
        v.x = lrelu1(v.x, p.pro_alpha); v.y = lrelu1(v.y, p.pro_alpha); v.z = lrelu1(v.z, p.pro_alpha); v.w = lrelu1(v.w, p.pro_alpha);
        v.x = nvalid[q] > 0 ? v.x : 0.0f; v.y = nvalid[q] > 1 ? v.y : 0.0f; v.z = nvalid[q] > 2 ? v.z : 0.0f; v.w = nvalid[q] > 3 ? v.w : 0.0f;
        *(float4*)(win + dst[q]) = v;
      }
    }
  };
  PH_WSTAMP(1);
  if (p.x2) stage(std::true_type{});
  else stage(std::false_type{});
  PH_WSTAMP(2);
  __syncthreads();
  PH_WSTAMP(3);

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; q++) acc[q] = 0.0f;

  const int rho = ct ? (mt * 32) / p.Cout : 0;
  const int off0 = ct ? (rho + p.ct_pad) / p.ct_stride : -p.padL;  // window position of tap t: off0 + t·dstep
  const int dstep = ct ? -1 : p.dil;
  const int lbase = h * Wp + shift + wn * 32 + r - off_min + off0;
  // scalar part of the window index, 2·c2·Wp + tap·dstep, starting at this wave's first step
  const int tap0 = s_begin / C2, c20 = s_begin - tap0 * C2;
  int sidx = 2 * c20 * Wp + min(tap0, taps - 1) * dstep;
  int c_n = c20, left = S - 1 - s_begin;  // advances still allowed (≤ 0 for all-padding ranges: index stays valid)
  if (left < 0) sidx = 0;
  const int wrap_delta = dstep - 2 * Wp * (C2 - 1);
  float b[2][4];
  auto read_b4 = [&](int slot) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      b[slot][e] = win[lbase + sidx];
      c_n++;
      const bool wrap = c_n == C2;
      c_n = wrap ? 0 : c_n;
      const int delta = wrap ? wrap_delta : 2 * Wp;
      sidx += left > 0 ? delta : 0;
      left--;
    }
  };
  auto group = [&](int u) {  // u = static ring slot
    load_a((u + kRA - 1) % kRA, kRA - 1 + u);
    // keep the load HERE: left alone, the scheduler clusters the ring's loads next to their first use at the end of the
    // unrolled body and the loop then waits out a full L2 round trip every iteration
    __builtin_amdgcn_sched_barrier(0);
    read_b4((u + 1) & 1);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u & 1][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u & 1][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u & 1][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u & 1][3], acc, 0, 0, 0);
  };
  read_b4(0);
  // The ring's first loads were issued before the staging loop and have long landed. Saying so explicitly (vmcnt(0)) gives
  // the waitcnt pass a clean state at the loop header; otherwise it merges the staging loop's pending loads into the first
  // wait of every iteration (vmcnt(2) instead of vmcnt(7): the ring drained once per 32 steps).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int G = Sw >> 2;
  const int full = G / kRA;
  for (int g = 0; g < full; g++) {
#pragma unroll
    for (int u = 0; u < kRA; u++) group(u);
    wa += kRA * 1024;
  }
  {
    const int rem = G - full * kRA;
#pragma unroll
    for (int u = 0; u < kRA - 1; u++)
      if (u < rem) group(u);
  }

  PH_WSTAMP(4);
  if constexpr (KS > 1) {  // slice 0 + slice 1 + … in fixed order
    __syncthreads();       // every wave is done reading the window: its memory becomes the exchange area
    float* red = win;
    if (ks > 0) {
      float* dst = red + ((ks - 1) * (WM * WN) + wt) * (16 * 64) + lane;
#pragma unroll
      for (int q = 0; q < 16; q++) dst[q * 64] = acc[q];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int s2 = 1; s2 < KS; s2++) {
      const float* src = red + ((s2 - 1) * (WM * WN) + wt) * (16 * 64) + lane;
#pragma unroll
      for (int q = 0; q < 16; q++) acc[q] += src[q * 64];
    }
  }

  PH_WSTAMP(5);
  // ---- epilogue. register q of lane (r,h): row (q&3) + 8·(q>>2) + 4·h, column r. Loads first (clamped), stores masked.
  const int rows_total = ct ? p.Cout * p.ct_stride : p.Cout;
  const int row0 = mt * 32;
  if (row0 >= rows_total) return;
  const int co0 = ct ? row0 - rho * p.Cout : row0;
  const int col = nb0 + wn * 32 + r;
  const bool okc = col < p.Lout;
  const int colc = min(col, p.Lout - 1);
  const int pos = ct ? colc * p.ct_stride + rho : colc;
  int64_t yi[16];
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const int coc = min(co0 + (q & 3) + 8 * (q >> 2) + 4 * h, p.Cout - 1);
    yi[q] = ((int64_t)n * p.Cout + coc) * p.y_len + pos;
    v[q] = acc[q] + (p.bias ? p.bias[coc] : 0.0f);
  }
  if (p.res) {
    float t[16];
#pragma unroll
    for (int q = 0; q < 16; q++) t[q] = p.res[yi[q]];
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] += t[q];
  }
  if (p.mrf_a) {
    float ta[16], tb[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { ta[q] = p.mrf_a[yi[q]]; tb[q] = p.mrf_b[yi[q]]; }
#pragma unroll
    for (int q = 0; q < 16; q++) v[q] = ((ta[q] + tb[q]) + v[q]) / 3.0f;
  }
#pragma unroll
  for (int q = 0; q < 16; q++)
    if (okc && co0 + (q & 3) + 8 * (q >> 2) + 4 * h < p.Cout) p.y[yi[q]] = lrelu1(v[q], p.out_alpha);
  PH_WSTAMP(6);
}

template <int WM, int WN, int KS>
int launch_inst(hipStream_t s, const ConvWinMulti& a, int batch, dim3 grid, size_t lds, int xcd_rows) {
  static bool raised[kMaxDevices] = {};  // the opt-in is per device
  if (lds > 64 * 1024 && lds_optin_needed(raised))
    (void)hipFuncSetAttribute((const void*)conv_win_kernel<WM, WN, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((conv_win_kernel<WM, WN, KS>), grid, dim3(64 * WM * WN * KS), lds, s, a, batch, xcd_rows);
  return PIPER_HIP_OK;
}

struct WinGeom {
  int MT, taps, reach, per_phase;
};
WinGeom geom(const ConvWinArgs& a) {
  WinGeom g;
  const bool ct = a.ct_stride > 0;
  const int rows = ct ? a.Cout * a.ct_stride : a.Cout;
  g.MT = (rows + 31) / 32;
  g.taps = ct ? a.K / a.ct_stride : a.K;
  g.reach = ct ? (a.ct_stride - 1 + a.ct_pad) / a.ct_stride + g.taps - 1 : (a.K - 1) * a.dil;
  g.per_phase = ct ? a.Cout / 32 : g.MT;
  return g;
}
size_t lds_bytes(int Cin, int wn, int reach, int ks, int wmwn) {
  const int Wp = (wn * 32 + reach + 6) & ~3;
  const size_t window = ((size_t)Cin * Wp + 4) * 4;
  const size_t red = (size_t)(ks - 1) * wmwn * 16 * 64 * 4;
  return window > red ? window : red;
}

}  // namespace

#ifdef PH_WIN_TRACE
void conv_win_set_trace(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(ph_win_trace_buf), &buf, sizeof buf); }
#endif

size_t packed_conv_win_floats(int Cout, int Cin, int K) { return (size_t)((Cout + 31) / 32) * padded_steps_win(Cin, K) * 64 + kTailFloats; }
size_t packed_convt_win_floats(int Cin, int Cout, int K, int stride) {
  return (size_t)((Cout * stride + 31) / 32) * padded_steps_win(Cin, K / stride) * 64 + kTailFloats;
}

int pack_conv_weights_win(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed) {
  const int64_t total = (int64_t)packed_conv_win_floats(Cout, Cin, K);
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBT), 4096);
  hipLaunchKernelGGL(pack_conv_win_kernel, dim3(grid), dim3(kBT), 0, s, w, Cout, Cin, K, K * (Cin / 2), padded_steps_win(Cin, K), packed, total);
  return PIPER_HIP_OK;
}
int pack_convt_weights_win(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, int pad, float* packed) {
  const int64_t total = (int64_t)packed_convt_win_floats(Cin, Cout, K, stride);
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBT), 4096);
  const int J = K / stride;
  hipLaunchKernelGGL(pack_convt_win_kernel, dim3(grid), dim3(kBT), 0, s, w, Cin, Cout, K, stride, pad, J * (Cin / 2), padded_steps_win(Cin, J),
                     packed, total);
  return PIPER_HIP_OK;
}

bool conv_win_eligible(int Cout, int Cin, int K, int dil, int padL, int Lin, int Lout) {
  if (Cout < 1 || Cin < 2 || (Cin & 1) || K < 1 || dil < 1 || padL < 0 || Lin < 4 || (Lin & 3) || Lout < 1) return false;
  return lds_bytes(Cin, 1, (K - 1) * dil, 1, 1) <= 160 * 1024;
}
bool convt_win_eligible(int Cin, int Cout, int K, int stride, int pad, int Lin) {
  if (Cin < 2 || (Cin & 1) || Cout < 32 || (Cout & 31) || stride < 1 || Lin < 4 || (Lin & 3)) return false;
  return K % stride == 0 && K - stride == 2 * pad;
}

int launch_conv_win(piper_hip_ctx* ctx, hipStream_t s, const ConvWinArgs& a) { return launch_conv_win_multi(ctx, s, &a, 1); }

int launch_conv_win_multi(piper_hip_ctx* ctx, hipStream_t s, const ConvWinArgs* convs, int count) {
  if (count < 1 || count > kWinMulti) PH_FAIL(PIPER_HIP_ERR_ARG, "conv_win: %d convs in one launch (1..%d)", count, kWinMulti);
  const ConvWinArgs& a = convs[0];
  if (a.N <= 0 || a.Lout <= 0) return PIPER_HIP_OK;
  const WinGeom g = geom(a);
  int reach = g.reach, Sp_min = padded_steps_win(a.Cin, g.taps);
  for (int i = 1; i < count; i++) {
    const ConvWinArgs& b = convs[i];
    if (b.N != a.N || b.Cin != a.Cin || b.Cout != a.Cout || b.Lin != a.Lin || b.Lout != a.Lout || b.ct_stride != a.ct_stride)
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv_win: convs of one launch must share N, Cin, Cout, Lin, Lout and kind");
    const WinGeom gb = geom(b);
    reach = std::max(reach, gb.reach);
    Sp_min = std::min(Sp_min, padded_steps_win(b.Cin, gb.taps));
  }
  const int64_t tiles = (int64_t)g.MT * ceil_div(a.Lout, 32) * a.N * count;
  const int64_t simds = 4 * (int64_t)ctx->num_cus;
  // split the contraction while the tiles give a SIMD fewer than two waves (each split wave keeps ≥ 16 steps); one wave per SIMD
  // was the target until the r2 sweeps (factor 8 stage-1 upsampler, 1344 tiles: 27.6 → 25.4 µs with the split)
  static const bool ks_one_wave = getenv("PIPER_HIP_WIN_KS_ONE_WAVE") != nullptr;  // A/B: round 1's one-wave-per-SIMD target
  int KS = ks_one_wave ? (tiles >= simds ? 1 : (2 * tiles >= simds ? 2 : 4)) : (tiles >= 2 * simds ? 1 : (tiles >= simds ? 2 : 4));
  while (KS > 1 && Sp_min / KS < 16) KS >>= 1;
  // candidates in order of preference: the wanted K-split first; waves along rows before columns (they then share the
  // staged window and nothing else); a single row tile (C = 32) puts the waves side by side along the columns
  static const int cand[7][3] = {{4, 1, 1}, {2, 2, 1}, {1, 4, 1}, {2, 1, 2}, {1, 2, 2}, {1, 1, 4}, {4, 1, 2}};
  constexpr int kCand4 = 6;  // the first six are the 4-wave blocks
  // 8 waves sharing one window: plain convs whose tiles would otherwise be split four ways, four row tiles each staging ≥ 128 rows
  int best = -1;
  static const char* force = getenv("PIPER_HIP_WIN_CFG");  // tuning hook: "WM,WN,KS"
  int fm = 0, fn = 0, fk = 0;
  const bool forced = force && sscanf(force, "%d,%d,%d", &fm, &fn, &fk) == 3;
  if (!forced && a.ct_stride == 0 && g.MT % 4 == 0 && a.Cin >= 128 && KS == 4 && Sp_min / 2 >= 16 && lds_bytes(a.Cin, 1, reach, 2, 4) <= 160 * 1024) best = 6;
  for (int pass = 0; pass < 3 && best < 0; pass++)  // 0: forced, 1: wanted KS, 2: anything that fits
    for (int i = 0; i < (pass == 0 ? 7 : kCand4) && best < 0; i++) {
      const int m = cand[i][0], nn = cand[i][1], k = cand[i][2];
      if (g.MT % m || (k > 1 && Sp_min / k < 16) || lds_bytes(a.Cin, nn, reach, k, m * nn) > 160 * 1024) continue;
      if (pass == 0 && !(forced && m == fm && nn == fn && k == fk)) continue;
      if (pass == 1 && k != KS) continue;
      best = i;
    }
  if (best < 0) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_win: no configuration fits (Cin=%d reach=%d rows/phase=%d)", a.Cin, reach, g.per_phase);
  const int WM = cand[best][0], WN = cand[best][1];
  KS = cand[best][2];
  const size_t lds = lds_bytes(a.Cin, WN, reach, KS, WM * WN);
  const dim3 grid((unsigned)ceil_div(a.Lout, WN * 32), (unsigned)ceil_div(g.MT, WM), (unsigned)(a.N * count));
  ConvWinMulti multi;
  for (int i = 0; i < kWinMulti; i++) multi.c[i] = convs[i < count ? i : 0];
  // row-tile groups pinned to XCDs when the weight image is the larger operand (see the kernel)
  static const bool no_xcd = getenv("PIPER_HIP_WIN_NO_XCD_ROWS") != nullptr;
  const int64_t w_bytes = (int64_t)g.MT * Sp_min * 256, x_bytes = (int64_t)a.Cin * a.Lin * 4 * (a.x2 ? 3 : 1);
  const int xcd_rows = (!no_xcd && grid.y % 8 == 0 && w_bytes > x_bytes) ? 1 : 0;
#define PH_WIN_CASE(M, N_, K_) \
  if (WM == M && WN == N_ && KS == K_) launch_inst<M, N_, K_>(s, multi, a.N, grid, lds, xcd_rows); else
  PH_WIN_CASE(4, 1, 1) PH_WIN_CASE(2, 2, 1) PH_WIN_CASE(1, 4, 1) PH_WIN_CASE(2, 1, 2) PH_WIN_CASE(1, 2, 2) PH_WIN_CASE(1, 1, 4)
  PH_WIN_CASE(4, 1, 2)
  PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_win: no instance for WM=%d WN=%d KS=%d", WM, WN, KS);
#undef PH_WIN_CASE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_win launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(conv_win, pack_conv_win_kernel); } }
