// conv.hip — Conv1d / ConvTranspose1d for gfx950.
//
// Main path: implicit GEMM on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32
//   D[32 co × 32 t] += A[32 co × 2 ci] · B[2 ci × 32 t]          (one instruction, 64 lanes)
// The contraction runs over (channel pair, tap); a step's B fragment is the activation row window
// x[ci][t + tap·dil − padL] — 32 consecutive floats per channel row, i.e. two 128-byte segments per
// wave load — so taps, dilation, zero padding, the pre-activation (LeakyReLU / MRF mean), the VITS
// `Flip`/`Split` channel remaps and the ConvTranspose phase decomposition are all address arithmetic on
// the B load; no im2col buffer is ever materialised.  A fragments come from a one-time packed image
// of the weights (64 consecutive floats per step: one coalesced 256-byte load, L2-resident).
// Numerics: the instruction is bit-for-bit a k-ordered fmaf chain (guide §3), accumulator seeded with
// the bias (bias-first like CPUBackend.swift:46-63).
//
// A 256-thread block holds 4 waves = (4/KS) output tiles × KS contraction slices; KS > 1 is used when the
// tensor is too short to give every SIMD a tile (utterances are small), slices are summed in a fixed
// order through LDS.  Each wave owns 32 (or 2×32 with the gate) output channels × NT·32 time steps.
//
// Replaces conv1d_f32 / convtranspose1d_f32 (Kernels/conv1d.metal:28-71, 97-142) and their encoders
// (MetalBackend.swift:1149-1228, 2812-2895).
#include "conv.h"
#include "conv_kernels.hpp"

namespace ph {
namespace detail {
// instantiated in conv_inst_k*.hip
#define PH_EXTERN_K(K)                                                                                                         \
  extern template bool launch_k<K, 32>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);              \
  extern template bool launch_k<K, 16>(hipStream_t, const ConvArgs&, int, int, int, int, int, int, dim3, size_t);              \
  extern template bool launch_tile_k<K>(hipStream_t, const ConvArgs&, int, int, int);
PH_EXTERN_K(1) PH_EXTERN_K(2) PH_EXTERN_K(3) PH_EXTERN_K(5) PH_EXTERN_K(7) PH_EXTERN_K(11)
#undef PH_EXTERN_K
}  // namespace detail
using namespace detail;
namespace {

// ---- weight packing (once per voice; per call for the op-level API) ----
__global__ __launch_bounds__(kBlock) void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                           int K, int mtiles, int nsteps, int tm, int gate_half) {
  const int cps = tm == 32 ? 2 : 4;
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int lane = (int)(i & 63);
    const int64_t ms = i >> 6;
    const int step = (int)(ms % nsteps), mt = (int)(ms / nsteps);
    const int cp = step / K, tap = step - cp * K;  // steps beyond the real channel pairs are zero padding
    int co = mt * tm + (lane & (tm - 1));
    const int ci = cps * cp + lane / tm;
    bool row_ok = co < Cout;
    if (gate_half) {  // tile row i < 8: tanh row 8·mt + i; i ≥ 8: its sigmoid partner gate_half + 8·mt + i − 8
      const int i16 = lane & 15, h = 8 * mt + (i16 & 7);
      row_ok = h < gate_half;
      co = (i16 < 8 ? 0 : gate_half) + h;
    }
    out[i] = (row_ok && ci < Cin) ? w[((int64_t)co * Cin + ci) * K + tap] : 0.0f;
  }
}

// 8-row fragments: element (k = channel of the quad, i = row of the tile) at [tile][step = (quad, tap)][k·8 + i]
__global__ __launch_bounds__(kBlock) void pack_conv_rows8_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int K, int mtiles, int nsteps) {
  const int64_t total = (int64_t)mtiles * nsteps * 32;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int l = (int)(i & 31);
    const int64_t ms = i >> 5;
    const int step = (int)(ms % nsteps), mt = (int)(ms / nsteps);
    const int cp = step / K, tap = step - cp * K;
    const int co = mt * 8 + (l & 7), ci = 4 * cp + (l >> 3);
    out[i] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * K + tap] : 0.0f;
  }
}

// ConvTranspose [Cin, Cout, K], stride s → GEMM rows (co, phase), taps j: weight W[ci][co][phase + s·j]
__global__ __launch_bounds__(kBlock) void pack_convt_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout,
                                                            int K, int s, int J, int mtiles, int nsteps, int tm) {
  const int cps = tm == 32 ? 2 : 4;
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  const int rows = Cout * s;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int lane = (int)(i & 63);
    const int64_t ms = i >> 6;
    const int step = (int)(ms % nsteps), mt = (int)(ms / nsteps);
    const int cp = step / J, jt = step - cp * J;
    const int row = mt * tm + (lane & (tm - 1)), ci = cps * cp + lane / tm;
    float v = 0.0f;
    if (row < rows && ci < Cin) {
      const int co = row / s, ph = row - co * s;
      const int k = ph + s * jt;
      if (k < K) v = w[((int64_t)ci * Cout + co) * K + k];
    }
    out[i] = v;
  }
}

// ---- direct kernels (thread per output) ----
template <int PRO>
__global__ __launch_bounds__(kBlock) void conv_direct_kernel(const ConvArgs p) {
  const int64_t total = (int64_t)p.N * p.Cout * p.Lout;
  const int cig = p.Cin / p.groups, cog = p.Cout / p.groups;
  for (int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * kBlock) {
    const int xo = (int)(gid % p.Lout);
    const int64_t t2 = gid / p.Lout;
    const int co = (int)(t2 % p.Cout), n = (int)(t2 / p.Cout);
    float acc = p.bias ? p.bias[co] : 0.0f;
    const int ciBase = (co / cog) * cig;
    const float* xb = p.x + (int64_t)n * p.x_batch_stride;
    const float* x2b = PRO == PRO_AVG3_LRELU ? p.x2 + (int64_t)n * p.x_batch_stride : nullptr;
    const float* x3b = PRO == PRO_AVG3_LRELU ? p.x3 + (int64_t)n * p.x_batch_stride : nullptr;
    const float* wr = p.w + (int64_t)co * cig * p.K;
    const int inX0 = xo * p.stride - p.padL;
    const int Lv = true_len(p, n);
    for (int ci = 0; ci < cig; ci++) {
      const int64_t roff = (int64_t)(p.in_ch_base + p.in_ch_sign * (ciBase + ci)) * p.Lin;
      for (int k = 0; k < p.K; k++) {
        const int pos = inX0 + k * p.dil;
        const float v = load_b<PRO>(p, xb + roff, PRO == PRO_AVG3_LRELU ? x2b + roff : nullptr,
                                    PRO == PRO_AVG3_LRELU ? x3b + roff : nullptr, pos, true, Lv);
        acc += v * wr[ci * p.K + k];
      }
    }
    store_elem(p, n, co, xo, acc);
  }
}

// Few output channels (HiFi-GAN conv_post: 32 → 1, k 7): HBM-bound, no matrix shape to speak of.  One output position
// per thread; the (pre-activated, zero-padded) input window of the block goes through LDS once, so every input element
// is read from memory exactly once and all of a thread's loads are independent (issued back to back).
constexpr int kSmallCoutMax = 4, kSmallCK = 32, kSmallBT = 256;
template <int PRO, int COUT>
__global__ __launch_bounds__(kSmallBT) void conv_small_cout_kernel(const ConvArgs p, const int vec4) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int halo = (p.K - 1) * (p.dil < 0 ? -p.dil : p.dil);
  // vec4 (rows 16-byte aligned, forward channels and taps): the window starts at the aligned position below `lo` and is staged
  // as float4s — 8 channels × (1 or 3 sources) per thread, ALL in flight (one memory round trip per 32-channel chunk instead
  // of twelve dependent batches of dword loads: conv_post 28 → r2 µs at factor 8)
  const int W = vec4 ? ((kSmallBT + halo + 3 + 3) & ~3) : kSmallBT + halo;
  float* xs = sm;                  // [kSmallCK][W]
  float* ws = sm + kSmallCK * W;   // [Cout][kSmallCK][K]
  const int n = blockIdx.y, t0 = blockIdx.x * kSmallBT, tid = threadIdx.x;
  const int lo = t0 - p.padL + (p.dil < 0 ? (p.K - 1) * p.dil : 0);  // input position of window column 0
  const float* xb = p.x + (int64_t)n * p.x_batch_stride;
  const float* x2b = PRO == PRO_AVG3_LRELU ? p.x2 + (int64_t)n * p.x_batch_stride : nullptr;
  const float* x3b = PRO == PRO_AVG3_LRELU ? p.x3 + (int64_t)n * p.x_batch_stride : nullptr;
  const int Lv = true_len(p, n);
  if (p.len_ptr && t0 >= Lv) return;  // every output of this block lies past the true length ('same' convs: Lout = Lin); block-uniform
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; co++) acc[co] = p.bias ? p.bias[co] : 0.0f;
  for (int c0 = 0; c0 < p.Cin; c0 += kSmallCK) {
    const int ck = min(kSmallCK, p.Cin - c0);
    __syncthreads();
    // channel rows × window columns: column = tid (+ 256 per extra pass), so there is no per-element division and the
    // 8-way unrolled channel loop keeps 8 independent loads per thread in flight
    if (vec4) {
      const int la = lo & ~3;              // aligned position of window column 0
      const int W4 = W >> 2;
      const int cg = tid >> 6, pt = tid & 63;  // 4 channel groups of 8 × 64 float4 columns
      for (int i4 = pt; i4 < W4; i4 += 64) {
        const int pos = la + 4 * i4;       // multiple of 4: a float4 is inside or outside [0, Lin) as a whole
        const bool inb = pos >= 0 && pos < p.Lin;
        const int nvalid = inb ? Lv - pos : 0;
        float4 t[8], t2[PRO == PRO_AVG3_LRELU ? 8 : 1], t3[PRO == PRO_AVG3_LRELU ? 8 : 1];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int c = min(cg * 8 + q, ck - 1);
          const int64_t off = (int64_t)(c0 + c) * p.Lin + (inb ? pos : 0);
          t[q] = *(const float4*)(xb + off);
          if constexpr (PRO == PRO_AVG3_LRELU) { t2[q] = *(const float4*)(x2b + off); t3[q] = *(const float4*)(x3b + off); }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          float4 v = t[q];
          if constexpr (PRO == PRO_AVG3_LRELU) {
            v.x = ((v.x + t2[q].x) + t3[q].x) / 3.0f; v.y = ((v.y + t2[q].y) + t3[q].y) / 3.0f;
            v.z = ((v.z + t2[q].z) + t3[q].z) / 3.0f; v.w = ((v.w + t2[q].w) + t3[q].w) / 3.0f;
          }
          if constexpr (PRO != PRO_NONE) { v.x = lrelu(v.x, p.alpha); v.y = lrelu(v.y, p.alpha); v.z = lrelu(v.z, p.alpha); v.w = lrelu(v.w, p.alpha); }
          v.x = nvalid > 0 ? v.x : 0.0f; v.y = nvalid > 1 ? v.y : 0.0f; v.z = nvalid > 2 ? v.z : 0.0f; v.w = nvalid > 3 ? v.w : 0.0f;
          if (cg * 8 + q < ck) *(float4*)(xs + (cg * 8 + q) * W + 4 * i4) = v;
        }
      }
    } else
    for (int cb = 0; cb < W; cb += kSmallBT) {
      const int col = cb + tid;
      const int pos = lo + col;
      const bool ok = col < W && pos >= 0 && pos < Lv;
      const int64_t poff = ok ? pos : 0;
      if (col < W) {
#pragma unroll 8
        for (int c = 0; c < ck; c++) {
          const int64_t off = (int64_t)(p.in_ch_base + p.in_ch_sign * (c0 + c)) * p.Lin + poff;
          float v = xb[off];
          if constexpr (PRO == PRO_AVG3_LRELU) v = ((v + x2b[off]) + x3b[off]) / 3.0f;
          if constexpr (PRO != PRO_NONE) v = lrelu(v, p.alpha);
          xs[c * W + col] = ok ? v : 0.0f;
        }
      }
    }
    for (int idx = tid; idx < p.Cout * ck * p.K; idx += kSmallBT) {
      const int co = idx / (ck * p.K), rem = idx - co * ck * p.K;
      const int c = rem / p.K, k = rem - c * p.K;
      ws[(co * kSmallCK + c) * p.K + k] = p.w[((int64_t)co * p.Cin + c0 + c) * p.K + k];
    }
    __syncthreads();
    const int colbase = tid - p.padL - (lo - t0) + (vec4 ? (lo & 3) : 0);  // window column of tap 0 for this thread
    for (int c = 0; c < ck; c++) {
      const float* xr = xs + c * W + colbase;
#pragma unroll 8
      for (int k = 0; k < p.K; k++) {
        const float xv = xr[k * p.dil];
#pragma unroll
        for (int co = 0; co < COUT; co++) acc[co] += xv * ws[(co * kSmallCK + c) * p.K + k];  // ci-major, then k: reference order
      }
    }
  }
  const int xo = t0 + tid;
  if (xo < p.Lout) {
#pragma unroll
    for (int co = 0; co < COUT; co++) store_elem(p, n, co, xo, acc[co]);
  }
}

// conv_post at its real size (Cout 1, kernel 7, 'same' padding, rows of 86 016 … 688 128 steps, input = MRF mean of three
// tensors): the one-output-per-thread kernel above is bound by its LDS reads (7 window + 7 weight reads per channel and
// output). Here a thread owns FOUR consecutive outputs: per channel 4 aligned ds_read_b128 give the 4 + K − 1 window values,
// 2 broadcast ds_read_b128 the taps — 6 LDS reads per 28 FMAs instead of 56 — and the window is staged 8 channels at a time as
// float4s with the NEXT chunk's loads (≤ 27 per thread) in flight under the current chunk's arithmetic.
// The block size is a template parameter: 256 threads (1024 outputs) for long rows; ONE wave (256 outputs) when 1024-output blocks
// would leave most CUs without one (a factor-8 utterance is 84 of them): r2, 24.8 → 22.7 µs there, 26.1 → 20.2 µs at factor 1.
// (Two chunks of loads in flight instead of one was also measured: no gain on short rows, 61 → 83 µs at factor 64 — 256 VGPRs.)
constexpr int kWideCK = 8, kWideK = 7, kWidePad = 3;
template <int PRO, int BT>
__global__ __launch_bounds__(BT) void conv_cout1_wide_kernel(const ConvArgs p) {
  constexpr int kWideBT = BT, kWideOut = 4 * BT, kWideW = kWideOut + 16;
  __shared__ __attribute__((aligned(16))) float xs[kWideCK * kWideW];
  __shared__ __attribute__((aligned(16))) float ws[256 * 8];  // [Cin ≤ 256][8]: 7 taps + a zero
  constexpr bool AVG = PRO == PRO_AVG3_LRELU;
  const int n = blockIdx.y, t0 = blockIdx.x * kWideOut, tid = threadIdx.x;
  const int Lv = true_len(p, n);
  if (p.len_ptr && t0 >= Lv) return;  // block-uniform: every output lies past the true length
  const int la = t0 - 4;               // aligned position of window column 0 (t0 − pad = la + 1)
  const float* xb = p.x + (int64_t)n * p.x_batch_stride;
  const float* x2b = AVG ? p.x2 + (int64_t)n * p.x_batch_stride : nullptr;
  const float* x3b = AVG ? p.x3 + (int64_t)n * p.x_batch_stride : nullptr;
  for (int i = tid; i < p.Cin * 8; i += kWideBT) ws[i] = (i & 7) < kWideK ? p.w[(i >> 3) * kWideK + (i & 7)] : 0.0f;
  // BT + 4 float4 per row: BT by the thread's own column, 4 more by threads 0..31 (row = tid >> 2)
  float4 t[kWideCK + 1], t2[AVG ? kWideCK + 1 : 1], t3[AVG ? kWideCK + 1 : 1];
  auto slot = [&](int q, int& row, int& i4) {  // staging slot q of this thread → (row, float4 column); row ≥ kWideCK: none
    row = q < kWideCK ? q : (tid < 32 ? (tid >> 2) : kWideCK);
    i4 = q < kWideCK ? tid : BT + (tid & 3);
  };
  auto issue = [&](int c0) {
#pragma unroll
    for (int q = 0; q <= kWideCK; q++) {
      int row, i4;
      slot(q, row, i4);
      const int pos = la + 4 * i4;
      const int c = min(c0 + min(row, kWideCK - 1), p.Cin - 1);
      const int64_t off = (int64_t)c * p.Lin + ((pos >= 0 && pos < p.Lin) ? pos : 0);
      t[q] = *(const float4*)(xb + off);
      if constexpr (AVG) { t2[q] = *(const float4*)(x2b + off); t3[q] = *(const float4*)(x3b + off); }
    }
  };
  auto commit = [&](int c0) {
#pragma unroll
    for (int q = 0; q <= kWideCK; q++) {
      int row, i4;
      slot(q, row, i4);
      const int pos = la + 4 * i4;
      const int nvalid = (pos >= 0 && pos < p.Lin && c0 + row < p.Cin) ? Lv - pos : 0;
      float4 v = t[q];
      if constexpr (AVG) {  // ((x + x2) + x3) / 3: the association of the graph's Add, Add, Div
        v.x = ((v.x + t2[q].x) + t3[q].x) / 3.0f; v.y = ((v.y + t2[q].y) + t3[q].y) / 3.0f;
        v.z = ((v.z + t2[q].z) + t3[q].z) / 3.0f; v.w = ((v.w + t2[q].w) + t3[q].w) / 3.0f;
      }
      if constexpr (PRO != PRO_NONE) { v.x = lrelu(v.x, p.alpha); v.y = lrelu(v.y, p.alpha); v.z = lrelu(v.z, p.alpha); v.w = lrelu(v.w, p.alpha); }
      v.x = nvalid > 0 ? v.x : 0.0f; v.y = nvalid > 1 ? v.y : 0.0f; v.z = nvalid > 2 ? v.z : 0.0f; v.w = nvalid > 3 ? v.w : 0.0f;
      if (row < kWideCK) *(float4*)(xs + row * kWideW + 4 * i4) = v;
    }
  };
  float acc[4];
#pragma unroll
  for (int o = 0; o < 4; o++) acc[o] = p.bias ? p.bias[0] : 0.0f;  // bias first (CPUBackend.conv1d)
  issue(0);
  for (int c0 = 0; c0 < p.Cin; c0 += kWideCK) {
    __syncthreads();                     // the previous chunk's readers are done (first pass: ws is complete)
    commit(c0);
    __syncthreads();
    if (c0 + kWideCK < p.Cin) issue(c0 + kWideCK);
    const int ck = min(kWideCK, p.Cin - c0);
    for (int c = 0; c < ck; c++) {       // ci-major, then k: the reference's order
      const float4* xr = (const float4*)(xs + c * kWideW + 4 * tid);
      const float4 v0 = xr[0], v1 = xr[1], v2 = xr[2], v3 = xr[3];
      const float4 w0 = *(const float4*)(ws + (c0 + c) * 8), w1 = *(const float4*)(ws + (c0 + c) * 8 + 4);
      const float v[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int k = 0; k < kWideK; k++)
#pragma unroll
        for (int o = 0; o < 4; o++) acc[o] += v[1 + o + k] * w[k];  // output 4·tid + o, tap k: window column 4·tid + 1 + o + k
    }
  }
#pragma unroll
  for (int o = 0; o < 4; o++) {
    const int xo = t0 + 4 * tid + o;
    if (xo < p.Lout) store_elem(p, n, 0, xo, acc[o]);
  }
}

// The same conv on SHORT rows (one utterance of a few seconds: 84 … 300 blocks of 1 024 outputs would leave most CUs idle, and the one-wave
// 256-output blocks of round 2 left three of a CU's four SIMDs idle while each wave issued ≈ 4 300 vector instructions — r3 PMC
// SQ_INSTS_VALU — a third of them the correctly rounded divisions of the MRF mean). Here a 256-output block is FOUR waves that split
// the input channels (wave w takes channels 8w … 8w+7 of every group of 32), each staging its own eight rows in its own piece of LDS
// (no block barrier between chunks), and the four partial sums meet once at the end: 4 × the waves, a quarter of the work each.
// The mean is x·(1/3) here — one rounding step from the graph's Add, Add, Div (≤ 1 ulp, far inside the stated waveform tolerance).
template <int PRO>
__global__ __launch_bounds__(256) void conv_cout1_split_kernel(const ConvArgs p) {
  constexpr int kOut = 256, kW = kOut + 16;
  __shared__ __attribute__((aligned(16))) float xs_all[4 * kWideCK * kW];
  __shared__ __attribute__((aligned(16))) float ws[256 * 8];  // [Cin ≤ 256][8]: 7 taps + a zero
  __shared__ float red[4 * kOut];
  constexpr bool AVG = PRO == PRO_AVG3_LRELU;
  const int n = blockIdx.y, t0 = blockIdx.x * kOut, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Lv = true_len(p, n);
  if (p.len_ptr && t0 >= Lv) return;  // block-uniform: every output lies past the true length
  float* xs = xs_all + wave * (kWideCK * kW);
  const int la = t0 - 4;               // aligned position of window column 0 (t0 − pad = la + 1)
  const float* xb = p.x + (int64_t)n * p.x_batch_stride;
  const float* x2b = AVG ? p.x2 + (int64_t)n * p.x_batch_stride : nullptr;
  const float* x3b = AVG ? p.x3 + (int64_t)n * p.x_batch_stride : nullptr;
  for (int i = tid; i < p.Cin * 8; i += 256) ws[i] = (i & 7) < kWideK ? p.w[(i >> 3) * kWideK + (i & 7)] : 0.0f;
  float4 t[kWideCK + 1], t2[AVG ? kWideCK + 1 : 1], t3[AVG ? kWideCK + 1 : 1];
  auto slot = [&](int q, int& row, int& i4) {  // staging slot q of this lane → (row, float4 column); row ≥ kWideCK: none
    row = q < kWideCK ? q : (lane < 32 ? (lane >> 2) : kWideCK);
    i4 = q < kWideCK ? lane : 64 + (lane & 3);
  };
  auto issue = [&](int c0) {
#pragma unroll
    for (int q = 0; q <= kWideCK; q++) {
      int row, i4;
      slot(q, row, i4);
      const int pos = la + 4 * i4;
      const int c = min(c0 + min(row, kWideCK - 1), p.Cin - 1);
      const int64_t off = (int64_t)c * p.Lin + ((pos >= 0 && pos < p.Lin) ? pos : 0);
      t[q] = *(const float4*)(xb + off);
      if constexpr (AVG) { t2[q] = *(const float4*)(x2b + off); t3[q] = *(const float4*)(x3b + off); }
    }
  };
  constexpr float third = 1.0f / 3.0f;
  auto commit = [&](int c0) {
#pragma unroll
    for (int q = 0; q <= kWideCK; q++) {
      int row, i4;
      slot(q, row, i4);
      const int pos = la + 4 * i4;
      const int nvalid = (pos >= 0 && pos < p.Lin && c0 + row < p.Cin) ? Lv - pos : 0;
      float4 v = t[q];
      if constexpr (AVG) {
        v.x = ((v.x + t2[q].x) + t3[q].x) * third; v.y = ((v.y + t2[q].y) + t3[q].y) * third;
        v.z = ((v.z + t2[q].z) + t3[q].z) * third; v.w = ((v.w + t2[q].w) + t3[q].w) * third;
      }
      if constexpr (PRO != PRO_NONE) { v.x = lrelu(v.x, p.alpha); v.y = lrelu(v.y, p.alpha); v.z = lrelu(v.z, p.alpha); v.w = lrelu(v.w, p.alpha); }
      v.x = nvalid > 0 ? v.x : 0.0f; v.y = nvalid > 1 ? v.y : 0.0f; v.z = nvalid > 2 ? v.z : 0.0f; v.w = nvalid > 3 ? v.w : 0.0f;
      if (row < kWideCK) *(float4*)(xs + row * kW + 4 * i4) = v;
    }
  };
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int c_first = wave * kWideCK;
  if (c_first < p.Cin) issue(c_first);
  __syncthreads();  // ws is complete
  for (int c0 = c_first; c0 < p.Cin; c0 += 4 * kWideCK) {
    commit(c0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (c0 + 4 * kWideCK < p.Cin) issue(c0 + 4 * kWideCK);
    const int ck = min(kWideCK, p.Cin - c0);
    for (int c = 0; c < ck; c++) {
      const float4* xr = (const float4*)(xs + c * kW + 4 * lane);
      const float4 v0 = xr[0], v1 = xr[1], v2 = xr[2], v3 = xr[3];
      const float4 w0 = *(const float4*)(ws + (c0 + c) * 8), w1 = *(const float4*)(ws + (c0 + c) * 8 + 4);
      const float v[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int k = 0; k < kWideK; k++)
#pragma unroll
        for (int o = 0; o < 4; o++) acc[o] += v[1 + o + k] * w[k];  // output 4·lane + o, tap k: window column 4·lane + 1 + o + k
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  *(float4*)(red + wave * kOut + 4 * lane) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  // thread tid finishes output tid: bias first, then the four channel groups in order (deterministic)
  float v = p.bias ? p.bias[0] : 0.0f;
#pragma unroll
  for (int w2 = 0; w2 < 4; w2++) v += red[w2 * kOut + tid];
  const int xo = t0 + tid;
  if (xo < p.Lout) store_elem(p, n, 0, xo, v);
}

template <int BT>
void launch_wide(hipStream_t s, const ConvArgs& a) {
  const dim3 g((unsigned)ceil_div(a.Lout, 4 * BT), (unsigned)a.N);
  switch (a.prologue) {
    case PRO_NONE: hipLaunchKernelGGL((conv_cout1_wide_kernel<PRO_NONE, BT>), g, dim3(BT), 0, s, a); break;
    case PRO_LRELU: hipLaunchKernelGGL((conv_cout1_wide_kernel<PRO_LRELU, BT>), g, dim3(BT), 0, s, a); break;
    default: hipLaunchKernelGGL((conv_cout1_wide_kernel<PRO_AVG3_LRELU, BT>), g, dim3(BT), 0, s, a); break;
  }
}

__global__ __launch_bounds__(kBlock) void convt_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y, int N, int Cin,
                                                              int Lin, int Cout, int K, int stride, int dil, int padL, int Lout,
                                                              int groups) {
  const int64_t total = (int64_t)N * Cout * Lout;
  const int cig = Cin / groups, cog = Cout / groups;
  for (int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * kBlock) {
    const int xo = (int)(gid % Lout);
    const int64_t t2 = gid / Lout;
    const int co = (int)(t2 % Cout), n = (int)(t2 / Cout);
    const int gi = co / cog, coin = co - gi * cog;
    float acc = bias ? bias[co] : 0.0f;
    for (int ci = 0; ci < cig; ci++) {
      const int inChan = gi * cig + ci;
      const float* xr = x + ((int64_t)n * Cin + inChan) * Lin;
      const float* wr = w + ((int64_t)inChan * cog + coin) * K;
      for (int k = 0; k < K; k++) {
        const int t = xo + padL - k * dil;
        if (t % stride != 0) continue;
        const int inX = t / stride;
        if (inX >= 0 && inX < Lin) acc += xr[inX] * wr[k];
      }
    }
    y[gid] = acc;
  }
}

// tap counts with a compiled streaming kernel (Piper: 1, 3, 5, 7, 11; ConvTranspose phases: 2)
bool try_launch_tile(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a) {
  static const int mode = [] { const char* e = getenv("PIPER_HIP_TILE"); return e ? atoi(e) : -1; }();  // A/B switch: 0 off, 1 force
  if (mode == 0) return false;
  if (a.gate || a.Lout < 2048 || a.prologue == PRO_LN || a.stats_out) return false;
  const int mtiles = (int)ceil_div(a.Cout, 32);
  int MT = mtiles >= 4 ? 4 : (mtiles >= 2 ? 2 : 1);
  int NTW = MT == 4 ? 1 : 2;
  if (a.K >= 11 && MT == 4) MT = 2;  // keep the weight stage ≤ 32 KiB and CPC ≥ 2
  {
    static const int f_ntw = [] { const char* e = getenv("PIPER_HIP_TILE_NTW"); return e ? atoi(e) : 0; }();  // tuning experiments
    static const int f_mt = [] { const char* e = getenv("PIPER_HIP_TILE_MT"); return e ? atoi(e) : 0; }();
    if (f_ntw) NTW = f_ntw;
    if (f_mt && f_mt <= MT) MT = f_mt;
    if (MT == 4) NTW = 1;
  }
  // enough blocks to cover the chip?
  const int64_t blocks = ceil_div(a.Lout, 128 * NTW) * ceil_div(mtiles, MT) * a.N;
  // measured (factor 64): the tile kernel wins for ≤ 64 output channels on very long rows, the streaming kernel
  // everywhere else (ConvTranspose phases, 128-channel stages, anything that gives < 4 blocks per CU)
  if (mode != 1 && (blocks < 4 * ctx->num_cus || MT > 2 || a.epilogue == EPI_CONVT)) return false;
  const int nsteps = padded_steps(a.Cin, a.K);
  switch (a.K) {
    case 1: return launch_tile_k<1>(s, a, MT, NTW, nsteps);
    case 2: return launch_tile_k<2>(s, a, MT, NTW, nsteps);
    case 3: return launch_tile_k<3>(s, a, MT, NTW, nsteps);
    case 5: return launch_tile_k<5>(s, a, MT, NTW, nsteps);
    case 7: return launch_tile_k<7>(s, a, MT, NTW, nsteps);
    case 11: return launch_tile_k<11>(s, a, MT, NTW, nsteps);
  }
  return false;
}

}  // namespace

size_t packed_conv_floats(int Cout, int Cin, int K, int tm) { return (size_t)ceil_div(Cout, tm) * (size_t)padded_steps(Cin, K, tm) * 64; }
size_t packed_convt_floats(int Cin, int Cout, int K, int s, int tm) {
  const int J = (K + s - 1) / s;
  return (size_t)ceil_div((int64_t)Cout * s, tm) * (size_t)padded_steps(Cin, J, tm) * 64;
}

int pack_conv_weights(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed, int tm) {
  const int mtiles = (int)ceil_div(Cout, tm), nsteps = padded_steps(Cin, K, tm);
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  if (total == 0) return PIPER_HIP_OK;
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), 4096);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid), dim3(kBlock), 0, s, w, packed, Cout, Cin, K, mtiles, nsteps, tm, 0);
  return PIPER_HIP_OK;
}

size_t packed_conv_rows8_floats(int Cout, int Cin, int K) { return (size_t)ceil_div(Cout, 8) * (size_t)((Cin + 3) / 4) * K * 32; }
int pack_conv_weights_rows8(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed) {
  const int mtiles = (int)ceil_div(Cout, 8), nsteps = ((Cin + 3) / 4) * K;
  const int64_t total = (int64_t)mtiles * nsteps * 32;
  if (total == 0) return PIPER_HIP_OK;
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), 4096);
  hipLaunchKernelGGL(pack_conv_rows8_kernel, dim3(grid), dim3(kBlock), 0, s, w, packed, Cout, Cin, K, mtiles, nsteps);
  return PIPER_HIP_OK;
}

int pack_conv_weights_gate16(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed) {
  const int mtiles = (int)ceil_div(Cout, 16), nsteps = padded_steps(Cin, K, 16);
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  if (total == 0) return PIPER_HIP_OK;
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), 4096);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid), dim3(kBlock), 0, s, w, packed, Cout, Cin, K, mtiles, nsteps, 16, Cout / 2);
  return PIPER_HIP_OK;
}

int pack_convt_weights(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, float* packed, int tm) {
  const int J = (K + stride - 1) / stride;
  const int mtiles = (int)ceil_div((int64_t)Cout * stride, tm), nsteps = padded_steps(Cin, J, tm);
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  if (total == 0) return PIPER_HIP_OK;
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), 4096);
  hipLaunchKernelGGL(pack_convt_kernel, dim3(grid), dim3(kBlock), 0, s, w, packed, Cin, Cout, K, stride, J, mtiles, nsteps, tm);
  return PIPER_HIP_OK;
}

bool conv_mfma_eligible(int Cout, int Cin, int K, int stride, int groups) {
  return stride == 1 && groups == 1 && Cout >= 8 && Cin >= 2 && k_supported(K);
}

// 16-wide tiles below this many 32-wide tiles (PIPER_HIP_TM16_BELOW). Round 1 used them only when 32-wide tiles could not give
// every CU one; the r2 sweeps say four times as many, smaller tiles win far beyond that: 0 … 128 cost +0.06 … +0.41 ms on the
// factor-8 utterance; 256 → 1024 → 2048 take 4 × factor 8 from 2.10 to 1.85 ms, 8 × factor 8 from 3.14 to 2.91, factor 64
// from 3.17 to 3.10; above 2048 (up to "always") nothing changes.
static int64_t tile16_limit(piper_hip_ctx* ctx) {
  static const int below = [] { const char* e = getenv("PIPER_HIP_TM16_BELOW"); return e ? atoi(e) : -1; }();
  return below >= 0 ? below : 8 * (int64_t)ctx->num_cus;
}

int launch_conv_mfma(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a_in) {
  ConvArgs a = a_in;
  a.ct_shift = -1;
  if (a.ct_stride > 0 && (a.ct_stride & (a.ct_stride - 1)) == 0) a.ct_shift = __builtin_ctz((unsigned)a.ct_stride);
  if (a.N <= 0 || a.Lout <= 0 || a.Cout <= 0) return PIPER_HIP_OK;
  if (a.Lin < 1) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv_mfma: empty input rows must take the direct path");
  if (a.N > 65535) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv: batch %d too large", a.N);
  if (!k_supported(a.K)) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_mfma: no streaming kernel for %d taps", a.K);
  if (a.gate && (a.Cout % 64)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "gated conv needs Cout %% 64 == 0 (got %d)", a.Cout);
  if (a.prologue == PRO_LN && ((!a.ln_stats && !a.ln_self) || !a.ln_gamma || !a.ln_beta || a.in_ch_sign != 1 || a.in_ch_base != 0 || a.dil < 1 || (a.K != 1 && a.K != 3)))
    PH_FAIL(PIPER_HIP_ERR_ARG, "conv_mfma: PRO_LN needs statistics, gamma, beta, the identity channel map and K in {1, 3}");
  if (a.prologue == PRO_LN && a.Cin > 256) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_mfma: PRO_LN covers at most 256 normalised channels");
  if (a.stats_out && (a.epilogue != EPI_STORE || a.gate || a.out_ch_sign != 1 || a.out_ch_base != 0))
    PH_FAIL(PIPER_HIP_ERR_ARG, "conv_mfma: stats_out needs the plain store epilogue");
  if (a.x_batch_stride * 4 > 0x7fffffffLL) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv: input larger than 2 GiB per batch item");
  if (try_launch_tile(ctx, s, a)) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_tile launch failed: %s", hipGetErrorString(e));
    return PIPER_HIP_OK;
  }
  {  // k = 1, few tiles: the minimal-instruction kernel (conv_lean.hip)
    const int r = try_launch_conv_lean(ctx, s, a);
    if (r < 0) return r;
    if (r == 1) return PIPER_HIP_OK;
  }
  {  // short rows with few tiles: the LDS-window kernel (conv_short.hip)
    const int r = try_launch_conv_short(ctx, s, a);
    if (r < 0) return r;
    if (r == 1) return PIPER_HIP_OK;
  }
  // Tile geometry: 16×16 tiles when 32×32 tiles alone cannot give every CU one (short utterances) and the caller has
  // the 16-wide fragment image.
  int TM = 32;
  {
    const int64_t tiles32 = (a.gate ? ceil_div(a.Cout, 64) : ceil_div(a.Cout, 32)) * ceil_div(a.Lout, 32) * a.N;
    static const int force_tm = [] { const char* e = getenv("PIPER_HIP_TM"); return e ? atoi(e) : 0; }();  // tuning experiments
    if (a.w16 && force_tm != 32 && (tiles32 < tile16_limit(ctx) || force_tm == 16) && (!a.gate || a.Cout % 32 == 0)) TM = 16;
  }
  if (TM == 16) a.w = a.w16;
  if (!a.w) PH_FAIL(PIPER_HIP_ERR_ARG, "conv_mfma: missing packed weights for %d-wide tiles", TM);
  const int cps = TM == 32 ? 2 : 4;
  const int mtiles = (int)ceil_div(a.Cout, TM);
  const int mt_eff = a.gate ? mtiles / 2 : mtiles;
  const int G = group_of(a.K);
  const int ngroups = (int)ceil_div((a.Cin + cps - 1) / cps, G);
  // tile shape: give every SIMD (4 per CU) two waves before growing the per-wave tile (one wave per SIMD until round 2's sweeps)
  static const int want_mul = [] { const char* e = getenv("PIPER_HIP_KS_WANT_MUL"); return e ? atoi(e) : 2; }();  // r2 sweep: 1 / 2 / 4 → 0.881 / 0.854 / 0.859 ms at factor 8
  static const int want_mul32 = [] { const char* e = getenv("PIPER_HIP_KS_WANT_MUL32"); return e ? atoi(e) : 2; }();  // 8 × factor 8: 3.30 → 3.18 ms, factor 64 −1 %, else neutral
  const int64_t want = (int64_t)ctx->num_cus * 4 * (TM == 16 ? want_mul : want_mul32);
  int NT = TM == 32 ? 4 : 1;
  // grow the per-wave tile only while every SIMD still gets ≥ 2 waves (a second wave is what hides load latency)
  auto waves = [&](int nt) { return (int64_t)mt_eff * ceil_div(a.Lout, TM * nt) * a.N; };
  while (NT > 1 && waves(NT) < 2 * want) NT >>= 1;
  {
    static const int force_nt = [] { const char* e = getenv("PIPER_HIP_NT"); return e ? atoi(e) : 0; }();  // tuning experiments
    static const int force_nt_min = [] { const char* e = getenv("PIPER_HIP_NT_MIN_L"); return e ? atoi(e) : 4096; }();
    if (force_nt > 0 && TM == 32 && a.Lout >= force_nt_min) NT = force_nt;
  }
  if ((a.gate || a.prologue == PRO_AVG3_LRELU) && NT > 2) NT = 2;  // register budget: 2 accumulator sets / 3 raw inputs
  if (a.K >= 11 && NT > 2) NT = 2;
  // Split the contraction over KS waves of one block while SIMDs would otherwise idle and every slice keeps ≥ 2 prefetch
  // groups: short utterances have tiny outputs and long contractions, so this is their only parallelism.
  int ks_log2 = 0;
  const int ks_cap = (a.gate && TM == 32) ? 3 : 4;  // the gated 32-wide tile holds two 16-register accumulator sets
  static const bool ks_any_nt = getenv("PIPER_HIP_KS_ANY_NT") != nullptr;  // tuning experiments: K-split also with NT > 1 (≤ 4 slices: BT 256)
  while (ks_log2 < (NT == 1 ? ks_cap : 2) && waves(NT) * (1 << ks_log2) < want && ngroups / (2 << ks_log2) >= 2 && (NT == 1 || ks_any_nt)) ks_log2++;
  const int KS = 1 << ks_log2;
  const int BT = KS <= 4 ? 256 : 64 * KS;
  const int WT = (BT / 64) / KS;
  const int nchunks = (int)ceil_div(a.Lout, TM * NT);
  const int64_t tiles = (int64_t)mt_eff * nchunks;
  if (tiles > 0x7fffffff) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv: too many tiles");
  dim3 grid((unsigned)ceil_div(tiles, WT), (unsigned)a.N);
  const int NA = a.gate ? 2 : 1;
  const int NR = TM == 32 ? 16 : 4;
  const size_t lds = KS > 1 ? (size_t)(KS - 1) * WT * NA * NT * NR * 64 * sizeof(float) : 0;
  bool ok = false;
  if (TM == 32) {
    switch (a.K) {
      case 1: ok = launch_k<1, 32>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 2: ok = launch_k<2, 32>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 3: ok = launch_k<3, 32>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 5: ok = launch_k<5, 32>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 7: ok = launch_k<7, 32>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 11: ok = launch_k<11, 32>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
    }
  } else {
    switch (a.K) {
      case 1: ok = launch_k<1, 16>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 2: ok = launch_k<2, 16>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 3: ok = launch_k<3, 16>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 5: ok = launch_k<5, 16>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 7: ok = launch_k<7, 16>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
      case 11: ok = launch_k<11, 16>(s, a, NT, BT, nchunks, mtiles, ks_log2, ngroups, grid, lds); break;
    }
  }
  if (!ok) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_mfma: variant K=%d NT=%d TM=%d gate=%d prologue=%d not compiled", a.K, NT, TM, a.gate, a.prologue);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_mfma launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

int conv_pick_tile(piper_hip_ctx* ctx, int Cout, int Lout, int N, int gate) {
  const int64_t tiles32 = (gate ? ceil_div(Cout, 64) : ceil_div(Cout, 32)) * ceil_div(Lout, 32) * N;
  return (tiles32 < tile16_limit(ctx) && (!gate || Cout % 32 == 0)) ? 16 : 32;
}

int launch_conv_direct(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a) {
  const int64_t total = (int64_t)a.N * a.Cout * a.Lout;
  if (total <= 0) return PIPER_HIP_OK;
  if (a.epilogue == EPI_WN_RES_SKIP || a.epilogue == EPI_WN_SKIP_LAST || a.epilogue == EPI_CONVT || a.gate)
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "direct conv does not implement epilogue %d", a.epilogue);
  const int halo = (a.K - 1) * (a.dil < 0 ? -a.dil : a.dil);
  static const bool no_wide = getenv("PIPER_HIP_NO_WIDE_POST") != nullptr;
  static const int wide_min = [] { const char* e = getenv("PIPER_HIP_WIDE_POST_MIN"); return e ? atoi(e) : 4096; }();  // shorter rows: small-Cout kernel
  if (!no_wide && a.Cout == 1 && a.K == kWideK && a.dil == 1 && a.padL == kWidePad && a.stride == 1 && a.groups == 1 && a.Lin == a.Lout && a.Cin <= 256 &&
      a.N <= 65535 && a.in_ch_sign > 0 && a.in_ch_base == 0 && a.out_ch_sign > 0 && a.out_ch_base == 0 && (a.Lin & 3) == 0 && (a.x_batch_stride & 3) == 0 &&
      ((uintptr_t)a.x & 15) == 0 && a.Lin >= wide_min &&
      (a.prologue != PRO_AVG3_LRELU || ((((uintptr_t)a.x2 | (uintptr_t)a.x3) & 15) == 0)) &&
      (a.epilogue == EPI_STORE || a.epilogue == EPI_TANH) && !a.res && !a.gate) {
    static const bool no_split = getenv("PIPER_HIP_NO_POST_SPLIT") != nullptr;
    if (ceil_div(a.Lout, 1024) * a.N < 2 * ctx->num_cus) {
      if (!no_split) {  // short rows: 256-output blocks of four channel-splitting waves
        const dim3 g((unsigned)ceil_div(a.Lout, 256), (unsigned)a.N);
        switch (a.prologue) {
          case PRO_NONE: hipLaunchKernelGGL((conv_cout1_split_kernel<PRO_NONE>), g, dim3(256), 0, s, a); break;
          case PRO_LRELU: hipLaunchKernelGGL((conv_cout1_split_kernel<PRO_LRELU>), g, dim3(256), 0, s, a); break;
          default: hipLaunchKernelGGL((conv_cout1_split_kernel<PRO_AVG3_LRELU>), g, dim3(256), 0, s, a); break;
        }
      } else launch_wide<64>(s, a);
    } else launch_wide<256>(s, a);
    hipError_t e3 = hipGetLastError();
    if (e3 != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_cout1_wide launch failed: %s", hipGetErrorString(e3));
    return PIPER_HIP_OK;
  }
  if (a.Cout <= kSmallCoutMax && a.stride == 1 && a.groups == 1 && a.N <= 65535 && halo <= 1024 && a.K <= 64) {
    const dim3 g((unsigned)ceil_div(a.Lout, kSmallBT), (unsigned)a.N);
    // float4 staging: every row 16-byte aligned, channels and taps ascending
    const int vec4 = (a.dil > 0 && a.in_ch_sign > 0 && a.in_ch_base == 0 && (a.Lin & 3) == 0 && (a.x_batch_stride & 3) == 0 && ((uintptr_t)a.x & 15) == 0 &&
                      (a.prologue != PRO_AVG3_LRELU || ((((uintptr_t)a.x2 | (uintptr_t)a.x3) & 15) == 0))) ? 1 : 0;
    const int Wl = vec4 ? ((kSmallBT + halo + 6) & ~3) : kSmallBT + halo;
    const size_t lds = ((size_t)kSmallCK * Wl + (size_t)a.Cout * kSmallCK * a.K) * sizeof(float);
#define PH_SMALL(PROV)                                                                                                   \
  switch (a.Cout) {                                                                                                     \
    case 1: hipLaunchKernelGGL((conv_small_cout_kernel<PROV, 1>), g, dim3(kSmallBT), lds, s, a, vec4); break;                 \
    case 2: hipLaunchKernelGGL((conv_small_cout_kernel<PROV, 2>), g, dim3(kSmallBT), lds, s, a, vec4); break;                 \
    case 3: hipLaunchKernelGGL((conv_small_cout_kernel<PROV, 3>), g, dim3(kSmallBT), lds, s, a, vec4); break;                 \
    default: hipLaunchKernelGGL((conv_small_cout_kernel<PROV, 4>), g, dim3(kSmallBT), lds, s, a, vec4); break;                \
  }
    switch (a.prologue) {
      case PRO_NONE: PH_SMALL(PRO_NONE) break;
      case PRO_LRELU: PH_SMALL(PRO_LRELU) break;
      default: PH_SMALL(PRO_AVG3_LRELU) break;
    }
#undef PH_SMALL
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_small_cout launch failed: %s", hipGetErrorString(e2));
    return PIPER_HIP_OK;
  }
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), (int64_t)ctx->num_cus * 16);
  switch (a.prologue) {
    case PRO_NONE: hipLaunchKernelGGL(conv_direct_kernel<PRO_NONE>, dim3(grid), dim3(kBlock), 0, s, a); break;
    case PRO_LRELU: hipLaunchKernelGGL(conv_direct_kernel<PRO_LRELU>, dim3(grid), dim3(kBlock), 0, s, a); break;
    default: hipLaunchKernelGGL(conv_direct_kernel<PRO_AVG3_LRELU>, dim3(grid), dim3(kBlock), 0, s, a); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_direct launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

int launch_convt_direct(piper_hip_ctx* ctx, hipStream_t s, const float* x, const float* w, const float* bias, float* y, int N,
                        int Cin, int Lin, int Cout, int K, int stride, int dil, int padL, int Lout, int groups) {
  const int64_t total = (int64_t)N * Cout * Lout;
  if (total <= 0) return PIPER_HIP_OK;
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), (int64_t)ctx->num_cus * 16);
  hipLaunchKernelGGL(convt_direct_kernel, dim3(grid), dim3(kBlock), 0, s, x, w, bias, y, N, Cin, Lin, Cout, K, stride, dil, padL,
                     Lout, groups);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "convt_direct launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(conv, pack_conv_kernel); } }
