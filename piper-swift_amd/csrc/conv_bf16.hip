// conv_bf16.hip — Conv1d / ConvTranspose1d with bf16 operands and fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// Role: SURVEY.md §8d config 5 ("high geometry, bf16 activations/weights with fp32 accumulate") — the HiFi-GAN generator
// of the voice path (voice.hip, precision = PIPER_HIP_PRECISION_BF16) and the op-level piper_hip_conv1d_bf16 /
// piper_hip_convtranspose1d_bf16 entry points. The arithmetic is the reference's Conv / ConvTranspose
// (conv1d.metal:28-71, 97-142) with both operands rounded to bf16 (nearest even) and everything else in fp32.
//
// One block = 4 waves = a (row group, column block) of the implicit GEMM  Y[rows, cols] = W[rows, Cin·K] · X[Cin·K, cols]:
//   1. the block's whole input window — every input channel × (columns + dilation reach) — is copied ONCE from the C8
//      image (conv_bf16.h) into LDS with 16-byte loads; the stored zero halo of the image is the conv's zero padding,
//      so the hot loop has no bounds tests and all K taps and all row tiles reuse the staged window;
//   2. each wave owns MTW row tiles × NTW column tiles of 32×32 accumulators; per (tap, 16 channels) it takes the weight
//      fragment (16 B per lane, contiguous per wave) from L2 through an 8-deep register ring and the activation fragment
//      (ds_read_b128, conflict-free: 32 consecutive positions of one channel block) from the window, double-buffered;
//   3. epilogue: bias, fp32 residual, MRF mean, fp32 store and/or the C8 bf16 image of LeakyReLU(result) for the next conv.
#include <algorithm>
#include <cstdlib>

#include "conv_bf16.h"

namespace ph {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBT = 256;  // 4 waves

__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lrelu1(float v, float alpha) { return v >= 0.0f ? v : v * alpha; }

// ---------------------------------------------------------------- packing
// conv: element e of lane l of step (tap, c16) of row tile mt  =  w[mt·32 + (l&31)][c16·16 + 8·(l>>5) + e][tap]
__global__ __launch_bounds__(kBT) void pack_conv_bf16_kernel(const float* __restrict__ w, int Cout, int Cin, int K,
                                                            uint16_t* __restrict__ out, int64_t total) {
  const int C16 = Cin >> 4;
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBT) {
    const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
    int64_t st = i >> 9;
    const int c16 = (int)(st % C16); st /= C16;
    const int tap = (int)(st % K);
    const int mt = (int)(st / K);
    const int row = mt * 32 + (l & 31), ch = c16 * 16 + 8 * (l >> 5) + e;
    const float v = row < Cout ? w[((int64_t)row * Cin + ch) * K + tap] : 0.0f;
    out[i] = (uint16_t)(pack2_bf16(v, 0.0f) & 0xffffu);
  }
}

// convT: GEMM row R = ρ·Cout + co, tap j  ⇒  w[ci][co][(ρ+pad) mod s + s·j]
__global__ __launch_bounds__(kBT) void pack_convt_bf16_kernel(const float* __restrict__ w, int Cin, int Cout, int K, int stride,
                                                             int pad, uint16_t* __restrict__ out, int64_t total) {
  const int C16 = Cin >> 4, J = K / stride;
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBT) {
    const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
    int64_t st = i >> 9;
    const int c16 = (int)(st % C16); st /= C16;
    const int j = (int)(st % J);
    const int mt = (int)(st / J);
    const int R = mt * 32 + (l & 31), ci = c16 * 16 + 8 * (l >> 5) + e;
    const int rho = R / Cout, co = R - rho * Cout;
    const int k = (rho + pad) % stride + stride * j;
    const float v = rho < stride ? w[((int64_t)ci * Cout + co) * K + k] : 0.0f;
    out[i] = (uint16_t)(pack2_bf16(v, 0.0f) & 0xffffu);
  }
}

// fp32 [N][C][L] → C8 image of lrelu(x): one thread per (channel block, position) writes 16 bytes
__global__ __launch_bounds__(kBT) void pack_act_c8_kernel(const float* __restrict__ x, int C, int L, int64_t row, float alpha,
                                                         uint16_t* __restrict__ act, const int* __restrict__ len_ptr) {
  const int CB = (C + 7) >> 3;
  const int n = blockIdx.z, cb = blockIdx.y;
  const int Lv = len_ptr ? min(len_ptr[n], L) : L;
  const float* xb = x + ((int64_t)n * C + cb * 8) * L;
  uint4* ab = (uint4*)act + ((int64_t)n * CB + cb) * row + kC8Halo;
  for (int pos = blockIdx.x * kBT + threadIdx.x; pos < L; pos += gridDim.x * kBT) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (cb * 8 + e < C && pos < Lv) ? lrelu1(xb[(int64_t)e * L + pos], alpha) : 0.0f;
    uint4 o;
    o.x = pack2_bf16(v[0], v[1]); o.y = pack2_bf16(v[2], v[3]); o.z = pack2_bf16(v[4], v[5]); o.w = pack2_bf16(v[6], v[7]);
    ab[pos] = o;
  }
}

// C8 image of lrelu(((a + b) + c) / 3, alpha): the MRF mean of a stage's three ResBlock outputs as the next stage's input
__global__ __launch_bounds__(kBT) void pack_mean3_c8_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int C,
                                                           int L, int64_t row, float alpha, uint16_t* __restrict__ act, const int* __restrict__ len_ptr,
                                                           int len_mul) {
  const int CB = (C + 7) >> 3;
  const int n = blockIdx.z, cb = blockIdx.y;
  const int Lv = len_ptr ? min(len_ptr[n] * len_mul, L) : L;
  const int64_t off = ((int64_t)n * C + cb * 8) * L;
  uint4* ab = (uint4*)act + ((int64_t)n * CB + cb) * row + kC8Halo;
  for (int pos = blockIdx.x * kBT + threadIdx.x; pos < L; pos += gridDim.x * kBT) {
    float va[8], vb[8], vc[8], v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int64_t i = off + (int64_t)min(e, C - cb * 8 - 1) * L + pos;
      va[e] = a[i]; vb[e] = b[i]; vc[e] = c[i];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (cb * 8 + e < C && pos < Lv) ? lrelu1(((va[e] + vb[e]) + vc[e]) / 3.0f, alpha) : 0.0f;
    uint4 o;
    o.x = pack2_bf16(v[0], v[1]); o.y = pack2_bf16(v[2], v[3]); o.z = pack2_bf16(v[4], v[5]); o.w = pack2_bf16(v[6], v[7]);
    ab[pos] = o;
  }
}

// ---------------------------------------------------------------- the conv
// WM waves along rows × (4/WM) along columns; each wave MTW × NTW tiles of 32×32.
struct ConvBf16Multi {
  ConvBf16Args c[kBf16Multi];
};

template <int MTW, int NTW, int WM>
__global__ __launch_bounds__(kBT) void conv_bf16_kernel(const ConvBf16Multi multi, int batch) {
  extern __shared__ __attribute__((aligned(16))) uint4 win[];  // [Cin/8][W] positions of 8 channels + 1 dump slot
  const ConvBf16Args& p = multi.c[blockIdx.z / batch];  // wave-uniform: which of the launch's convs this block works on
  constexpr int WN = 4 / WM;
  constexpr int NBC = WN * NTW * 32;  // columns per block
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 31, h = lane >> 5;
  const bool ct = p.ct_stride > 0;
  const int taps = ct ? p.K / p.ct_stride : p.K;
  const int C16 = p.Cin >> 4, CB = p.Cin >> 3;
  const int n = blockIdx.z % batch;
  const int nb0 = blockIdx.x * NBC;                              // first column of the block
  const int mt0 = (blockIdx.y * WM + wm) * MTW;                  // first row tile of this wave
  // window: input positions [nb0 + off_min, nb0 + NBC + off_max)
  const int off_min = ct ? -(taps - 1) : -p.padL;
  const int off_max = ct ? (p.ct_stride - 1 + p.ct_pad) / p.ct_stride : (p.K - 1) * p.dil - p.padL;
  const int W = NBC + off_max - off_min;
  const int S = taps * C16;  // steps: (tap, 16 channels)
  // weight fragments: one 1 KB wave-load per step and row tile, streamed through a kDA-deep register ring. The image is
  // padded by kBf16WeightPad elements (launcher contract), so the ring may run past the last step without a clamp.
  constexpr int kDA = MTW == 1 ? 16 : 8;
  // uniform base (SGPR pair) + 32-bit lane offset: the loads take the saddr form, no 64-bit VALU adds per step
  const char* wa = (const char*)p.w + (int64_t)mt0 * S * 1024;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int64_t wtile = (int64_t)S * 1024;  // bytes per row tile
  uint4 a[kDA][MTW];
  auto load_a = [&](int slot, int ahead) {
#pragma unroll
    for (int m = 0; m < MTW; m++) a[slot][m] = *(const uint4*)(wa + m * wtile + ahead * 1024 + lane16);
  };
#pragma unroll
  for (int d = 0; d < kDA - 1; d++) load_a(d, d);  // independent of the window: in flight before the staging round trips
  {
    const uint4* xb = (const uint4*)p.x + (int64_t)n * CB * p.x_row;
    const int g0 = kC8Halo + nb0 + off_min;
    // rows of the window by wave, 64 positions per wave instruction; kStage loads in flight before the first LDS store
    // (one round trip per batch instead of one per row chunk). (row, chunk) are wave-uniform counters: no division.
    // Every lane stores: lanes past the row end / rows past the window go to a dump slot behind it, so the loads are not
    // sunk into conditional blocks (which would serialise them behind s_waitcnt vmcnt(0)).
    constexpr int kStage = 8;
    const int WCH = (W + 63) >> 6;
    const int dump = CB * W;
    int row = wave, chunk = 0;
    while (row < CB) {
      uint4 t[kStage];
      int dst[kStage];
#pragma unroll
      for (int q = 0; q < kStage; q++) {
        const int rr = min(row, CB - 1);
        const int wpos = chunk * 64 + lane;
        const int gp = min(max(g0 + wpos, 0), p.x_row - 1);  // partial last block: clamped columns are masked at the store
        t[q] = xb[(int64_t)rr * p.x_row + gp];
        dst[q] = (row < CB && wpos < W) ? rr * W + wpos : dump;
        chunk++;
        const bool wrap = chunk == WCH;
        chunk = wrap ? 0 : chunk;
        row += wrap ? 4 : 0;
      }
#pragma unroll
      for (int q = 0; q < kStage; q++) win[dst[q]] = t[q];
    }
  }
  __syncthreads();

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; m++)
#pragma unroll
    for (int j = 0; j < NTW; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[m][j][q] = 0.0f;

  // ConvTranspose: the wave's row tiles lie in one phase ρ when Cout % (32·MTW) == 0 (checked by the launcher)
  const int rho = ct ? (mt0 * 32) / p.Cout : 0;
  // window position read by tap t: conv t·dil − padL, convT ⌊(ρ+pad)/s⌋ − t  ⇒  off0 + t·dstep
  const int off0 = ct ? (rho + p.ct_pad) / p.ct_stride : -p.padL;
  const int dstep = ct ? -1 : p.dil;
  // LDS read base of this lane: channel block h, position r + wave's column offset, relative to the window start
  const int lbase = h * W + wn * NTW * 32 + r - off_min + off0;

  // activation fragments: ds_read_b128 from the window, kDB steps ahead of their MFMA (LDS latency ≈ 3 MFMAs). The
  // window index is a wave-uniform part (2·c16·W + tap·dstep, advanced with scalar selects: no branches, no multiplies)
  // plus the lane part; it stops advancing after the last step, so the ring never reads outside the window.
  constexpr int kDB = 4;
  uint4 b[kDB][NTW];
  int sidx = 0, c_n = 0, left = S - 1;
  const int wrap_delta = dstep - 2 * W * (C16 - 1);
  auto read_b = [&](int slot) {
    const int idx = lbase + sidx;
#pragma unroll
    for (int j = 0; j < NTW; j++) b[slot][j] = win[idx + 32 * j];
    c_n++;
    const bool wrap = c_n == C16;
    c_n = wrap ? 0 : c_n;
    const int delta = wrap ? wrap_delta : 2 * W;
    sidx += left > 0 ? delta : 0;
    left--;
  };
  auto step = [&](int u) {  // u = static ring slot
    load_a((u + kDA - 1) % kDA, kDA - 1 + u);
    __builtin_amdgcn_sched_barrier(0);  // keep the ring's load at the top of its step (the scheduler otherwise clusters them)
    read_b((u + kDB - 1) % kDB);
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
      for (int j = 0; j < NTW; j++)
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u][m]),
                                                            __builtin_bit_cast(bf16x8, b[u % kDB][j]), acc[m][j], 0, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < kDB - 1; d++) read_b(d);
  // the ring's first loads were issued before the staging loop and have landed: an explicit vmcnt(0) gives the waitcnt pass
  // a clean state at the loop header (otherwise the first wait of each iteration inherits the staging loop's pending loads)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int full = S / kDA;
  for (int g = 0; g < full; g++) {  // guard-free groups of kDA steps
#pragma unroll
    for (int u = 0; u < kDA; u++) step(u);
    wa += kDA * 1024;
  }
  {
    const int rem = S - full * kDA;  // wave-uniform tail
#pragma unroll
    for (int u = 0; u < kDA - 1; u++)
      if (u < rem) step(u);
  }

  // ---- epilogue. accumulator register q of lane (r,h): row (q&3) + 8·(q>>2) + 4·h, column r.
  // All loads of a tile (bias, residual, MRF partners) are issued up front with clamped, always-valid indices — no
  // per-element branches, so they are one round trip instead of sixteen; only the stores are masked.
  const int rows_total = ct ? p.Cout * p.ct_stride : p.Cout;
  const int ACB = p.Cout >> 3;
#pragma unroll
  for (int m = 0; m < MTW; m++) {
    const int row0 = (mt0 + m) * 32;
    if (row0 >= rows_total) continue;
    const int co0 = ct ? row0 - rho * p.Cout : row0;  // channel of tile row 0
    float bias_v[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int coc = min(co0 + (q & 3) + 8 * (q >> 2) + 4 * h, p.Cout - 1);
      bias_v[q] = p.bias ? p.bias[coc] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NTW; j++) {
      const int col = nb0 + (wn * NTW + j) * 32 + r;
      const bool okc = col < p.Lout;
      const int colc = min(col, p.Lout - 1);
      const int pos = ct ? colc * p.ct_stride + rho : colc;
      int64_t yi[16];
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int coc = min(co0 + (q & 3) + 8 * (q >> 2) + 4 * h, p.Cout - 1);
        yi[q] = ((int64_t)n * p.Cout + coc) * p.y_len + pos;
        v[q] = acc[m][j][q] + bias_v[q];
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
      if (p.y) {
#pragma unroll
        for (int q = 0; q < 16; q++)
          if (okc && co0 + (q & 3) + 8 * (q >> 2) + 4 * h < p.Cout) p.y[yi[q]] = v[q];
      }
      if (p.act) {
        const bool live = !p.len_ptr || pos < p.len_ptr[n] * p.len_mul;  // past the true length the image must hold zeros
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int cg = co0 + 8 * g + 4 * h;  // first of 4 consecutive channels
          uint2 o;
          o.x = live ? pack2_bf16(lrelu1(v[4 * g], p.act_alpha), lrelu1(v[4 * g + 1], p.act_alpha)) : 0u;
          o.y = live ? pack2_bf16(lrelu1(v[4 * g + 2], p.act_alpha), lrelu1(v[4 * g + 3], p.act_alpha)) : 0u;
          if (okc && cg < p.Cout) {
            uint2* ap = (uint2*)((uint4*)p.act + ((int64_t)n * ACB + (cg >> 3)) * p.act_row + kC8Halo + pos) + ((cg >> 2) & 1);
            *ap = o;
          }
        }
      }
    }
  }
}

template <int MTW, int NTW, int WM>
void launch_inst(hipStream_t s, const ConvBf16Multi& a, int batch, dim3 grid, size_t lds) {
  hipLaunchKernelGGL((conv_bf16_kernel<MTW, NTW, WM>), grid, dim3(kBT), lds, s, a, batch);
}

template <int MTW, int NTW, int WM>
void raise_lds() {
  static bool done[kMaxDevices] = {};  // windows above the 64 KB default need the opt-in (160 KB per CU on gfx950), per device
  if (lds_optin_needed(done))
    (void)hipFuncSetAttribute((const void*)conv_bf16_kernel<MTW, NTW, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int launch_cfg(hipStream_t s, const ConvBf16Multi& a, int batch, int MTW, int NTW, int WM, dim3 grid, size_t lds) {
#define PH_BF16_CASE(M, NT_, W_)                                 \
  if (MTW == M && NTW == NT_ && WM == W_) {                      \
    if (lds > 64 * 1024) raise_lds<M, NT_, W_>();                \
    launch_inst<M, NT_, W_>(s, a, batch, grid, lds);             \
  } else
  PH_BF16_CASE(1, 1, 1) PH_BF16_CASE(1, 2, 1) PH_BF16_CASE(1, 4, 1)
  PH_BF16_CASE(1, 1, 2) PH_BF16_CASE(1, 2, 2) PH_BF16_CASE(1, 4, 2)
  PH_BF16_CASE(2, 1, 2) PH_BF16_CASE(2, 2, 2) PH_BF16_CASE(2, 4, 2)
  PH_BF16_CASE(1, 1, 4) PH_BF16_CASE(1, 2, 4) PH_BF16_CASE(1, 4, 4)
  PH_BF16_CASE(2, 1, 4) PH_BF16_CASE(2, 2, 4) PH_BF16_CASE(2, 4, 4)
  PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_bf16: no instance for MTW=%d NTW=%d WM=%d", MTW, NTW, WM);
#undef PH_BF16_CASE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_bf16 launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace

size_t packed_conv_bf16_elems(int Cout, int Cin, int K) { return (size_t)((Cout + 31) / 32) * K * (Cin / 16) * 64 * 8 + kBf16WeightPad; }
size_t packed_convt_bf16_elems(int Cin, int Cout, int K, int stride) {
  return (size_t)((Cout * stride + 31) / 32) * (K / stride) * (Cin / 16) * 64 * 8 + kBf16WeightPad;
}

int pack_conv_weights_bf16(hipStream_t s, const float* w, int Cout, int Cin, int K, uint16_t* packed) {
  const int64_t total = (int64_t)packed_conv_bf16_elems(Cout, Cin, K);
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBT), 4096);
  hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3(grid), dim3(kBT), 0, s, w, Cout, Cin, K, packed, total);
  return PIPER_HIP_OK;
}
int pack_convt_weights_bf16(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, int pad, uint16_t* packed) {
  const int64_t total = (int64_t)packed_convt_bf16_elems(Cin, Cout, K, stride);
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBT), 4096);
  hipLaunchKernelGGL(pack_convt_bf16_kernel, dim3(grid), dim3(kBT), 0, s, w, Cin, Cout, K, stride, pad, packed, total);
  return PIPER_HIP_OK;
}
int pack_act_c8(hipStream_t s, const float* x, int N, int C, int L, float alpha, uint16_t* act, int64_t row, const int* len_ptr) {
  if (N <= 0 || C <= 0 || L <= 0) return PIPER_HIP_OK;
  if (row <= 0) row = c8_row_len(L);
  const dim3 grid((unsigned)std::min<int64_t>(ceil_div(L, kBT), 1024), (unsigned)((C + 7) / 8), (unsigned)N);
  hipLaunchKernelGGL(pack_act_c8_kernel, grid, dim3(kBT), 0, s, x, C, L, row, alpha, act, len_ptr);
  return PIPER_HIP_OK;
}

int pack_mean3_c8(hipStream_t s, const float* a, const float* b, const float* c, int N, int C, int L, float alpha, uint16_t* act, int64_t row,
                  const int* len_ptr, int len_mul) {
  if (N <= 0 || C <= 0 || L <= 0) return PIPER_HIP_OK;
  if (row == 0) row = c8_row_len(L);
  const dim3 grid((unsigned)std::min<int64_t>(ceil_div(L, kBT), 1024), (unsigned)((C + 7) / 8), (unsigned)N);
  hipLaunchKernelGGL(pack_mean3_c8_kernel, grid, dim3(kBT), 0, s, a, b, c, C, L, row, alpha, act, len_ptr, len_mul);
  return PIPER_HIP_OK;
}

bool conv_bf16_eligible(int Cout, int Cin, int K, int dil, int padL, int padR) {
  if (Cout < 1 || Cin < 32 || (Cin & 31) || K < 1 || dil < 1) return false;
  const int reach = (K - 1) * dil;
  return padL >= 0 && padR >= 0 && padL <= kC8Halo && reach - padL <= kC8Halo && padR <= kC8Halo;
}
bool convt_bf16_eligible(int Cin, int Cout, int K, int stride, int padL, int padR, int dil, int out_pad) {
  if (Cin < 32 || (Cin & 31) || Cout < 32 || (Cout & 31) || stride < 1 || dil != 1 || out_pad != 0) return false;
  return K % stride == 0 && padL == padR && K - stride == 2 * padL;
}

int launch_conv_bf16(piper_hip_ctx* ctx, hipStream_t s, const ConvBf16Args& a) { return launch_conv_bf16_multi(ctx, s, &a, 1); }

int launch_conv_bf16_multi(piper_hip_ctx* ctx, hipStream_t s, const ConvBf16Args* convs, int count) {
  if (count < 1 || count > kBf16Multi) PH_FAIL(PIPER_HIP_ERR_ARG, "conv_bf16: %d convs in one launch (1..%d)", count, kBf16Multi);
  const ConvBf16Args& a = convs[0];
  if (a.N <= 0 || a.Lout <= 0) return PIPER_HIP_OK;
  const bool ct = a.ct_stride > 0;
  const int rows = ct ? a.Cout * a.ct_stride : a.Cout;
  const int MT = (rows + 31) / 32;
  auto reach_of = [&](const ConvBf16Args& b) {
    const int taps = ct ? b.K / b.ct_stride : b.K;
    return ct ? (b.ct_stride - 1 + b.ct_pad) / b.ct_stride + taps - 1 : (b.K - 1) * b.dil;
  };
  int reach = reach_of(a);
  for (int i = 1; i < count; i++) {
    const ConvBf16Args& b = convs[i];
    if (b.N != a.N || b.Cin != a.Cin || b.Cout != a.Cout || b.Lout != a.Lout || b.ct_stride != a.ct_stride || b.x_row != a.x_row)
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv_bf16: convs of one launch must share N, Cin, Cout, Lout, row length and kind");
    reach = std::max(reach, reach_of(b));
  }
  // waves along rows: as many of the 4 as the row-tile count allows (ConvTranspose: a wave's tiles stay in one phase)
  const int per_phase = ct ? a.Cout / 32 : MT;
  int WM = (per_phase % 4 == 0) ? 4 : (per_phase % 2 == 0) ? 2 : 1;
  int WN = 4 / WM;
  int MTW = (per_phase % (2 * WM) == 0) ? 2 : 1;
  int NTW = 4;
  auto blocks = [&](int mtw, int ntw) { return ceil_div(MT, WM * mtw) * ceil_div(a.Lout, WN * ntw * 32) * a.N * count; };
  auto lds_bytes = [&](int wn, int ntw) { return (size_t)(a.Cin / 8) * (wn * ntw * 32 + reach) * 16 + 16; };
  static const int want_blocks = [] { const char* e = getenv("PIPER_HIP_BF16_WANT_BLOCKS"); return e ? atoi(e) : 2; }();  // per CU (tuning)
  const int64_t want = want_blocks * (int64_t)ctx->num_cus;  // keep every CU busy before growing the per-wave tile
  while (NTW > 1 && (blocks(MTW, NTW) < want || lds_bytes(WN, NTW) > 160 * 1024)) NTW >>= 1;
  if (MTW == 2 && blocks(MTW, NTW) < want) MTW = 1;
  if (const char* force = getenv("PIPER_HIP_BF16_CFG")) {  // tuning hook: "MTW,NTW,WM" (ignored when it does not divide the problem)
    int m = 0, nt = 0, wmf = 0;
    if (sscanf(force, "%d,%d,%d", &m, &nt, &wmf) == 3 && (wmf == 1 || wmf == 2 || wmf == 4) && per_phase % (wmf * m) == 0 &&
        (m == 1 || (m == 2 && wmf > 1)) && (nt == 1 || nt == 2 || nt == 4) && lds_bytes(4 / wmf, nt) <= 160 * 1024) {
      MTW = m; NTW = nt; WM = wmf; WN = 4 / wmf;
    }
  }
  const size_t lds = lds_bytes(WN, NTW);
  if (lds > 160 * 1024) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_bf16: window of %zu bytes does not fit LDS (Cin=%d reach=%d)", lds, a.Cin, reach);
  const dim3 grid((unsigned)ceil_div(a.Lout, WN * NTW * 32), (unsigned)ceil_div(MT, WM * MTW), (unsigned)(a.N * count));
  ConvBf16Multi multi;
  for (int i = 0; i < kBf16Multi; i++) multi.c[i] = convs[i < count ? i : 0];
  return launch_cfg(s, multi, a.N, MTW, NTW, WM, grid, lds);
}

}  // namespace ph
namespace ph { namespace { PH_WARM(conv_bf16, pack_conv_bf16_kernel); } }
