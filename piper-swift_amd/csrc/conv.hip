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

namespace ph {
namespace {

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

template <int PRO>
__device__ __forceinline__ float load_b(const ConvArgs& p, const float* xrow, const float* x2row, const float* x3row, int pos,
                                        bool ch_ok) {
  if (!ch_ok || pos < 0 || pos >= p.Lin) return 0.0f;
  float v = xrow[pos];
  if constexpr (PRO == PRO_AVG3_LRELU) v = ((v + x2row[pos]) + x3row[pos]) / 3.0f;
  if constexpr (PRO != PRO_NONE) v = lrelu(v, p.alpha);
  return v;
}

__device__ __forceinline__ void store_elem(const ConvArgs& p, int n, int row, int col, float v) {
  switch (p.epilogue) {
    case EPI_STORE:
    case EPI_RELU:
    case EPI_TANH:
    case EPI_RSUB: {
      const int64_t idx = (int64_t)n * p.y_batch_stride + (int64_t)(p.out_ch_base + p.out_ch_sign * row) * p.y_len + col;
      if (p.epilogue == EPI_RELU) v = v > 0.0f ? v : 0.0f;
      else if (p.epilogue == EPI_TANH) v = tanhf(v);
      else if (p.epilogue == EPI_RSUB) v = p.res[idx] - v;
      else if (p.res) v = v + p.res[idx];
      p.y[idx] = v;
      break;
    }
    case EPI_WN_RES_SKIP: {
      if (row < p.wn_c) {
        const int64_t idx = (int64_t)n * p.y_batch_stride + (int64_t)row * p.y_len + col;
        p.y[idx] = p.res[idx] + v;
      } else {
        const int64_t idx = (int64_t)n * p.y2_batch_stride + (int64_t)(row - p.wn_c) * p.y_len + col;
        p.y2[idx] = (p.skip ? p.skip[idx] : 0.0f) + v;
      }
      break;
    }
    case EPI_WN_SKIP_LAST: {
      const int64_t idx = (int64_t)n * p.y2_batch_stride + (int64_t)row * p.y_len + col;
      p.y2[idx] = (p.skip ? p.skip[idx] : 0.0f) + v;
      break;
    }
    case EPI_CONVT: {
      const int co = row / p.ct_stride, ph = row - co * p.ct_stride;
      const int xo = col * p.ct_stride + ph - p.ct_padL;
      if (xo >= 0 && xo < p.ct_Lout) p.y[(int64_t)n * p.y_batch_stride + (int64_t)co * p.y_len + xo] = v;
      break;
    }
  }
}

template <int NT, int KS, bool GATE, int PRO>
__global__ __launch_bounds__(kBlock) void conv_mfma_kernel(const ConvArgs p, const int nchunks, const int mtiles) {
  constexpr int WT = 4 / KS;
  constexpr int NA = GATE ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [KS-1][WT][NA][NT][16][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tw = wave / KS, ks = wave - tw * KS;
  const int n = blockIdx.y;
  const int mt_eff = GATE ? mtiles / 2 : mtiles;
  const int64_t tile = (int64_t)blockIdx.x * WT + tw;
  const bool active = tile < (int64_t)mt_eff * nchunks;
  const int mt = active ? (int)(tile % mt_eff) : 0;
  const int chunk = active ? (int)(tile / mt_eff) : 0;
  const int t0 = chunk * 32 * NT;
  const int j = lane & 31, kk = lane >> 5;
  const int ncp = (p.Cin + 1) >> 1;
  const int nsteps = ncp * p.K;
  const int brow = p.ct_stride > 0 ? p.ct_stride : 1;  // bias index = row / brow

  f32x16 acc[NA][NT];
#pragma unroll
  for (int a = 0; a < NA; a++) {
    const int mbase = (a == 0 ? mt : mt + mt_eff) * 32;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = mbase + acc_row(r, lane);
      const float b = (ks == 0 && p.bias && row < p.Cout) ? p.bias[row / brow] : 0.0f;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[a][nt][r] = b;
    }
  }

  if (active) {
    const int cp_begin = (int)((int64_t)ncp * ks / KS), cp_end = (int)((int64_t)ncp * (ks + 1) / KS);
    const float* xb = p.x + (int64_t)n * p.x_batch_stride;
    const float* x2b = PRO == PRO_AVG3_LRELU ? p.x2 + (int64_t)n * p.x_batch_stride : nullptr;
    const float* x3b = PRO == PRO_AVG3_LRELU ? p.x3 + (int64_t)n * p.x_batch_stride : nullptr;
    const float* wa = p.w + ((int64_t)mt * nsteps) * 64 + lane;
    const float* wb = GATE ? p.w + ((int64_t)(mt + mt_eff) * nsteps) * 64 + lane : nullptr;
    for (int cp = cp_begin; cp < cp_end; cp++) {
      const int ci = 2 * cp + kk;
      const bool ch_ok = ci < p.Cin;
      const int64_t roff = (int64_t)(p.in_ch_base + p.in_ch_sign * (ch_ok ? ci : 0)) * p.Lin;
      const float* xrow = xb + roff;
      const float* x2row = PRO == PRO_AVG3_LRELU ? x2b + roff : nullptr;
      const float* x3row = PRO == PRO_AVG3_LRELU ? x3b + roff : nullptr;
      const int step0 = cp * p.K;
      for (int tap = 0; tap < p.K; tap++) {
        const float a0 = wa[(int64_t)(step0 + tap) * 64];
        float a1 = 0.0f;
        if constexpr (GATE) a1 = wb[(int64_t)(step0 + tap) * 64];
        const int base = t0 + j + tap * p.dil - p.padL;
        float bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) bv[nt] = load_b<PRO>(p, xrow, x2row, x3row, base + 32 * nt, ch_ok);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[nt], acc[0][nt], 0, 0, 0);
          if constexpr (GATE) acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[nt], acc[1][nt], 0, 0, 0);
        }
      }
    }
  }

  if constexpr (KS > 1) {
    // fixed-order reduction: slice 0 + slice 1 + … (deterministic)
    constexpr int per_wave = NA * NT * 16 * 64;
    if (ks > 0) {
      float* dst = red + ((int64_t)((ks - 1) * WT + tw)) * per_wave + lane;
#pragma unroll
      for (int a = 0; a < NA; a++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < 16; r++) dst[((a * NT + nt) * 16 + r) * 64] = acc[a][nt][r];
    }
    __syncthreads();
    if (ks == 0) {
#pragma unroll
      for (int s = 1; s < KS; s++) {
        const float* src = red + ((int64_t)((s - 1) * WT + tw)) * per_wave + lane;
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][nt][r] += src[((a * NT + nt) * 16 + r) * 64];
      }
    }
  }

  if (!active || ks != 0) return;
  const int rows_out = GATE ? p.Cout / 2 : p.Cout;
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    const int col = t0 + 32 * nt + j;
    if (col >= p.Lout) continue;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = mt * 32 + acc_row(r, lane);
      if (row >= rows_out) continue;
      float v = acc[0][nt][r];
      if constexpr (GATE) v = tanhf(v) * sigmoid_stable(acc[1][nt][r]);
      store_elem(p, n, row, col, v);
    }
  }
}

// ---- weight packing (once per voice; per call for the op-level API) ----
__global__ __launch_bounds__(kBlock) void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                           int K, int mtiles, int nsteps) {
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int lane = (int)(i & 63);
    const int64_t ms = i >> 6;
    const int step = (int)(ms % nsteps), mt = (int)(ms / nsteps);
    const int cp = step / K, tap = step - cp * K;
    const int co = mt * 32 + (lane & 31), ci = 2 * cp + (lane >> 5);
    out[i] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * K + tap] : 0.0f;
  }
}

// ConvTranspose [Cin, Cout, K], stride s → GEMM rows (co, phase), taps j: weight W[ci][co][phase + s·j]
__global__ __launch_bounds__(kBlock) void pack_convt_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout,
                                                            int K, int s, int J, int mtiles, int nsteps) {
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  const int rows = Cout * s;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int lane = (int)(i & 63);
    const int64_t ms = i >> 6;
    const int step = (int)(ms % nsteps), mt = (int)(ms / nsteps);
    const int cp = step / J, jt = step - cp * J;
    const int row = mt * 32 + (lane & 31), ci = 2 * cp + (lane >> 5);
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
    for (int ci = 0; ci < cig; ci++) {
      const int64_t roff = (int64_t)(p.in_ch_base + p.in_ch_sign * (ciBase + ci)) * p.Lin;
      for (int k = 0; k < p.K; k++) {
        const int pos = inX0 + k * p.dil;
        const float v = load_b<PRO>(p, xb + roff, PRO == PRO_AVG3_LRELU ? x2b + roff : nullptr,
                                    PRO == PRO_AVG3_LRELU ? x3b + roff : nullptr, pos, true);
        acc += v * wr[ci * p.K + k];
      }
    }
    store_elem(p, n, co, xo, acc);
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

template <int NT, int KS, bool GATE>
void launch_variant(hipStream_t s, const ConvArgs& a, int nchunks, int mtiles, dim3 grid, size_t lds) {
  switch (a.prologue) {
    case PRO_NONE:
      hipLaunchKernelGGL((conv_mfma_kernel<NT, KS, GATE, PRO_NONE>), grid, dim3(kBlock), lds, s, a, nchunks, mtiles);
      break;
    case PRO_LRELU:
      hipLaunchKernelGGL((conv_mfma_kernel<NT, KS, GATE, PRO_LRELU>), grid, dim3(kBlock), lds, s, a, nchunks, mtiles);
      break;
    default:
      hipLaunchKernelGGL((conv_mfma_kernel<NT, KS, GATE, PRO_AVG3_LRELU>), grid, dim3(kBlock), lds, s, a, nchunks, mtiles);
      break;
  }
}

}  // namespace

size_t packed_conv_floats(int Cout, int Cin, int K) {
  return (size_t)ceil_div(Cout, 32) * (size_t)(((Cin + 1) / 2) * K) * 64;
}
size_t packed_convt_floats(int Cin, int Cout, int K, int s) {
  const int J = (K + s - 1) / s;
  return (size_t)ceil_div((int64_t)Cout * s, 32) * (size_t)(((Cin + 1) / 2) * J) * 64;
}

int pack_conv_weights(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed) {
  const int mtiles = (int)ceil_div(Cout, 32), nsteps = ((Cin + 1) / 2) * K;
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  if (total == 0) return PIPER_HIP_OK;
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), 4096);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid), dim3(kBlock), 0, s, w, packed, Cout, Cin, K, mtiles, nsteps);
  return PIPER_HIP_OK;
}

int pack_convt_weights(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, float* packed) {
  const int J = (K + stride - 1) / stride;
  const int mtiles = (int)ceil_div((int64_t)Cout * stride, 32), nsteps = ((Cin + 1) / 2) * J;
  const int64_t total = (int64_t)mtiles * nsteps * 64;
  if (total == 0) return PIPER_HIP_OK;
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBlock), 4096);
  hipLaunchKernelGGL(pack_convt_kernel, dim3(grid), dim3(kBlock), 0, s, w, packed, Cin, Cout, K, stride, J, mtiles, nsteps);
  return PIPER_HIP_OK;
}

bool conv_mfma_eligible(int Cout, int Cin, int K, int stride, int groups) {
  return stride == 1 && groups == 1 && Cout >= 8 && Cin >= 2 && K >= 1;
}

int launch_conv_mfma(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a) {
  if (a.N <= 0 || a.Lout <= 0 || a.Cout <= 0) return PIPER_HIP_OK;
  if (a.N > 65535) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv: batch %d too large", a.N);
  if (a.gate && (a.Cout % 64)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "gated conv needs Cout %% 64 == 0 (got %d)", a.Cout);
  const int mtiles = (int)ceil_div(a.Cout, 32);
  const int mt_eff = a.gate ? mtiles / 2 : mtiles;
  const int ncp = (a.Cin + 1) / 2;
  // tile shape: give every SIMD (4 per CU) a wave before growing the per-wave tile
  const int64_t want = (int64_t)ctx->num_cus * 4;
  int NT = 4;
  auto waves = [&](int nt) { return (int64_t)mt_eff * ceil_div(a.Lout, 32 * nt) * a.N; };
  while (NT > 1 && waves(NT) < want) NT >>= 1;
  if (a.gate && NT > 2) NT = 2;  // 2 accumulator sets per time tile
  int KS = 1;
  while (KS < 4 && waves(NT) * KS < want && ncp / (KS * 2) >= 8) KS <<= 1;
  const int nchunks = (int)ceil_div(a.Lout, 32 * NT);
  const int WT = 4 / KS;
  const int64_t tiles = (int64_t)mt_eff * nchunks;
  dim3 grid((unsigned)ceil_div(tiles, WT), (unsigned)a.N);
  const int NA = a.gate ? 2 : 1;
  const size_t lds = KS > 1 ? (size_t)(KS - 1) * WT * NA * NT * 16 * 64 * sizeof(float) : 0;
#define PH_CASE(NTV, KSV)                                                                      \
  if (NT == NTV && KS == KSV) {                                                                \
    if (a.gate) launch_variant<NTV, KSV, true>(s, a, nchunks, mtiles, grid, lds);              \
    else launch_variant<NTV, KSV, false>(s, a, nchunks, mtiles, grid, lds);                    \
  }
  PH_CASE(1, 1) PH_CASE(1, 2) PH_CASE(1, 4) PH_CASE(2, 1) PH_CASE(2, 2) PH_CASE(2, 4) PH_CASE(4, 1)
#undef PH_CASE
  if (NT == 4 && KS != 1) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv: internal tiling error");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_mfma launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

int launch_conv_direct(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a) {
  const int64_t total = (int64_t)a.N * a.Cout * a.Lout;
  if (total <= 0) return PIPER_HIP_OK;
  if (a.epilogue == EPI_WN_RES_SKIP || a.epilogue == EPI_WN_SKIP_LAST || a.epilogue == EPI_CONVT || a.gate)
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "direct conv does not implement epilogue %d", a.epilogue);
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
