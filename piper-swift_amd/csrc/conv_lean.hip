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

// NQ channel quads per wave; 8 waves = the whole contraction (Cin = 32 · NQ)
template <int NQ, int MODE>
__global__ __launch_bounds__(512) void conv_k1_kernel(const LeanArgs a) {
  __shared__ float red[8 * 4 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mt = blockIdx.x, t0 = blockIdx.y * 16, n = blockIdx.z;
  if (a.len_ptr) {  // bucketed / ragged batches: a chunk past the item's true length produces nothing anyone reads
    if (t0 >= a.len_ptr[n] * a.len_mul) return;
  }
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)n * a.x_bs), 0, a.x_batch_bytes, 0x00020000);
  const float* wa = a.w + (((long long)mt * a.nsteps + wave * NQ) << 6) + lane;
  const int voff = (a.kk_sign > 0 ? kk : 3 - kk) * a.x_row_bytes + min(t0 + j, a.Lin - 1) * 4;  // (columns past the row end are not stored)
  const int soff0 = a.x_base_bytes + wave * NQ * a.q_stride;
  float av[NQ], bv[NQ];
#pragma unroll
  for (int i = 0; i < NQ; i++) av[i] = wa[i * 64];
#pragma unroll
  for (int i = 0; i < NQ; i++) bv[i] = bload(rx, voff, soff0 + i * a.q_stride);
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < NQ; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave >= 4) return;
  // wave w finishes accumulator register w: rows 16·mt + 4·kk + w, column t0 + j; slices added in fixed order on top of the bias
  const int row = 16 * mt + 4 * kk + wave, col = t0 + j;
  const bool ok = row < a.Cout && col < a.Lout;
  float v = a.bias ? a.bias[min(row, a.Cout - 1)] : 0.0f;  // bias first (CPUBackend.swift:46-63)
  float part[8];
#pragma unroll
  for (int s = 0; s < 8; s++) part[s] = red[(s * 4 + wave) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 8; s++) v += part[s];
  if (!ok) return;
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
__global__ __launch_bounds__(512) void conv_gate_kernel(const GateArgs a) {
  constexpr int W = 16 + K - 1, PITCH = (W + 3) & ~3, ROWS = 4 * NQ, NS = NQ * K;
  constexpr int NLB = (ROWS + 15) / 16;  // halo loads: 16 rows × 4 columns each (K − 1 ≤ 4)
  constexpr int OOB = 0x7fffffff;
  static_assert(K - 1 <= 4, "halo piece is four columns wide");
  __shared__ float xs_all[8 * ROWS * PITCH];
  __shared__ float red[8 * 4 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int mt = blockIdx.x, t0 = blockIdx.y * 16, n = blockIdx.z;
  int Lv = a.Lin;
  if (a.len_ptr) {
    Lv = min(a.len_ptr[n] * a.len_mul, a.Lin);
    if (t0 >= Lv) return;  // a chunk past the item's true length produces nothing anyone reads
  }
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)n * a.x_bs), 0, a.x_batch_bytes, 0x00020000);
  const float* wa = a.w + (((long long)mt * a.nsteps + wave * NS) << 6) + lane;
  float av[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) av[s] = wa[s * 64];
  // window: main piece (column j of rows 4i + kk), halo piece (column 16 + (lane & 3) of rows 16i + (lane >> 2))
  const int posA = t0 - a.pad + j;
  const int voffA = (posA >= 0 && posA < Lv) ? (kk * a.Lin + posA) * 4 : OOB;
  const int rB = lane >> 2, cB = 16 + (lane & 3);
  const int posB = t0 - a.pad + cB;
  const int voffB = ((lane & 3) < K - 1 && posB >= 0 && posB < Lv) ? (rB * a.Lin + posB) * 4 : OOB;
  const int sbase = wave * ROWS * a.Lin * 4;
  float xa[NQ], xb[NLB];
#pragma unroll
  for (int i = 0; i < NQ; i++) xa[i] = bload(rx, voffA, sbase + i * 16 * a.Lin);
#pragma unroll
  for (int i = 0; i < NLB; i++) xb[i] = bload(rx, (rB + 16 * i < ROWS) ? voffB : OOB, sbase + i * 64 * a.Lin);
  float* xs = xs_all + wave * (ROWS * PITCH);
#pragma unroll
  for (int i = 0; i < NQ; i++) xs[(4 * i + kk) * PITCH + j] = xa[i];
#pragma unroll
  for (int i = 0; i < NLB; i++)
    if ((lane & 3) < K - 1 && rB + 16 * i < ROWS) xs[(rB + 16 * i) * PITCH + cB] = xb[i];
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
  // wave w finishes register w: tile rows i = 4·kk + w — lanes 0–31 hold tanh rows (i < 8), lanes 32–63 their sigmoid partners
  const int i = 4 * kk + wave;
  const int h = 8 * mt + (i & 7);
  float v = a.bias ? a.bias[(i < 8 ? 0 : a.rows_out) + min(h, a.rows_out - 1)] : 0.0f;  // bias first (CPUBackend.swift:46-63)
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

template <int NQ>
bool launch_lean_nq(hipStream_t s, dim3 grid, const LeanArgs& a, int mode) {
  switch (mode) {
    case EPI_STORE: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_STORE>), grid, dim3(512), 0, s, a); return true;
    case EPI_RSUB: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_RSUB>), grid, dim3(512), 0, s, a); return true;
    case EPI_WN_RES_SKIP: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_WN_RES_SKIP>), grid, dim3(512), 0, s, a); return true;
    case EPI_WN_SKIP_LAST: hipLaunchKernelGGL((conv_k1_kernel<NQ, EPI_WN_SKIP_LAST>), grid, dim3(512), 0, s, a); return true;
  }
  return false;
}

}  // namespace

// 1 = enqueued, 0 = not this kernel's case, < 0 = error. Called by launch_conv_mfma ahead of the general short-row kernels.
int try_launch_conv_lean(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& c) {
  static const bool off = getenv("PIPER_HIP_NO_LEAN") != nullptr;
  if (off) return 0;
  if (c.gate) {  // the flow's gated conv: k 5 (3), 192 (96) channels, identity channel maps
    if (!c.w16g || c.prologue != PRO_NONE || c.epilogue != EPI_STORE || c.res || c.stats_out || c.dil != 1 || c.Lin != c.Lout || (c.Cout % 16)) return 0;
    if (c.in_ch_sign != 1 || c.in_ch_base != 0 || c.out_ch_sign != 1 || c.out_ch_base != 0 || c.N > 65535) return 0;
    if ((c.K != 5 && c.K != 3) || c.Cin != 192 || c.padL != (c.K - 1) / 2) return 0;
    const int mt_g = c.Cout / 16, nch = (int)ceil_div(c.Lout, 16);
    if ((int64_t)mt_g * nch * c.N > 8 * (int64_t)ctx->num_cus || nch > 65535 || c.x_batch_stride * 4 >= 0x7fffffffLL) return 0;
    GateArgs g;
    g.x = c.x; g.w = c.w16g; g.bias = c.bias; g.y = c.y; g.len_ptr = c.len_ptr; g.len_mul = c.len_mul;
    g.Lin = c.Lin; g.Lout = c.Lout; g.rows_out = c.Cout / 2; g.y_len = c.y_len; g.nsteps = (c.Cin / 4) * c.K; g.pad = c.padL;
    g.x_batch_bytes = (int)(c.x_batch_stride * 4); g.x_bs = c.x_batch_stride; g.y_bs = c.y_batch_stride;
    const dim3 grid((unsigned)mt_g, (unsigned)nch, (unsigned)c.N);
    if (c.K == 5) hipLaunchKernelGGL((conv_gate_kernel<5, 6>), grid, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((conv_gate_kernel<3, 6>), grid, dim3(512), 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_gate launch failed: %s", hipGetErrorString(e));
    return 1;
  }
  if (c.K != 1 || c.prologue != PRO_NONE || c.stats_out || !c.w16) return 0;
  if (c.epilogue != EPI_STORE && c.epilogue != EPI_RSUB && c.epilogue != EPI_WN_RES_SKIP && c.epilogue != EPI_WN_SKIP_LAST) return 0;
  if ((c.epilogue == EPI_RSUB || c.epilogue == EPI_WN_RES_SKIP) && !c.res) return 0;
  if (c.Cin % 32 || c.padL != 0 || c.Lin != c.Lout || c.N > 65535 || c.Lout < 1) return 0;
  const int NQ = c.Cin / 32;
  if (NQ != 3 && NQ != 6) return 0;
  if (c.in_ch_sign != 1 && c.in_ch_sign != -1) return 0;
  const int mtiles = (int)ceil_div(c.Cout, 16), nchunks = (int)ceil_div(c.Lout, 16);
  if ((int64_t)mtiles * nchunks * c.N > 8 * (int64_t)ctx->num_cus) return 0;  // enough tiles for the kernels that reuse operands
  if (nchunks > 65535 || c.x_batch_stride * 4 >= 0x7fffffffLL) return 0;
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
  const dim3 grid((unsigned)mtiles, (unsigned)nchunks, (unsigned)c.N);
  const bool ok = NQ == 3 ? launch_lean_nq<3>(s, grid, a, c.epilogue) : launch_lean_nq<6>(s, grid, a, c.epilogue);
  if (!ok) return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_lean launch failed: %s", hipGetErrorString(e));
  return 1;
}

}  // namespace ph
