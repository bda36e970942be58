// rb_pair_bf16.hip — a ResBlock1 pair of the HiFi-GAN generator with bf16 operands in ONE kernel:
//     y = x + conv_b(lrelu(conv_a(lrelu(x)) + b_a)) + b_b          (convs1[i] dilated, convs2[i] dilation 1; Piper "high")
// The bf16 twin of rb_pair.hip (same tiling: 256 x1 columns = 224 output columns + a 16-column halo each side; 4 wave
// columns × 2 column tiles) on v_mfma_f32_32x32x16_bf16 with the operand layouts of conv_bf16.hip:
//   * the fp32 residual stream x is read once, LeakyReLU'd, rounded to bf16 and laid out in LDS as "C8" — [C/8][positions] of
//     16-byte entries holding 8 consecutive channels — so a lane's B fragment is one ds_read_b128;
//   * conv a's result (+ bias, LeakyReLU, bf16, zero outside [0, len)) goes to a second C8 image in LDS: the rounding points
//     are exactly those of the two-launch path (conv_bf16_kernel writes the same bf16 image to HBM), so fused == unfused up
//     to fp32 summation order;
//   * conv b's result + bias + the fp32 x → y (fp32) and, when the next conv is not fused, the C8 image of lrelu(y) in HBM.
// Per pair this removes one launch, the bf16 image round trip of the intermediate and one pass over the fp32 stream.
#include <algorithm>
#include <type_traits>

#include "conv_bf16.h"
#include "conv_win.h"

namespace ph {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWN = 4, kNTW = 2, kT1 = kWN * kNTW;
constexpr int kColsA = 32 * kT1, kHalo = 16, kColsB = kColsA - 2 * kHalo;  // 256 x1 columns, 224 output columns
constexpr int kDA = 8, kDB = 4;                                              // weight ring / LDS fragment ring depths (steps)

__device__ __forceinline__ unsigned pack2(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lrelu1(float v, float alpha) { return v >= 0.0f ? v : v * alpha; }  // conv_bf16.hip's form
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, float v, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}

struct RbPairBf16Multi {
  RbPairBf16Args c[kWinMulti];
};

// C = 32·MT channels; a wave owns MTW row tiles × kNTW column tiles; the block has (MT / MTW) · kWN waves.
template <int MT, int MTW>
__global__ __launch_bounds__((MT / MTW) * kWN * 64, 2) void rb_pair_bf16_kernel(const RbPairBf16Multi multi, const int batch, const int order, const int Wx,
                                                                              const int W1) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  constexpr int WMB = MT / MTW, BT = WMB * kWN * 64;
  constexpr int C = 32 * MT, CB = C / 8, C16 = C / 16;
  const int jz = blockIdx.y / batch;
  const RbPairBf16Args& p = multi.c[(order >> (4 * jz)) & 15];  // heaviest pair first
  const int n = blockIdx.y - jz * batch;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave % WMB, wn = wave / WMB;
  const int mt0 = wm * MTW;
  const int r = lane & 31, h = lane >> 5;
  const int pa = (p.Ka - 1) * p.dila / 2, pb = (p.Kb - 1) * p.dilb / 2;
  const int c0 = blockIdx.x * kColsB;
  const int Lv = p.len_ptr ? min(p.len_ptr[n] * p.len_mul, p.L) : p.L;
  if (c0 >= Lv) {  // nothing anyone reads — except the activation image, whose "zero past the true length" invariant must hold
    if (p.act && c0 < p.L) {
      for (int i = threadIdx.x; i < CB * kColsB; i += BT) {
        const int cb = i / kColsB, g = c0 + (i - cb * kColsB);
        if (g < p.L) ((uint4*)p.act)[((int64_t)n * CB + cb) * p.act_row + kC8Halo + g] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    return;
  }
  const int g0 = c0 - kHalo - pa;  // input position of window column `shift`
  const int ga = g0 & ~3;
  const int shift = g0 - ga;
  uint4* xs = lds;                          // lrelu(x) as bf16 C8 [CB][Wx]
  uint4* x1s = lds + CB * Wx + 1;           // lrelu(x1) as bf16 C8 [CB][W1]   (+1: the staging dump entry)
  float* biasS = (float*)(x1s + CB * W1);   // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += BT) biasS[i] = i < C ? p.ba[i] : p.bb[i - C];
  const float alpha = p.alpha;

  // ---- weight ring (uniform base + 32-bit lane offset); conv a's first steps are requested before the window exists
  const int Sa = p.Ka * C16, Sb = p.Kb * C16;
  const unsigned lane16 = (unsigned)lane * 16u;
  uint4 a[kDA][MTW];
  const char* wa = nullptr;
  int64_t wtile = 0;
  auto load_a = [&](int slot, int ahead) {
#pragma unroll
    for (int m = 0; m < MTW; m++) a[slot][m] = *(const uint4*)(wa + m * wtile + ahead * 1024 + lane16);
  };
  auto ring_start = [&](const uint16_t* w, int S) {
    wtile = (int64_t)S * 1024;
    wa = (const char*)w + (int64_t)mt0 * wtile;
#pragma unroll
    for (int d = 0; d < kDA - 1; d++) load_a(d, d);
  };
  ring_start(p.wa, Sa);

  // ---- stage lrelu(x) → bf16 C8. item = (channel block cb, 4 consecutive positions): 8 aligned float4 loads (the block's 8 channel
  // rows), 4 entries written. Two items per thread and pass: 16 loads in flight.
  {
    const float* xb = p.x + (int64_t)n * C * p.L;
    const int W4 = Wx >> 2, items = CB * W4, dump = CB * Wx;
    for (int base = threadIdx.x; base < items; base += 2 * BT) {
      float4 t[2][8];
      int dst[2], nv[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = base + u * BT;
        const int ic = min(i, items - 1);
        const int cb = ic / W4, p4 = ic - cb * W4;
        const int pos = ga + 4 * p4;
        const bool inb = pos >= 0 && pos < p.L;
        nv[u] = inb ? Lv - pos : 0;
        dst[u] = i < items ? cb * Wx + 4 * p4 : dump;
#pragma unroll
        for (int e = 0; e < 8; e++) t[u][e] = *(const float4*)(xb + (int64_t)(cb * 8 + e) * p.L + (inb ? pos : 0));
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const float* f = (const float*)&t[u][0];  // f[4·e + k]: channel e, position k
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint4 o;
          const bool ok = nv[u] > k;
          o.x = ok ? pack2(lrelu1(f[0 + k], alpha), lrelu1(f[4 + k], alpha)) : 0u;
          o.y = ok ? pack2(lrelu1(f[8 + k], alpha), lrelu1(f[12 + k], alpha)) : 0u;
          o.z = ok ? pack2(lrelu1(f[16 + k], alpha), lrelu1(f[20 + k], alpha)) : 0u;
          o.w = ok ? pack2(lrelu1(f[24 + k], alpha), lrelu1(f[28 + k], alpha)) : 0u;
          if (dst[u] != dump || k == 0) xs[dst[u] + (dst[u] != dump ? k : 0)] = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- one conv over this wave's tiles: B fragments by ds_read_b128 from a C8 image [CB][Wrow], A through the ring
  f32x16 acc[MTW][kNTW];
  auto run_conv = [&](auto nt_tag, const uint4* img, const int Wrow, const int S, const int dil, const int col0, const int bias_off) {
    constexpr int NT = decltype(nt_tag)::value;
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {  // accumulators start at the bias; register q of lane (r,h): row (q&3) + 8·(q>>2) + 4·h
        const float4 bv = *(const float4*)(biasS + bias_off + (mt0 + m) * 32 + 4 * h + 8 * g4);
#pragma unroll
        for (int j = 0; j < kNTW; j++) { acc[m][j][4 * g4] = bv.x; acc[m][j][4 * g4 + 1] = bv.y; acc[m][j][4 * g4 + 2] = bv.z; acc[m][j][4 * g4 + 3] = bv.w; }
      }
    const int lbase = h * Wrow + col0 + r;
    uint4 b[kDB][NT];
    int sidx = 0, c_n = 0, left = S - 1;  // step = tap·C16 + c16: the scalar part of the index, 2·c16·Wrow + tap·dil
    const int wrap_delta = dil - 2 * Wrow * (C16 - 1);
    auto read_b = [&](int slot) {
      const int idx = lbase + sidx;
#pragma unroll
      for (int j = 0; j < NT; j++) b[slot][j] = img[idx + 32 * j];
      c_n++;
      const bool wrap = c_n == C16;
      c_n = wrap ? 0 : c_n;
      const int delta = wrap ? wrap_delta : 2 * Wrow;
      sidx += left > 0 ? delta : 0;
      left--;
    };
    auto step = [&](int u) {
      load_a((u + kDA - 1) % kDA, kDA - 1 + u);
      read_b((u + kDB - 1) % kDB);
      __builtin_amdgcn_sched_barrier(0);  // ring load and the LDS read stay ahead of this step's MFMAs
#pragma unroll
      for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int j = 0; j < NT; j++)
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u][m]), __builtin_bit_cast(bf16x8, b[u % kDB][j]), acc[m][j], 0,
                                                              0, 0);
    };
#pragma unroll
    for (int d = 0; d < kDB - 1; d++) read_b(d);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): clean state for the loop's counted waits
    const int full = S / kDA;
    for (int g = 0; g < full; g++) {
#pragma unroll
      for (int u = 0; u < kDA; u++) step(u);
      wa += kDA * 1024;
    }
    const int rem = S - full * kDA;
#pragma unroll
    for (int u = 0; u < kDA - 1; u++)
      if (u < rem) step(u);
  };

  // ======== conv a: x1 columns [c0 − 16, c0 + 240)
  run_conv(std::integral_constant<int, kNTW>{}, xs, Wx, Sa, p.dila, shift + wn * kNTW * 32, 0);
  ring_start(p.wb, Sb);  // conv b's ring: lands behind the epilogue and the barrier
  {  // lrelu(x1) as bf16 → x1s; a lane holds 4 consecutive channels (rows 8g + 4h …) of one column: 8 bytes of an entry
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
      for (int j = 0; j < kNTW; j++) {
        const int colw = (wn * kNTW + j) * 32 + r;
        const int g = c0 - kHalo + colw;
        const bool in = g >= 0 && g < Lv;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          uint2 o;
          o.x = in ? pack2(lrelu1(acc[m][j][4 * g4], alpha), lrelu1(acc[m][j][4 * g4 + 1], alpha)) : 0u;
          o.y = in ? pack2(lrelu1(acc[m][j][4 * g4 + 2], alpha), lrelu1(acc[m][j][4 * g4 + 3], alpha)) : 0u;
          ((uint2*)(x1s + ((mt0 + m) * 4 + g4) * W1 + colw))[h] = o;
        }
      }
  }
  __syncthreads();

  // ======== conv b: the 7 output tiles, two per wave column (the last one gets one)
  const int ntb = min(max((kColsB >> 5) - wn * kNTW, 0), kNTW);  // wave-uniform
  if (ntb == 0) return;
  const int col0b = kHalo - pb + wn * kNTW * 32;
  if (ntb == 2) run_conv(std::integral_constant<int, 2>{}, x1s, W1, Sb, p.dilb, col0b, C);
  else run_conv(std::integral_constant<int, 1>{}, x1s, W1, Sb, p.dilb, col0b, C);

  {  // y = x + acc (bias inside) → fp32; optionally the C8 image of lrelu(y) for a conv that is not fused
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * C * p.L), 0, C * p.L * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + (int64_t)n * C * p.L : nullptr), 0, p.y ? C * p.L * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rma = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mrf_a ? p.mrf_a + (int64_t)n * C * p.L : nullptr), 0, p.mrf_a ? C * p.L * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mrf_b ? p.mrf_b + (int64_t)n * C * p.L : nullptr), 0, p.mrf_b ? C * p.L * 4 : 0, 0x00020000);
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
      for (int j = 0; j < kNTW; j++) {
        if (j >= ntb) break;
        const int colo = (wn * kNTW + j) * 32 + r;
        const int g = c0 + colo;
        const int rowl = (mt0 + m) * 32 + 4 * h;
        const int voff = g < p.L ? (rowl * p.L + g) * 4 : -4;  // −4: out of range ⇒ load gives 0, store is dropped
        float xr[16];
#pragma unroll
        for (int q = 0; q < 16; q++) xr[q] = bload(rx, voff, ((q & 3) + 8 * (q >> 2)) * p.L * 4);
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = acc[m][j][q] + xr[q];
        if (p.mrf_a) {  // ((mrf_a + mrf_b) + v) / 3: the association of the graph's Add, Add, Div
          float ta[16], tb[16];
#pragma unroll
          for (int q = 0; q < 16; q++) {
            ta[q] = bload(rma, voff, ((q & 3) + 8 * (q >> 2)) * p.L * 4);
            tb[q] = bload(rmb, voff, ((q & 3) + 8 * (q >> 2)) * p.L * 4);
          }
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] = ((ta[q] + tb[q]) + v[q]) / 3.0f;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) bstore(ry, v[q], voff, ((q & 3) + 8 * (q >> 2)) * p.L * 4);  // no y: zero-sized descriptor, stores dropped
        if (p.act && g < p.L) {
          const bool live = g < Lv;  // past the true length the image must hold zeros
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            uint2 o;
            o.x = live ? pack2(lrelu1(v[4 * g4], alpha), lrelu1(v[4 * g4 + 1], alpha)) : 0u;
            o.y = live ? pack2(lrelu1(v[4 * g4 + 2], alpha), lrelu1(v[4 * g4 + 3], alpha)) : 0u;
            ((uint2*)((uint4*)p.act + ((int64_t)n * CB + (mt0 + m) * 4 + g4) * p.act_row + kC8Halo + g))[h] = o;
          }
        }
      }
  }
}

struct PairGeom {
  int Wx, W1;
  size_t lds;
};
PairGeom pair_geom(int C, int pa_max) {
  PairGeom g;
  g.Wx = (kColsA + 2 * pa_max + 3 + 3) & ~3;
  g.W1 = kColsA;
  g.lds = ((size_t)(C / 8) * g.Wx + 1 + (size_t)(C / 8) * g.W1) * 16 + (size_t)2 * C * sizeof(float);
  return g;
}

template <int MT, int MTW>
void launch_inst(hipStream_t s, const RbPairBf16Multi& m, int batch, int order, const PairGeom& g, dim3 grid) {
  static bool raised[kMaxDevices] = {};
  if (g.lds > 64 * 1024 && lds_optin_needed(raised))
    (void)hipFuncSetAttribute((const void*)rb_pair_bf16_kernel<MT, MTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((rb_pair_bf16_kernel<MT, MTW>), grid, dim3((MT / MTW) * kWN * 64), g.lds, s, m, batch, order, g.Wx, g.W1);
}

}  // namespace

bool rb_pair_bf16_eligible(int C, int Ka, int dila, int Kb, int dilb, int L) {
  if (C != 32 && C != 64 && C != 128) return false;
  if (Ka < 1 || Kb < 1 || !(Ka & 1) || !(Kb & 1) || dila < 1 || dilb < 1 || L < 4 || (L & 3)) return false;
  const int pa = (Ka - 1) * dila / 2, pb = (Kb - 1) * dilb / 2;
  if (pb > kHalo || pa > kC8Halo) return false;
  return pair_geom(C, pa).lds <= 160 * 1024;
}

int launch_rb_pair_bf16_multi(piper_hip_ctx* ctx, hipStream_t s, const RbPairBf16Args* pairs, int count) {
  (void)ctx;
  if (count < 1 || count > kWinMulti) PH_FAIL(PIPER_HIP_ERR_ARG, "rb_pair_bf16: %d pairs in one launch (1..%d)", count, kWinMulti);
  const RbPairBf16Args& a = pairs[0];
  if (a.N <= 0 || a.L <= 0) return PIPER_HIP_OK;
  int pa_max = 0, idx[kWinMulti] = {0, 1, 2};
  for (int i = 0; i < count; i++) {
    const RbPairBf16Args& b = pairs[i];
    if (b.N != a.N || b.C != a.C || b.L != a.L) PH_FAIL(PIPER_HIP_ERR_SHAPE, "rb_pair_bf16: pairs of one launch must share N, C and L");
    if (!rb_pair_bf16_eligible(b.C, b.Ka, b.dila, b.Kb, b.dilb, b.L))
      PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rb_pair_bf16: geometry not covered (C=%d K=%d,%d d=%d,%d L=%d)", b.C, b.Ka, b.Kb, b.dila, b.dilb, b.L);
    if (!b.x || (!b.y && !b.act) || !b.wa || !b.wb || !b.ba || !b.bb || (!b.mrf_a != !b.mrf_b)) PH_FAIL(PIPER_HIP_ERR_ARG, "rb_pair_bf16: null operand");
    pa_max = std::max(pa_max, (b.Ka - 1) * b.dila / 2);
  }
  const PairGeom g = pair_geom(a.C, pa_max);
  if (g.lds > 160 * 1024) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "rb_pair_bf16: window of %zu bytes exceeds LDS", g.lds);
  std::sort(idx, idx + count, [&](int l, int r2) { return pairs[l].Ka + pairs[l].Kb > pairs[r2].Ka + pairs[r2].Kb; });
  int order = 0;
  for (int i = 0; i < count; i++) order |= idx[i] << (4 * i);
  RbPairBf16Multi m;
  for (int i = 0; i < kWinMulti; i++) m.c[i] = pairs[i < count ? i : 0];
  const dim3 grid((unsigned)ceil_div(a.L, kColsB), (unsigned)(a.N * count));
  if (a.C == 32) launch_inst<1, 1>(s, m, a.N, order, g, grid);
  else if (a.C == 64) launch_inst<2, 1>(s, m, a.N, order, g, grid);
  else launch_inst<4, 2>(s, m, a.N, order, g, grid);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "rb_pair_bf16 launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(rb_pair_bf16, (rb_pair_bf16_kernel<1, 1>)); } }
