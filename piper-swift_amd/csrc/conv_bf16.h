// conv_bf16.h — internal interface of the bf16-operand Conv1d / ConvTranspose1d kernels (conv_bf16.hip).
//
// "bf16 activations/weights with fp32 accumulate" (SURVEY.md §8d config 5). Operands of the contraction are bf16, everything
// that is ADDED to its result (bias, residual stream, MRF mean) stays fp32:
//   activations  "C8" image  bf16 [N][C/8][Lp][8]   Lp = kC8Halo + round_up(L,128) + kC8Halo, zero outside [0,L)
//                            — one 16-byte load is the 8 consecutive-k elements a lane feeds to v_mfma_f32_32x32x16_bf16,
//                              and the stored zero halo IS the conv's zero padding (no bounds tests on the load path)
//   weights      fragment image bf16 [row tile][tap][Cin/16][64 lanes][8]
//   residual stream / outputs  fp32 [N][C][L] (the layout every other kernel of the library uses)
#pragma once
#include "common.h"

namespace ph {

// the weight ring of conv_bf16_kernel prefetches up to 16 steps (16 × 64 lanes × 8 elements) past the last fragment of a
// row tile: every packed image carries that many readable elements behind it (zero-filled by the pack kernels)
constexpr int kBf16WeightPad = 16 * 64 * 8 * 2;
constexpr int kC8Halo = 64;  // ≥ every "same" padding of a Piper generator (max (7·12−12)/2 = 36 medium, (11·5−5)/2 = 25 high)

inline int64_t c8_round_len(int64_t L) { return (L + 127) / 128 * 128; }
inline int64_t c8_row_len(int64_t L) { return kC8Halo + c8_round_len(L) + kC8Halo; }                 // positions per channel block
inline int64_t c8_elems(int64_t N, int64_t C, int64_t L) { return N * ((C + 7) / 8) * c8_row_len(L) * 8; }  // bf16 elements

struct ConvBf16Args {
  const uint16_t* x = nullptr;  // C8 image of the input [N][Cin/8][x_row][8]
  const uint16_t* w = nullptr;  // packed fragments (pack_conv_weights_bf16 / pack_convt_weights_bf16)
  const float* bias = nullptr;  // [Cout] or null
  const float* res = nullptr;   // fp32 [N][Cout][y_len] added to the result (residual stream), may be null
  const float* mrf_a = nullptr; // both set: result = ((mrf_a + mrf_b) + result) / 3  (HiFi-GAN MRF mean)
  const float* mrf_b = nullptr;
  float* y = nullptr;           // fp32 [N][Cout][y_len] result, may be null
  uint16_t* act = nullptr;      // C8 image of lrelu(result, act_alpha) [N][Cout/8][act_row][8], may be null
  float act_alpha = 1.0f;       // 1 ⇒ identity
  int N = 1, Cin = 0, Cout = 0, K = 1, dil = 1, padL = 0;
  int Lout = 0;                 // GEMM columns (conv: output length; convT: input length)
  int x_row = 0, act_row = 0;   // positions per channel block of x / act
  int y_len = 0;                // row length of y / res / mrf_*
  // ConvTranspose1d, stride s | K, K − s = 2·pad: GEMM rows = (phase ρ, co), tap j reads q + ⌊(ρ+pad)/s⌋ − j and the
  // result lands at output position s·q + ρ. Cout is the real channel count; the row count is s·Cout.
  int ct_stride = 0, ct_pad = 0;
  // bucketed schedules: batch item n really has len_ptr[n]·len_mul OUTPUT positions (≤ the bucket's). The activation image
  // written for the next conv gets zeros beyond them, so that the image's "zero outside [0, L)" invariant — which IS the next
  // conv's zero padding — holds for the true length. null ⇒ everything up to y_len is real.
  const int* len_ptr = nullptr;
  int len_mul = 1;
};

// bf16 elements of the packed fragment images (including kBf16WeightPad)
size_t packed_conv_bf16_elems(int Cout, int Cin, int K);
size_t packed_convt_bf16_elems(int Cin, int Cout, int K, int stride);
// w fp32 [Cout][Cin][K] → fragment image (round to nearest even)
int pack_conv_weights_bf16(hipStream_t s, const float* w, int Cout, int Cin, int K, uint16_t* packed);
// w fp32 [Cin][Cout][K] → fragment image with rows (phase, co), taps K/stride
int pack_convt_weights_bf16(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, int pad, uint16_t* packed);
// fp32 [N][C][L] → C8 image of lrelu(x, alpha) (interior only: the image must have been zeroed once).
// row = positions per channel block of the image (0 ⇒ c8_row_len(L))
int pack_act_c8(hipStream_t s, const float* x, int N, int C, int L, float alpha, uint16_t* act, int64_t row = 0,
                const int* len_ptr = nullptr);  // positions ≥ len_ptr[n] are written as zeros

// C8 image of lrelu(((a + b) + c) / 3, alpha) — the MRF mean of three fp32 [N][C][L] tensors (true length of item n = len_ptr[n]·len_mul)
int pack_mean3_c8(hipStream_t s, const float* a, const float* b, const float* c, int N, int C, int L, float alpha, uint16_t* act, int64_t row,
                  const int* len_ptr, int len_mul);

// geometry the bf16 kernels cover (stride-1, ungrouped, Cin % 32 == 0, padding within the halo)
bool conv_bf16_eligible(int Cout, int Cin, int K, int dil, int padL, int padR);
bool convt_bf16_eligible(int Cin, int Cout, int K, int stride, int padL, int padR, int dil, int out_pad);

int launch_conv_bf16(piper_hip_ctx* ctx, hipStream_t s, const ConvBf16Args& a);
// up to kBf16Multi independent convs of the same shape class (N, Cin, Cout, Lout, conv vs convT) in one launch
constexpr int kBf16Multi = 3;
int launch_conv_bf16_multi(piper_hip_ctx* ctx, hipStream_t s, const ConvBf16Args* convs, int count);

// ---- rb_pair_bf16.hip: a ResBlock1 pair  y = x + conv_b(lrelu(conv_a(lrelu(x)) + ba)) + bb  in one kernel (C ∈ {32, 64, 128},
// odd kernels, conv b reaching ≤ 16 positions); x / y fp32 [N][C][L], weights = the fragment images above
struct RbPairBf16Args {
  const float* x = nullptr;
  float* y = nullptr;            // must not alias x; may be null when only `act` is wanted
  const float* mrf_a = nullptr;  // both set: the result becomes ((mrf_a + mrf_b) + y) / 3 — the MRF mean over the stage's ResBlocks
  const float* mrf_b = nullptr;
  uint16_t* act = nullptr;       // optional C8 image of lrelu(result, alpha) [N][C/8][act_row][8] for a consumer that is not fused
  int act_row = 0;
  const uint16_t *wa = nullptr, *wb = nullptr;
  const float *ba = nullptr, *bb = nullptr;
  int Ka = 1, dila = 1, Kb = 1, dilb = 1;
  float alpha = 0.1f;
  int N = 1, C = 0, L = 0;       // L % 4 == 0
  const int* len_ptr = nullptr;  // true length of item n = len_ptr[n]·len_mul; null ⇒ L
  int len_mul = 1;
};
bool rb_pair_bf16_eligible(int C, int Ka, int dila, int Kb, int dilb, int L);
int launch_rb_pair_bf16_multi(piper_hip_ctx* ctx, hipStream_t s, const RbPairBf16Args* pairs, int count);  // ≤ 3 pairs, same N, C, L

}  // namespace ph
