// conv_win.h — internal interface of the fp32 "window" Conv1d / ConvTranspose1d kernel (conv_win.hip).
//
// Same contraction as conv_stream_kernel (exact fp32 on v_mfma_f32_32x32x2_f32) for the long-row convs of the HiFi-GAN
// generator, organised like conv_bf16_kernel: the block's input window is staged ONCE into LDS (LeakyReLU and the zero
// padding applied on the way in), weight fragments stream from L2 through a register ring, and small problems split the
// contraction over the block's waves (fixed-order reduction through LDS).
#pragma once
#include "common.h"

namespace ph {

struct ConvWinArgs {
  const float* x = nullptr;     // fp32 [N][Cin][Lin]
  const float* x2 = nullptr;    // both set: the input is ((x + x2) + x3) / 3 (HiFi-GAN MRF mean folded into the consumer)
  const float* x3 = nullptr;
  const float* w4 = nullptr;    // fragment image (pack_conv_weights_win / pack_convt_weights_win)
  const float* bias = nullptr;  // [Cout] or null
  const float* res = nullptr;   // [N][Cout][y_len] added to the result, may be null
  const float* mrf_a = nullptr; // both set: result = ((mrf_a + mrf_b) + result) / 3
  const float* mrf_b = nullptr;
  float* y = nullptr;           // [N][Cout][y_len]
  float pro_alpha = 1.0f;       // LeakyReLU slope applied to x while staging (1 ⇒ none)
  float out_alpha = 1.0f;       // LeakyReLU slope applied to the final value (1 ⇒ none)
  int N = 1, Cin = 0, Cout = 0, K = 1, dil = 1, padL = 0;
  int Lin = 0;                  // input row length (multiple of 4)
  int Lout = 0;                 // GEMM columns (conv: output length; convT: input length)
  int y_len = 0;                // row length of y / res / mrf_*
  int ct_stride = 0, ct_pad = 0;  // ConvTranspose1d: rows (phase ρ, co), K/s taps, output position s·q + ρ
  // bucketed schedules: true input length of batch item n = len_ptr[n]·len_mul (≤ Lin); positions beyond it read as zero
  // padding, exactly as if the tensor ended there. null ⇒ Lin.
  const int* len_ptr = nullptr;
  int len_mul = 1;
};

size_t packed_conv_win_floats(int Cout, int Cin, int K);
size_t packed_convt_win_floats(int Cin, int Cout, int K, int stride);
int pack_conv_weights_win(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed);
int pack_convt_weights_win(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, int pad, float* packed);

// geometry covered: stride 1, groups 1, Cin even, Lin % 4 == 0, window within LDS
bool conv_win_eligible(int Cout, int Cin, int K, int dil, int padL, int Lin, int Lout);
bool convt_win_eligible(int Cin, int Cout, int K, int stride, int pad, int Lin);

int launch_conv_win(piper_hip_ctx* ctx, hipStream_t s, const ConvWinArgs& a);
// Up to kWinMulti independent convs of the SAME shape class (N, Cin, Cout, Lin, Lout, conv vs convT) in one launch — the
// three ResBlocks of a HiFi-GAN stage differ only in kernel size, dilation, weights and buffers. One launch instead of
// three short dependent-free ones: the blocks of all convs share the chip, and two launch boundaries disappear.
constexpr int kWinMulti = 3;
int launch_conv_win_multi(piper_hip_ctx* ctx, hipStream_t s, const ConvWinArgs* convs, int count);

// ---- conv_pipe.hip: the same contraction as a persistent, chunk-pipelined kernel (Cin % 32 == 0); its own fragment order
size_t packed_conv_pipe_floats(int Cout, int Cin, int K);
size_t packed_convt_pipe_floats(int Cin, int Cout, int K, int stride);
int pack_conv_weights_pipe(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed);
int pack_convt_weights_pipe(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, int pad, float* packed);
bool conv_pipe_eligible(int Cout, int Cin, int K, int dil, int padL, int Lin, int Lout);
bool convt_pipe_eligible(int Cin, int Cout, int K, int stride, int pad, int Lin);
// convs[i].w4 must be the pipe image; same multi-conv contract as launch_conv_win_multi
int launch_conv_pipe_multi(piper_hip_ctx* ctx, hipStream_t s, const ConvWinArgs* convs, int count);

// ---- rb_pair.hip: two chained ResBlock convs in one kernel, the intermediate kept in LDS (C ∈ {32, 64}, odd kernels)
//   x1 = [res_a ? x : 0] + conv_a(lrelu(x)) + ba            (dilation dila, 'same' padding)
//   y  = (res_b_x ? x : x1) + conv_b(lrelu(x1)) + bb        (dilation dilb)
// ResBlock2 step pair: res_a = 1, res_b_x = 0. ResBlock1 pair (convs1[i], convs2[i]): res_a = 0, res_b_x = 1.
struct RbPairArgs {
  const float* x = nullptr;    // [N][C][L]
  float* y = nullptr;          // [N][C][L]; must not alias x
  const float *wa4 = nullptr, *ba = nullptr, *wb4 = nullptr, *bb = nullptr;  // conv_win fragment images + biases
  int Ka = 1, dila = 1, Kb = 1, dilb = 1;
  int res_a = 1, res_b_x = 0;
  float alpha = 0.1f;          // LeakyReLU slope, 0 < α < 1
  int N = 1, C = 0, L = 0;     // L % 4 == 0
  const int* len_ptr = nullptr;  // true length of item n = len_ptr[n]·len_mul (bucketed schedules); null ⇒ L
  int len_mul = 1;
};
bool rb_pair_eligible(int C, int Ka, int dila, int Kb, int dilb, int L);
int launch_rb_pair_multi(piper_hip_ctx* ctx, hipStream_t s, const RbPairArgs* pairs, int count);  // ≤ kWinMulti pairs, same N, C, L

}  // namespace ph
