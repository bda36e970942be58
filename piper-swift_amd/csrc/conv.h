// conv.h — internal interface of the Conv1d / ConvTranspose1d kernel family (conv.hip).
#pragma once
#include "common.h"

namespace ph {

// input transform applied while the B operand (activations) is loaded
enum Prologue : int {
  PRO_NONE = 0,
  PRO_LRELU = 1,       // x = lrelu(x, alpha)
  PRO_AVG3_LRELU = 2,  // x = lrelu(((x + x2) + x3) / 3, alpha)   (HiFi-GAN MRF mean folded into the consumer)
  PRO_LN = 3,          // x = ((x − mean_t) / sd_t)·ln_gamma[c] + ln_beta[c]: channel LayerNorm folded into the consumer (K ∈ {1,3})
};

// output transform / routing applied to the accumulator tile
enum Epilogue : int {
  EPI_STORE = 0,        // y = v (+ res)                         v = bias + conv
  EPI_RELU = 1,         // y = relu(v)
  EPI_TANH = 2,         // y = tanh(v)
  EPI_RSUB = 3,         // y = res − v                           (flow coupling reverse: x1 − m)
  EPI_WN_RES_SKIP = 4,  // co <  wn_c: y[co] = res[co] + v ; co ≥ wn_c: y2[co−wn_c] = (skip? skip[..]:0) + v
  EPI_WN_SKIP_LAST = 5, // y2[co] = (skip? skip[co]:0) + v
  EPI_CONVT = 6,        // rows are (co, phase): y[co][q·s + phase − padL] = v
  EPI_MRF_MEAN = 7,     // r2 = v + res;  y = lrelu(((mrf_a + mrf_b) + r2) / 3, alpha2)   (last ResBlock conv of a stage)
};

struct ConvArgs {
  // tensors
  const float* x = nullptr;   // [N, x_channels, Lin]
  const float* x2 = nullptr;  // PRO_AVG3_LRELU
  const float* x3 = nullptr;
  const float* w = nullptr;   // MFMA path: packed 32-wide fragments (pack_conv_weights); direct path: raw ONNX layout
  const float* w16 = nullptr; // optional packed 16-wide fragments (short-utterance geometry)
  const float* w8 = nullptr;   // optional 8-row fragment image (pack_conv_weights_rows8): many input channels, few output rows (conv_lean.hip)
  const float* w16g = nullptr; // gated convs: 16-wide fragments with 8 tanh rows + their 8 sigmoid rows per tile (pack_conv_weights_gate16; conv_short.hip)
  const float* bias = nullptr;
  const float* res = nullptr;   // residual / minuend, same addressing as y
  const float* skip = nullptr;  // EPI_WN_*: running skip sum, same addressing as y2 (may be null)
  float* y = nullptr;
  float* y2 = nullptr;
  // geometry
  int N = 1, Cin = 0, Cout = 0, K = 1, dil = 1, padL = 0, Lin = 0, Lout = 0;
  int stride = 1, groups = 1;  // direct path only
  int64_t x_batch_stride = 0;  // floats between batch items of x (x2/x3 share it)
  int64_t y_batch_stride = 0;  // floats between batch items of y / res
  int64_t y2_batch_stride = 0;
  int in_ch_base = 0, in_ch_sign = 1;    // physical input channel = in_ch_base + in_ch_sign·ci  (folds Flip/Split)
  int out_ch_base = 0, out_ch_sign = 1;  // physical output channel of y/res = out_ch_base + out_ch_sign·co
  int y_len = 0;                         // row length of y/res/y2/skip (floats)
  int prologue = PRO_NONE;
  float alpha = 0.0f;
  int epilogue = EPI_STORE;
  int gate = 0;  // MFMA path: Cout = 2·H rows; emits H rows tanh(a)·sigmoid(b)
  unsigned long long* trace = nullptr;  // tools/probe/streamprobe (-DPH_STREAM_TRACE): s_memtime stamps per wave; unused otherwise
  int wn_c = 0;  // EPI_WN_RES_SKIP split point
  // ConvTranspose (EPI_CONVT): GEMM rows = Cout_ct·ct_stride, GEMM cols = q
  int ct_stride = 0, ct_padL = 0, ct_Lout = 0;
  int ct_shift = -1;  // log2(ct_stride) when it is a power of two (set by launch_conv_mfma)
  // EPI_MRF_MEAN: the two other ResBlock outputs of the stage (addressed like y) and the slope applied after the mean
  const float* mrf_a = nullptr;
  const float* mrf_b = nullptr;
  float alpha2 = 0.0f;
  // bucketed schedules: batch item n really holds len_ptr[n]·len_mul input positions per row (≤ Lin, the row stride); taps
  // beyond read as zero padding, exactly as if the tensor ended there. null ⇒ Lin. Outputs past the true length are
  // don't-care values (never read unmasked by anything downstream).
  const int* len_ptr = nullptr;
  int len_mul = 1;
  // LayerNorm across a kernel boundary without a kernel of its own (GraphExecutor.swift:2071-2125: ReduceMean / Sub / Pow /
  // ReduceMean / Add / Sqrt / Div / Mul / Add). PRODUCER (EPI_STORE with stats_out): besides y = res + conv it writes, per
  // 16-row slot and column, Σ y and the CENTRED Σ (y − mean_slot)² of its rows → stats_out [N][ceil(Cout/16)][y_len][2] (a 32-row
  // tile writes its two slots, so the consumer never needs to know the producer's tile size; fixed slots ⇒ deterministic).
  // CONSUMER (PRO_LN): combines the slots of a column with Chan's parallel-variance formula — mean = ΣΣ / C, M2 = Σ [M2_i +
  // n_i·(mean_i − mean)²], var = M2 / C: the two-pass accuracy of the graph's chain in one memory round trip (r3; the one-pass
  // Σ y² / C − mean² of round 2 lost (mean / sigma)²·6e-8) — and normalises its B operand on load; the waves of row tile 0 also
  // write the normalised tensor to ln_out (the residual operand of the next Add, and the "enc_out" tap).
  float* stats_out = nullptr;
  const float* ln_stats = nullptr;
  const float* ln_gamma = nullptr;
  const float* ln_beta = nullptr;
  float* ln_out = nullptr;
  float ln_eps = 1e-5f;
  int ln_self = 0;  // PRO_LN without ln_stats: the consumer computes the statistics of its own operand (conv_lean.hip; conv_lean_ln_self_ok says when)
};

// number of floats of the packed fragment image for a [Cout, Cin, K] conv
size_t packed_conv_floats(int Cout, int Cin, int K, int tm = 32);
// ConvTranspose [Cin, Cout, K] stride s → rows Cout·s, taps ceil(K/s)
size_t packed_convt_floats(int Cin, int Cout, int K, int s, int tm = 32);
int pack_conv_weights(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed, int tm = 32);
int pack_convt_weights(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, float* packed, int tm = 32);
// gated conv (Cout = 2·H rows, tanh half then sigmoid half): 16-row tiles of rows {8m … 8m+7} ∪ {H + 8m … H + 8m+7}; same size as the tm = 16 image
int pack_conv_weights_gate16(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed);
// 8-row tiles: [ceil(Cout/8)][Cin/4 · K steps][32 floats = 4 channels × 8 rows]; Cin % 4 == 0
size_t packed_conv_rows8_floats(int Cout, int Cin, int K);
int pack_conv_weights_rows8(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed);
// tile geometry launch_conv_mfma will pick for this problem when 16-wide fragments are available (32 or 16)
int conv_pick_tile(piper_hip_ctx* ctx, int Cout, int Lout, int N, int gate);

// true when the MFMA implicit-GEMM path can run this geometry
bool conv_mfma_eligible(int Cout, int Cin, int K, int stride, int groups);
// Enqueue one conv on `s`. args.w must already be packed for the MFMA path.
int launch_conv_mfma(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a);
// The k = 1 convs of one utterance with a minimal instruction count: conv_lean.hip. Same return convention as try_launch_conv_short.
int try_launch_conv_lean(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a);
bool conv_lean_ln_self_ok(piper_hip_ctx* ctx, int Cin, int Cout, int K, int padL, int L, int N);
// Short rows (one utterance's encoder / flow convs): conv_short.hip. 1 = enqueued, 0 = not this kernel's case (use launch_conv_mfma's
// streaming kernel), < 0 = error. Called by launch_conv_mfma.
int try_launch_conv_short(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a);
// Thread-per-output kernel for everything else (groups, stride, tiny Cout). args.w is raw [Cout, Cin/g, K].
int launch_conv_direct(piper_hip_ctx* ctx, hipStream_t s, const ConvArgs& a);
// Thread-per-output ConvTranspose for geometries the phase decomposition does not cover. w raw [Cin, Cout/g, K].
int launch_convt_direct(piper_hip_ctx* ctx, hipStream_t s, const float* x, const float* w, const float* bias, float* y, int N,
                        int Cin, int Lin, int Cout, int K, int stride, int dil, int padL, int Lout, int groups);

// algorithmic work of one conv (SURVEY.md Appendix A recipe)
inline double conv_flops(int Cout, int Cin_per_g, int K, int64_t Lout) { return 2.0 * Cout * (double)Lout * Cin_per_g * K; }

}  // namespace ph
