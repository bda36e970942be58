// ops_conv.cpp — C-ABI entry points for Conv / ConvTranspose and the fused WaveNet-layer / HiFi-GAN-ResBlock ops.
// Shape contracts and error wording follow MetalBackend.conv1dF32 / convTranspose1dF32
// (MetalBackend.swift:1149-1228, 2812-2895).
#include "conv.h"
#include "conv_bf16.h"
#include "conv_win.h"

using namespace ph;

namespace {
int pool_floats(piper_hip_ctx* ctx, size_t n, float** out) {
  void* p = nullptr;
  int rc = ctx->pool.alloc(n * sizeof(float), &p);
  if (rc) return rc;
  *out = (float*)p;
  return PIPER_HIP_OK;
}
bool fits_i32(int64_t v) { return v >= 0 && v <= 0x7fffffff; }
}  // namespace

PH_EXPORT int piper_hip_conv1d_f32(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], const float* w,
                                   const int64_t w_shape[3], const float* bias, const piper_hip_conv1d_params* p,
                                   float** out, int64_t out_shape[3], piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (!x_shape || !w_shape || !p) PH_FAIL(PIPER_HIP_ERR_ARG, "conv1dF32: null shape/params");
  const int64_t N = x_shape[0], Cin = x_shape[1], Lin = x_shape[2];
  const int64_t Cout = w_shape[0], K = w_shape[2];
  const int64_t g = p->groups < 1 ? 1 : p->groups;
  if (N < 0 || Cin < 0 || Lin < 0 || Cout < 0 || K <= 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dF32: negative/zero dimension");
  if (p->stride < 1 || p->dilation < 1 || p->pad_l < 0 || p->pad_r < 0)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dF32: stride/dilation must be >=1 and pads >=0");
  if (Cin % g) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dF32 invalid groups: C_in=%lld groups=%lld", (long long)Cin, (long long)g);
  if (Cout % g) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dF32 invalid groups: C_out=%lld groups=%lld", (long long)Cout, (long long)g);
  if (w_shape[1] != Cin / g)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dF32 weight C_in mismatch: weight[1]=%lld input C=%lld groups=%lld", (long long)w_shape[1],
            (long long)Cin, (long long)g);
  const int64_t num = Lin + p->pad_l + p->pad_r - (int64_t)p->dilation * (K - 1) - 1;
  // Swift Int division truncates toward zero (MetalBackend.swift:1177)
  const int64_t Lout = num / p->stride + 1;
  if (Lout < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dF32 produced invalid L_out=%lld", (long long)Lout);
  if (out_shape) { out_shape[0] = N; out_shape[1] = Cout; out_shape[2] = Lout; }
  const int64_t count = N * Cout * Lout;
  if (!fits_i32(Cin * Lin) || !fits_i32(Cout * Lout) || !fits_i32(Cin / g * K * Cout))
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dF32: tensor too large for 32-bit indexing");
  int rc = ensure_out(ctx, out, (size_t)count, 0);
  if (rc) return rc;
  if (count == 0) return PIPER_HIP_OK;
  if (!x || !w) PH_FAIL(PIPER_HIP_ERR_ARG, "conv1dF32: null input");
  StreamScope ss(ctx, stream);
  ConvArgs a;
  a.x = x; a.bias = bias; a.y = *out;
  a.N = (int)N; a.Cin = (int)Cin; a.Cout = (int)Cout; a.K = (int)K; a.dil = p->dilation; a.padL = p->pad_l;
  a.Lin = (int)Lin; a.Lout = (int)Lout; a.stride = p->stride; a.groups = (int)g;
  a.x_batch_stride = Cin * Lin; a.y_batch_stride = Cout * Lout; a.y_len = (int)Lout;
  static const bool no_win = getenv("PIPER_HIP_NO_WIN") != nullptr;
  if (!no_win && g == 1 && p->stride == 1 && Lout >= 1024 && Cout >= 16 && ((uintptr_t)x & 15) == 0 &&
      conv_win_eligible((int)Cout, (int)Cin, (int)K, p->dilation, p->pad_l, (int)Lin, (int)Lout)) {
    // long rows: the persistent chunk-pipelined kernel when the channel count allows (Cin % 32 == 0), else the one-shot
    // window kernel (input window staged once in LDS)
    static const bool no_pipe = getenv("PIPER_HIP_NO_PIPE") != nullptr;
    const bool pipe = !no_pipe && conv_pipe_eligible((int)Cout, (int)Cin, (int)K, p->dilation, p->pad_l, (int)Lin, (int)Lout);
    float* packed = nullptr;
    rc = pool_floats(ctx, pipe ? packed_conv_pipe_floats((int)Cout, (int)Cin, (int)K) : packed_conv_win_floats((int)Cout, (int)Cin, (int)K), &packed);
    if (rc) return rc;
    defer_free(ctx, packed);
    if (pipe) pack_conv_weights_pipe(ss.s, w, (int)Cout, (int)Cin, (int)K, packed);
    else pack_conv_weights_win(ss.s, w, (int)Cout, (int)Cin, (int)K, packed);
    ConvWinArgs wa;
    wa.x = x; wa.w4 = packed; wa.bias = bias; wa.y = *out;
    wa.N = (int)N; wa.Cin = (int)Cin; wa.Cout = (int)Cout; wa.K = (int)K; wa.dil = p->dilation; wa.padL = p->pad_l;
    wa.Lin = (int)Lin; wa.Lout = (int)Lout; wa.y_len = (int)Lout;
    rc = pipe ? launch_conv_pipe_multi(ctx, ss.s, &wa, 1) : launch_conv_win(ctx, ss.s, wa);
  } else if (Lin >= 1 && conv_mfma_eligible((int)Cout, (int)Cin, (int)K, p->stride, (int)g)) {
    const int tm = conv_pick_tile(ctx, (int)Cout, (int)Lout, (int)N, 0);
    float* packed = nullptr;
    rc = pool_floats(ctx, packed_conv_floats((int)Cout, (int)Cin, (int)K, tm), &packed);
    if (rc) return rc;
    defer_free(ctx, packed);
    pack_conv_weights(ss.s, w, (int)Cout, (int)Cin, (int)K, packed, tm);
    if (tm == 16) a.w16 = packed; else a.w = packed;
    rc = launch_conv_mfma(ctx, ss.s, a);
  } else {
    a.w = w;
    rc = launch_conv_direct(ctx, ss.s, a);
  }
  if (rc) return rc;
  return ss.finish("conv1d_f32");
}

PH_EXPORT int piper_hip_convtranspose1d_f32(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], const float* w,
                                            const int64_t w_shape[3], const float* bias,
                                            const piper_hip_convtranspose1d_params* p, float** out, int64_t out_shape[3],
                                            piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (!x_shape || !w_shape || !p) PH_FAIL(PIPER_HIP_ERR_ARG, "convTranspose1dF32: null shape/params");
  const int64_t N = x_shape[0], Cin = x_shape[1], Lin = x_shape[2];
  const int64_t g = p->groups < 1 ? 1 : p->groups;
  if (N < 0 || Cin < 0 || Lin < 0 || w_shape[1] < 0 || w_shape[2] <= 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dF32: bad dimension");
  if (p->stride < 1 || p->dilation < 1 || p->pad_l < 0 || p->pad_r < 0 || p->output_padding < 0)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dF32: stride/dilation must be >=1 and pads >=0");
  if (Cin % g) PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dF32 invalid groups: C_in=%lld groups=%lld", (long long)Cin, (long long)g);
  if (w_shape[0] != Cin) PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dF32 weight[0] mismatch: %lld vs C_in=%lld", (long long)w_shape[0], (long long)Cin);
  const int64_t cog = w_shape[1], Cout = cog * g, K = w_shape[2];
  const int64_t Lout = (Lin - 1) * p->stride - p->pad_l - p->pad_r + (int64_t)p->dilation * (K - 1) + p->output_padding + 1;
  if (Lout <= 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dF32 produced invalid L_out=%lld", (long long)Lout);
  if (out_shape) { out_shape[0] = N; out_shape[1] = Cout; out_shape[2] = Lout; }
  const int64_t count = N * Cout * Lout;
  if (!fits_i32(Cin * Lin) || !fits_i32(Cout * Lout) || !fits_i32(Cin * cog * K))
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dF32: tensor too large for 32-bit indexing");
  int rc = ensure_out(ctx, out, (size_t)count, 0);
  if (rc) return rc;
  if (count == 0) return PIPER_HIP_OK;
  if (!x || !w) PH_FAIL(PIPER_HIP_ERR_ARG, "convTranspose1dF32: null input");
  StreamScope ss(ctx, stream);
  const int s = p->stride;
  static const bool no_win = getenv("PIPER_HIP_NO_WIN") != nullptr;
  if (!no_win && g == 1 && p->dilation == 1 && p->output_padding == 0 && p->pad_l == p->pad_r && Lin >= 256 &&
      ((uintptr_t)x & 15) == 0 && convt_win_eligible((int)Cin, (int)Cout, (int)K, s, p->pad_l, (int)Lin)) {
    // the chunk-pipelined kernel only on request (PIPER_HIP_PIPE_CT_MIN_GFLOP): the window kernel is faster for ConvTranspose (voice.hip)
    static const bool no_pipe = getenv("PIPER_HIP_NO_PIPE") != nullptr;
    static const double pipe_ct_min = [] { const char* e = getenv("PIPER_HIP_PIPE_CT_MIN_GFLOP"); return e ? atof(e) * 1e9 : 1e30; }();
    const bool pipe = !no_pipe && 2.0 * (double)Cout * (double)Cin * (double)K * (double)Lin * (double)N >= pipe_ct_min &&
                      convt_pipe_eligible((int)Cin, (int)Cout, (int)K, s, p->pad_l, (int)Lin);
    float* packed = nullptr;
    rc = pool_floats(ctx, pipe ? packed_convt_pipe_floats((int)Cin, (int)Cout, (int)K, s) : packed_convt_win_floats((int)Cin, (int)Cout, (int)K, s), &packed);
    if (rc) return rc;
    defer_free(ctx, packed);
    if (pipe) pack_convt_weights_pipe(ss.s, w, (int)Cin, (int)Cout, (int)K, s, p->pad_l, packed);
    else pack_convt_weights_win(ss.s, w, (int)Cin, (int)Cout, (int)K, s, p->pad_l, packed);
    ConvWinArgs wa;
    wa.x = x; wa.w4 = packed; wa.bias = bias; wa.y = *out;
    wa.N = (int)N; wa.Cin = (int)Cin; wa.Cout = (int)Cout; wa.K = (int)K; wa.Lin = (int)Lin; wa.Lout = (int)Lin; wa.y_len = (int)Lout;
    wa.ct_stride = s; wa.ct_pad = p->pad_l;
    rc = pipe ? launch_conv_pipe_multi(ctx, ss.s, &wa, 1) : launch_conv_win(ctx, ss.s, wa);
  } else if (g == 1 && p->dilation == 1 && Cin >= 2 && Cout * s >= 8 && Lin >= 1) {
    // phase decomposition: output phase (x+padL) mod s is a dense conv with ceil(K/s) taps (no '%' test per tap)
    const int J = (int)((K + s - 1) / s);
    const int Lg = (int)((Lout - 1 + p->pad_l) / s + 1);
    const int tm = conv_pick_tile(ctx, (int)(Cout * s), Lg, (int)N, 0);
    float* packed = nullptr;
    rc = pool_floats(ctx, packed_convt_floats((int)Cin, (int)Cout, (int)K, s, tm), &packed);
    if (rc) return rc;
    defer_free(ctx, packed);
    pack_convt_weights(ss.s, w, (int)Cin, (int)Cout, (int)K, s, packed, tm);
    ConvArgs a;
    a.x = x; a.bias = bias; a.y = *out;
    if (tm == 16) a.w16 = packed; else a.w = packed;
    a.N = (int)N; a.Cin = (int)Cin; a.Cout = (int)(Cout * s); a.K = J; a.dil = -1; a.padL = 0;
    a.Lin = (int)Lin; a.Lout = (int)((Lout - 1 + p->pad_l) / s + 1);
    a.x_batch_stride = Cin * Lin; a.y_batch_stride = Cout * Lout; a.y_len = (int)Lout;
    a.epilogue = EPI_CONVT; a.ct_stride = s; a.ct_padL = p->pad_l; a.ct_Lout = (int)Lout;
    rc = launch_conv_mfma(ctx, ss.s, a);
  } else {
    rc = launch_convt_direct(ctx, ss.s, x, w, bias, *out, (int)N, (int)Cin, (int)Lin, (int)Cout, (int)K, s, p->dilation,
                             p->pad_l, (int)Lout, (int)g);
  }
  if (rc) return rc;
  return ss.finish("convtranspose1d_f32");
}

// bf16-operand variants (SURVEY.md §8b "bf16 variants", §8d config 5): same contract and shapes as the f32 entry points,
// fp32 tensors in and out; x and w are rounded to bf16 (nearest even) on the device, products accumulate in fp32.
// Geometry outside the bf16 kernels' coverage (stride ≠ 1, groups ≠ 1, Cin % 32 ≠ 0, padding > 64) → UNSUPPORTED.
PH_EXPORT int piper_hip_conv1d_bf16(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], const float* w,
                                    const int64_t w_shape[3], const float* bias, const piper_hip_conv1d_params* p,
                                    float** out, int64_t out_shape[3], piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (!x_shape || !w_shape || !p) PH_FAIL(PIPER_HIP_ERR_ARG, "conv1dBF16: null shape/params");
  const int64_t N = x_shape[0], Cin = x_shape[1], Lin = x_shape[2];
  const int64_t Cout = w_shape[0], K = w_shape[2];
  if (N < 0 || Cin < 0 || Lin < 0 || Cout < 0 || K <= 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dBF16: negative/zero dimension");
  if (p->stride < 1 || p->dilation < 1 || p->pad_l < 0 || p->pad_r < 0)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dBF16: stride/dilation must be >=1 and pads >=0");
  if (w_shape[1] != Cin) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dBF16 weight C_in mismatch: weight[1]=%lld input C=%lld", (long long)w_shape[1], (long long)Cin);
  if (p->stride != 1 || (p->groups > 1) || !conv_bf16_eligible((int)Cout, (int)Cin, (int)K, p->dilation, p->pad_l, p->pad_r))
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv1dBF16: geometry not covered (stride 1, groups 1, C_in %% 32 == 0, padding <= 64)");
  const int64_t Lout = Lin + p->pad_l + p->pad_r - (int64_t)p->dilation * (K - 1);
  if (Lout < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dBF16 produced invalid L_out=%lld", (long long)Lout);
  if (out_shape) { out_shape[0] = N; out_shape[1] = Cout; out_shape[2] = Lout; }
  const int64_t count = N * Cout * Lout;
  if (!fits_i32(Cin * Lin) || !fits_i32(Cout * Lout) || !fits_i32(Cin * K * Cout))
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv1dBF16: tensor too large for 32-bit indexing");
  int rc = ensure_out(ctx, out, (size_t)count, 0);
  if (rc) return rc;
  if (count == 0) return PIPER_HIP_OK;
  if (!x || !w) PH_FAIL(PIPER_HIP_ERR_ARG, "conv1dBF16: null input");
  StreamScope ss(ctx, stream);
  const int64_t Lmax = Lin > Lout ? Lin : Lout;
  const int64_t row = c8_row_len(Lmax);
  const size_t img_bytes = (size_t)N * (Cin / 8) * row * 16;
  void *img = nullptr, *pw = nullptr;
  rc = ctx->pool.alloc(img_bytes, &img);
  if (rc) return rc;
  defer_free(ctx, img);
  rc = ctx->pool.alloc(packed_conv_bf16_elems((int)Cout, (int)Cin, (int)K) * 2, &pw);
  if (rc) return rc;
  defer_free(ctx, pw);
  PH_HIP(hipMemsetAsync(img, 0, img_bytes, ss.s), PIPER_HIP_ERR_LAUNCH);
  pack_act_c8(ss.s, x, (int)N, (int)Cin, (int)Lin, 1.0f, (uint16_t*)img, row);
  pack_conv_weights_bf16(ss.s, w, (int)Cout, (int)Cin, (int)K, (uint16_t*)pw);
  ConvBf16Args a;
  a.x = (const uint16_t*)img; a.w = (const uint16_t*)pw; a.bias = bias; a.y = *out;
  a.N = (int)N; a.Cin = (int)Cin; a.Cout = (int)Cout; a.K = (int)K; a.dil = p->dilation; a.padL = p->pad_l;
  a.Lout = (int)Lout; a.x_row = (int)row; a.y_len = (int)Lout;
  rc = launch_conv_bf16(ctx, ss.s, a);
  if (rc) return rc;
  return ss.finish("conv1d_bf16");
}

PH_EXPORT int piper_hip_convtranspose1d_bf16(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], const float* w,
                                             const int64_t w_shape[3], const float* bias,
                                             const piper_hip_convtranspose1d_params* p, float** out, int64_t out_shape[3],
                                             piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (!x_shape || !w_shape || !p) PH_FAIL(PIPER_HIP_ERR_ARG, "convTranspose1dBF16: null shape/params");
  const int64_t N = x_shape[0], Cin = x_shape[1], Lin = x_shape[2];
  if (N < 0 || Cin < 0 || Lin < 0 || w_shape[1] < 0 || w_shape[2] <= 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dBF16: bad dimension");
  if (p->stride < 1 || p->dilation < 1 || p->pad_l < 0 || p->pad_r < 0 || p->output_padding < 0)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dBF16: stride/dilation must be >=1 and pads >=0");
  if (w_shape[0] != Cin) PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dBF16 weight[0] mismatch: %lld vs C_in=%lld", (long long)w_shape[0], (long long)Cin);
  const int64_t Cout = w_shape[1], K = w_shape[2];
  if (p->groups > 1 || !convt_bf16_eligible((int)Cin, (int)Cout, (int)K, p->stride, p->pad_l, p->pad_r, p->dilation, p->output_padding))
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "convTranspose1dBF16: geometry not covered (groups 1, dilation 1, C_in,C_out %% 32 == 0, K %% stride == 0, pads = (K-stride)/2)");
  const int64_t Lout = Lin * p->stride;  // (L−1)s − 2·pad + K with K − s = 2·pad
  if (out_shape) { out_shape[0] = N; out_shape[1] = Cout; out_shape[2] = Lout; }
  const int64_t count = N * Cout * Lout;
  if (!fits_i32(Cin * Lin) || !fits_i32(Cout * Lout) || !fits_i32(Cin * Cout * K))
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "convTranspose1dBF16: tensor too large for 32-bit indexing");
  int rc = ensure_out(ctx, out, (size_t)count, 0);
  if (rc) return rc;
  if (count == 0) return PIPER_HIP_OK;
  if (!x || !w) PH_FAIL(PIPER_HIP_ERR_ARG, "convTranspose1dBF16: null input");
  StreamScope ss(ctx, stream);
  const int64_t row = c8_row_len(Lin);
  const size_t img_bytes = (size_t)N * (Cin / 8) * row * 16;
  void *img = nullptr, *pw = nullptr;
  rc = ctx->pool.alloc(img_bytes, &img);
  if (rc) return rc;
  defer_free(ctx, img);
  rc = ctx->pool.alloc(packed_convt_bf16_elems((int)Cin, (int)Cout, (int)K, p->stride) * 2, &pw);
  if (rc) return rc;
  defer_free(ctx, pw);
  PH_HIP(hipMemsetAsync(img, 0, img_bytes, ss.s), PIPER_HIP_ERR_LAUNCH);
  pack_act_c8(ss.s, x, (int)N, (int)Cin, (int)Lin, 1.0f, (uint16_t*)img, row);
  pack_convt_weights_bf16(ss.s, w, (int)Cin, (int)Cout, (int)K, p->stride, p->pad_l, (uint16_t*)pw);
  ConvBf16Args a;
  a.x = (const uint16_t*)img; a.w = (const uint16_t*)pw; a.bias = bias; a.y = *out;
  a.N = (int)N; a.Cin = (int)Cin; a.Cout = (int)Cout; a.K = (int)K; a.Lout = (int)Lin; a.x_row = (int)row; a.y_len = (int)Lout;
  a.ct_stride = p->stride; a.ct_pad = p->pad_l;
  rc = launch_conv_bf16(ctx, ss.s, a);
  if (rc) return rc;
  return ss.finish("convtranspose1d_bf16");
}

PH_EXPORT int piper_hip_wavenet_layer_f32(piper_hip_ctx* ctx, const float* x, const float* skip_in, const float* w_in,
                                          const float* b_in, const float* w_rs, const float* b_rs, int64_t n, int64_t c,
                                          int64_t t, int64_t k, int64_t dilation, int last, float** x_out, float** skip_out,
                                          piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (n < 0 || c <= 0 || t < 0 || k <= 0 || dilation < 1) PH_FAIL(PIPER_HIP_ERR_SHAPE, "wavenet_layer: bad shape");
  if (c % 32) PH_FAIL(PIPER_HIP_ERR_SHAPE, "wavenet_layer: channels must be a multiple of 32 (got %lld)", (long long)c);
  if (!(k & 1)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "wavenet_layer: kernel size must be odd");
  if (!fits_i32(2 * c * t) || !fits_i32(2 * c * c * k)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "wavenet_layer: tensor too large");
  const size_t cnt = (size_t)(n * c * t);
  int rc;
  if (!last) {
    rc = ensure_out(ctx, x_out, cnt, 0);
    if (rc) return rc;
  }
  rc = ensure_out(ctx, skip_out, cnt, 0);
  if (rc) return rc;
  if (cnt == 0) return PIPER_HIP_OK;
  if (!x || !w_in || !w_rs) PH_FAIL(PIPER_HIP_ERR_ARG, "wavenet_layer: null input");
  StreamScope ss(ctx, stream);
  const int C = (int)c, T = (int)t, K = (int)k;
  const int Crs = last ? C : 2 * C;
  float *p_in = nullptr, *p_rs = nullptr, *acts = nullptr;
  const int tm_in = conv_pick_tile(ctx, 2 * C, T, (int)n, 1), tm_rs = conv_pick_tile(ctx, Crs, T, (int)n, 0);
  if ((rc = pool_floats(ctx, packed_conv_floats(2 * C, C, K, tm_in), &p_in))) return rc;
  defer_free(ctx, p_in);
  if ((rc = pool_floats(ctx, packed_conv_floats(Crs, C, 1, tm_rs), &p_rs))) return rc;
  defer_free(ctx, p_rs);
  if ((rc = pool_floats(ctx, cnt, &acts))) return rc;
  defer_free(ctx, acts);
  pack_conv_weights(ss.s, w_in, 2 * C, C, K, p_in, tm_in);
  pack_conv_weights(ss.s, w_rs, Crs, C, 1, p_rs, tm_rs);
  ConvArgs a;  // in_conv + tanh·sigmoid gate
  a.x = x; a.bias = b_in; a.y = acts;
  if (tm_in == 16) a.w16 = p_in; else a.w = p_in;
  if (tm_in == 16 && dilation == 1) {  // short rows: the gate-interleaved image of conv_short.hip (8 tanh rows + their 8 sigmoid rows per tile)
    float* p_g = nullptr;
    if ((rc = pool_floats(ctx, packed_conv_floats(2 * C, C, K, 16), &p_g))) return rc;
    defer_free(ctx, p_g);
    pack_conv_weights_gate16(ss.s, w_in, 2 * C, C, K, p_g);
    a.w16g = p_g;
  }
  a.N = (int)n; a.Cin = C; a.Cout = 2 * C; a.K = K; a.dil = (int)dilation; a.padL = (int)((k * dilation - dilation) / 2);
  a.Lin = T; a.Lout = T; a.x_batch_stride = (int64_t)C * T; a.y_batch_stride = (int64_t)C * T; a.y_len = T; a.gate = 1;
  if ((rc = launch_conv_mfma(ctx, ss.s, a))) return rc;
  ConvArgs b;  // res/skip 1×1 conv routed into x and skip
  b.x = acts; b.bias = b_rs;
  if (tm_rs == 16) b.w16 = p_rs; else b.w = p_rs;
  b.N = (int)n; b.Cin = C; b.Cout = Crs; b.K = 1; b.Lin = T; b.Lout = T;
  b.x_batch_stride = (int64_t)C * T; b.y_batch_stride = (int64_t)C * T; b.y2_batch_stride = (int64_t)C * T; b.y_len = T;
  b.skip = skip_in; b.y2 = *skip_out;
  if (last) b.epilogue = EPI_WN_SKIP_LAST;
  else { b.epilogue = EPI_WN_RES_SKIP; b.wn_c = C; b.res = x; b.y = *x_out; }
  if ((rc = launch_conv_mfma(ctx, ss.s, b))) return rc;
  return ss.finish("wavenet_layer_f32");
}

PH_EXPORT int piper_hip_hifigan_resblock_f32(piper_hip_ctx* ctx, int type, const float* x, int64_t n, int64_t c, int64_t t,
                                             int64_t k, const int32_t* dilations, int n_dil, const float* const* weights,
                                             const float* const* biases, float lrelu_slope, float** out,
                                             piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (type != 1 && type != 2) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "hifigan_resblock: type %d", type);
  if (n < 0 || c <= 0 || t < 0 || k <= 0 || n_dil < 1 || n_dil > 8 || !dilations || !weights || !biases)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "hifigan_resblock: bad arguments");
  if (!(k & 1)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "hifigan_resblock: kernel size must be odd");
  if (!fits_i32(c * t) || !fits_i32(c * c * k)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "hifigan_resblock: tensor too large");
  const size_t cnt = (size_t)(n * c * t);
  int rc = ensure_out(ctx, out, cnt, 0);
  if (rc) return rc;
  if (cnt == 0) return PIPER_HIP_OK;
  if (!x) PH_FAIL(PIPER_HIP_ERR_ARG, "hifigan_resblock: null input");
  StreamScope ss(ctx, stream);
  const int C = (int)c, T = (int)t, K = (int)k;
  const bool mfma = conv_mfma_eligible(C, C, K, 1, 1);
  // ping-pong so the residual source is never the buffer being written
  float* tmp[2] = {nullptr, nullptr};
  float* mid = nullptr;
  for (int i = 0; i < 2; i++) {
    if ((rc = pool_floats(ctx, cnt, &tmp[i]))) return rc;
    defer_free(ctx, tmp[i]);
  }
  if (type == 1) {
    if ((rc = pool_floats(ctx, cnt, &mid))) return rc;
    defer_free(ctx, mid);
  }
  const float* cur = x;
  auto conv = [&](const float* src, const float* w, const float* b, int dil, const float* res, float* dst) -> int {
    ConvArgs a;
    a.x = src; a.bias = b; a.res = res; a.y = dst;
    a.N = (int)n; a.Cin = C; a.Cout = C; a.K = K; a.dil = dil; a.padL = (K * dil - dil) / 2; a.Lin = T; a.Lout = T;
    a.x_batch_stride = (int64_t)C * T; a.y_batch_stride = (int64_t)C * T; a.y_len = T;
    a.prologue = PRO_LRELU; a.alpha = lrelu_slope;
    if (mfma) {
      const int tm = conv_pick_tile(ctx, C, T, (int)n, 0);
      float* packed = nullptr;
      int r = pool_floats(ctx, packed_conv_floats(C, C, K, tm), &packed);
      if (r) return r;
      defer_free(ctx, packed);
      pack_conv_weights(ss.s, w, C, C, K, packed, tm);
      if (tm == 16) a.w16 = packed; else a.w = packed;
      return launch_conv_mfma(ctx, ss.s, a);
    }
    a.w = w;
    return launch_conv_direct(ctx, ss.s, a);
  };
  // two chained convs per launch (rb_pair.hip) where the geometry allows: ResBlock1 — every (convs1[i], convs2[i]) pair;
  // ResBlock2 — steps (i, i+1)
  static const bool no_pair = getenv("PIPER_HIP_NO_RB_PAIR") != nullptr;
  auto pair = [&](const float* src, int ia, int da, int ib, int db, bool res_a, bool res_b_x, float* dst) -> int {
    RbPairArgs pa;
    float* pk[2];
    for (int q = 0; q < 2; q++) {
      int r = pool_floats(ctx, packed_conv_win_floats(C, C, K), &pk[q]);
      if (r) return r;
      defer_free(ctx, pk[q]);
      pack_conv_weights_win(ss.s, weights[q ? ib : ia], C, C, K, pk[q]);
    }
    pa.x = src; pa.y = dst; pa.wa4 = pk[0]; pa.ba = biases[ia]; pa.wb4 = pk[1]; pa.bb = biases[ib];
    pa.Ka = K; pa.dila = da; pa.Kb = K; pa.dilb = db; pa.res_a = res_a; pa.res_b_x = res_b_x; pa.alpha = lrelu_slope;
    pa.N = (int)n; pa.C = C; pa.L = T;
    return launch_rb_pair_multi(ctx, ss.s, &pa, 1);
  };
  const bool slope_ok = lrelu_slope > 0.0f && lrelu_slope < 1.0f;
  auto next_buf = [&](bool last) { return last ? *out : (cur == tmp[0] ? tmp[1] : tmp[0]); };  // never the buffer being read
  for (int i = 0; i < n_dil; i++) {
    float* dst = next_buf(i == n_dil - 1);
    if (dilations[i] < 1) PH_FAIL(PIPER_HIP_ERR_SHAPE, "hifigan_resblock: dilation must be >= 1");
    if (!no_pair && slope_ok && type == 1 && biases[2 * i] && biases[2 * i + 1] && rb_pair_eligible(C, K, dilations[i], K, 1, T)) {
      if ((rc = pair(cur, 2 * i, dilations[i], 2 * i + 1, 1, false, true, dst))) return rc;
      cur = dst;
      continue;
    }
    if (!no_pair && slope_ok && type == 2 && i + 1 < n_dil && dilations[i + 1] >= 1 && biases[i] && biases[i + 1] &&
        rb_pair_eligible(C, K, dilations[i], K, dilations[i + 1], T)) {
      float* d2 = next_buf(i + 1 == n_dil - 1);
      if ((rc = pair(cur, i, dilations[i], i + 1, dilations[i + 1], true, false, d2))) return rc;
      cur = d2;
      i++;
      continue;
    }
    if (type == 1) {
      if ((rc = conv(cur, weights[2 * i], biases[2 * i], dilations[i], nullptr, mid))) return rc;
      if ((rc = conv(mid, weights[2 * i + 1], biases[2 * i + 1], 1, cur, dst))) return rc;
    } else {
      if ((rc = conv(cur, weights[i], biases[i], dilations[i], cur, dst))) return rc;
    }
    cur = dst;
  }
  return ss.finish("hifigan_resblock_f32");
}
