/*
 * piper_hip.h — C-ABI of the MI355X (gfx950) operator backend for Piper VITS inference.
 *
 * This is the drop-in boundary for the slot `Sources/PiperMetal` fills in ocrickard/piper-swift:
 * one entry point per `MetalBackend` per-op method that `GraphExecutor.executeNode` calls on the
 * hot path, plus fused / whole-utterance entry points whose results equal the
 * unfused composition.  Every declaration cites the reference interface it replaces (file:line,
 * relative to the reference repo root).
 *
 * Conventions (mirroring MetalBackend):
 *   - Tensors are dense row-major float32 on the device; shapes are int64_t arrays.
 *   - The callee ALLOCATES and returns the output buffer (MetalBackend returns a fresh MTLBuffer,
 *     MetalBackend.swift:1184, 1261, 1334).  If `*out` is non-NULL on entry it is used instead
 *     (caller-owned, must hold the output) — the zero-allocation path for static schedules.
 *   - Inputs are borrowed and never written.
 *   - `stream == NULL`  ⇔ `commandBuffer == nil`: run and BLOCK until complete
 *     (MetalBackend.swift:1223-1226).  Non-NULL ⇔ encode only; `piper_hip_stream_sync` ⇔ `flush`.
 *   - Swift `throws` → int status (0 ok, negative below) + thread-local `piper_hip_last_error()`.
 *   - One context per GPU, used from one host thread at a time (the reference has one queue, one
 *     executor, no locks: GraphExecutor.swift:27, MetalContext.swift:6,14).
 *   - No CPU fallback exists anywhere behind this header: without a gfx950 device every compute
 *     entry point returns PIPER_HIP_ERR_UNAVAILABLE.
 */
#ifndef PIPER_HIP_H
#define PIPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIPER_HIP_ABI_VERSION 3

/* ---- status codes: ExecutionError (CPUBackend.swift:3-17) + NSError domain "MetalBackend" ---- */
enum {
  PIPER_HIP_OK = 0,
  PIPER_HIP_ERR_SHAPE = -1,       /* ExecutionError.shapeMismatch */
  PIPER_HIP_ERR_TYPE = -2,        /* ExecutionError.typeMismatch */
  PIPER_HIP_ERR_UNSUPPORTED = -3, /* ExecutionError.unsupportedOp */
  PIPER_HIP_ERR_UNAVAILABLE = -4, /* ExecutionError.metalUnavailable: no gfx950 device / HIP init failed */
  PIPER_HIP_ERR_ALLOC = -5,       /* NSError 600/610/620/760/904: output allocation failed */
  PIPER_HIP_ERR_LAUNCH = -6,      /* NSError 601/611/621/761/900: encode / execution failed */
  PIPER_HIP_ERR_ARG = -7          /* NULL / invalid handle (Swift's type system excludes these) */
};

typedef struct piper_hip_ctx piper_hip_ctx;     /* MetalBackend + MetalContext (device, queue, pipelines) */
typedef struct piper_hip_voice piper_hip_voice; /* a loaded voice: weights resident + static schedule */
typedef struct piper_hip_comm piper_hip_comm;   /* an RCCL communicator: one rank (= one GPU, one process) of a node */
typedef void* piper_hip_stream;                 /* hipStream_t; stands in for MTLCommandBuffer? */

/* Thread-local description of the last failure on this thread ("" if none). */
const char* piper_hip_last_error(void);
int piper_hip_abi_version(void);
/* The tuning / A-B switches (PIPER_HIP_* environment variables) this process has honoured so far, as "NAME=value …"; empty
 * when none. Switches are honoured only when PIPER_HIP_TUNING=1 is set too: an inherited environment does not change which kernels run. */
int piper_hip_config_string(char* buf, size_t n);
/* Number of visible HIP devices (0 when none); never initialises a device. */
int piper_hip_device_count(void);

/* ---- context, buffers, transfers ---- */
/* MetalContext.init (Metal/MetalContext.swift:9-33) + MetalBackend.init (MetalBackend.swift:12-15). */
int piper_hip_create(int device, piper_hip_ctx** out);
void piper_hip_destroy(piper_hip_ctx* ctx);
/* Device memory held by the context's pool (buffers + cached free blocks) and the part currently handed out. The pool
 * recycles power-of-two blocks, so a host that sees many utterance shapes plateaus instead of growing; `memory_trim`
 * returns the cached free blocks to the driver (the role ARC plays for MTLBuffer in the reference). */
int piper_hip_memory_stats(piper_hip_ctx* ctx, size_t* reserved_bytes, size_t* live_bytes);
int piper_hip_memory_trim(piper_hip_ctx* ctx);
/* Take `bytes` of device memory from the driver NOW, in one piece, and serve later allocations of this context (plan arenas, op outputs) from
 * it before asking the driver again. A fresh driver allocation is cheap to make but its first use can stall the GPU for tens of milliseconds
 * (page mapping / clearing): a serving process pays that while it loads. One slab per context (later calls are no-ops); piper_hip_voice_create
 * reserves 8 GiB (of the 288 GB of an MI355X; halved until it fits on a smaller part) if nothing was reserved before it (the reference's heaps: MetalBackend.swift:34-39 allocates per buffer). */
int piper_hip_memory_reserve(piper_hip_ctx* ctx, size_t bytes);
/* MetalBackend.allocateBuffer(length:) (MetalBackend.swift:34-39): at least 1 byte is allocated. */
int piper_hip_alloc(piper_hip_ctx* ctx, size_t bytes, void** out);
/* Buffer release (ARC drop in the reference; GraphExecutor.swift:216-225). Returns memory to the
 * context pool; stream-ordered with respect to work already enqueued on the context's streams. */
int piper_hip_free(piper_hip_ctx* ctx, void* buf);
/* MetalBackend.uploadFloat32 (MetalBackend.swift:983-993). */
int piper_hip_upload_f32(piper_hip_ctx* ctx, const float* host, size_t count, float** out);
int piper_hip_upload_i64(piper_hip_ctx* ctx, const int64_t* host, size_t count, int64_t** out);
/* MetalBackend.downloadFloat32 (MetalBackend.swift:963-981): blocks until `buf` is complete. */
int piper_hip_download_f32(piper_hip_ctx* ctx, const float* buf, float* host, size_t count);
/* Page-locked host memory — the analogue of reading `MTLBuffer.contents()` of a shared-storage buffer (MetalBackend.swift:963-981 copies out
 * of one): piper_hip_voice_collect / piper_hip_download_f32 into such a buffer is a single DMA, without the staging copy a pageable
 * destination needs. Plain malloc'ed buffers keep working everywhere. */
int piper_hip_host_alloc(piper_hip_ctx* ctx, size_t bytes, void** out);
int piper_hip_host_free(piper_hip_ctx* ctx, void* host);
/* MetalBackend.makeCommandBuffer / flush / flushWithTimings (MetalBackend.swift:841-874). */
int piper_hip_stream_create(piper_hip_ctx* ctx, piper_hip_stream* out);
int piper_hip_stream_destroy(piper_hip_ctx* ctx, piper_hip_stream s);
int piper_hip_stream_sync(piper_hip_ctx* ctx, piper_hip_stream s);
/* GPU time bracket on a stream (gpuStartTime/gpuEndTime, MetalBackend.swift:868-873). */
int piper_hip_timer_begin(piper_hip_ctx* ctx, piper_hip_stream s);
int piper_hip_timer_end(piper_hip_ctx* ctx, piper_hip_stream s, double* gpu_ms); /* syncs the stream */

/* ---- param PODs (MetalBackend.swift:876-961; MSL twins conv1d.metal:12-26,80-95) ---- */
typedef struct {
  int32_t stride, dilation, pad_l, pad_r, groups;
} piper_hip_conv1d_params;
typedef struct {
  int32_t stride, dilation, pad_l, pad_r, output_padding, groups;
} piper_hip_convtranspose1d_params;

typedef enum { /* unaryF32 / unaryAlphaF32 kernels (MetalBackend.swift:1501-1586; elementwise.metal:165-312) */
  PIPER_HIP_RELU = 0,
  PIPER_HIP_LEAKYRELU = 1, /* alpha */
  PIPER_HIP_TANH = 2,
  PIPER_HIP_SIGMOID = 3,
  PIPER_HIP_EXP = 4,
  PIPER_HIP_NEG = 5,
  PIPER_HIP_SQRT = 6,
  PIPER_HIP_SOFTPLUS = 7,
  PIPER_HIP_CEIL = 8,
  PIPER_HIP_ERF = 9
} piper_hip_unary_op;
typedef enum { /* add/sub/mul/div/pow broadcast kernels (MetalBackend.swift:2592-2610; elementwise.metal:52-130) */
  PIPER_HIP_ADD = 0,
  PIPER_HIP_SUB = 1,
  PIPER_HIP_MUL = 2,
  PIPER_HIP_DIV = 3,
  PIPER_HIP_POW = 4
} piper_hip_binary_op;

/* ---- per-op entry points (§8a rows a3-a9) ---- */

/* MetalBackend.conv1dF32 (MetalBackend.swift:1149-1228) → conv1d_f32 (conv1d.metal:28-71);
 * arithmetic spec CPUBackend.conv1d (CPUBackend.swift:20-73).
 * x [N,Cin,L], w [Cout,Cin/g,K], bias [Cout] or NULL → y [N,Cout,L_out],
 * L_out = (L + pad_l + pad_r − dilation·(K−1) − 1)/stride + 1 (integer division). */
int piper_hip_conv1d_f32(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], const float* w,
                         const int64_t w_shape[3], const float* bias, const piper_hip_conv1d_params* p,
                         float** out, int64_t out_shape[3], piper_hip_stream stream);

/* MetalBackend.convTranspose1dF32 (MetalBackend.swift:2812-2895) → convtranspose1d_f32
 * (conv1d.metal:97-142). w is ONNX layout [Cin, Cout/g, K].
 * L_out = (L−1)·stride − pad_l − pad_r + dilation·(K−1) + output_padding + 1 (must be > 0). */
int piper_hip_convtranspose1d_f32(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], const float* w,
                                  const int64_t w_shape[3], const float* bias,
                                  const piper_hip_convtranspose1d_params* p, float** out, int64_t out_shape[3],
                                  piper_hip_stream stream);

/* bf16-operand variants of the two contractions (SURVEY.md §8b "bf16 variants", §8d config 5). The reference has no
 * bf16 path, so these are a build extension with the SAME shapes, parameter structs and error behaviour as
 * conv1dF32 / convTranspose1dF32 (MetalBackend.swift:1149-1228, 2812-2895): fp32 tensors in and out, x and w rounded
 * to bf16 (nearest even) on the device, fp32 accumulation and bias on v_mfma_f32_32x32x16_bf16. Covered geometry:
 * stride 1 (conv) / K % stride == 0 with pads (K−stride)/2 (convT), groups 1, C_in % 32 == 0 (convT: C_out too),
 * padding and dilation reach ≤ 64; anything else returns PIPER_HIP_ERR_UNSUPPORTED so the caller can take the f32 op. */
int piper_hip_conv1d_bf16(piper_hip_ctx* ctx, const float* input, const int64_t input_shape[3], const float* weight,
                          const int64_t weight_shape[3], const float* bias, const piper_hip_conv1d_params* p, float** out,
                          int64_t out_shape[3], piper_hip_stream stream);
int piper_hip_convtranspose1d_bf16(piper_hip_ctx* ctx, const float* input, const int64_t input_shape[3],
                                   const float* weight, const int64_t weight_shape[3], const float* bias,
                                   const piper_hip_convtranspose1d_params* p, float** out, int64_t out_shape[3],
                                   piper_hip_stream stream);

/* MetalBackend.matmulF32 (MetalBackend.swift:1232-1323) → matmul_f32 (matmul.metal:22-49), with the
 * executor's rank-4 lead-dim broadcast (GraphExecutor.swift:1870-1899) done by stride-0 addressing
 * instead of a materialised expandF32.  Equal ranks ≥ 2 required (MetalBackend.swift:1236-1238);
 * lead dims must be equal or 1-broadcastable. */
int piper_hip_matmul_f32(piper_hip_ctx* ctx, const float* a, const int64_t* a_shape, const float* b,
                         const int64_t* b_shape, int rank, float** out, int64_t* out_shape,
                         piper_hip_stream stream);

/* MetalBackend.softmaxLastDimF32 (MetalBackend.swift:1326-1355) → softmax_lastdim_f32
 * (softmax.metal:13-41). Last dim must be > 0. */
int piper_hip_softmax_lastdim_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank, float** out,
                                  piper_hip_stream stream);

/* MetalBackend.{relu,leakyRelu,tanh,sigmoid,exp,neg,sqrt,softplus,ceil,erf}F32
 * (MetalBackend.swift:1501-1586). `alpha` is read only by PIPER_HIP_LEAKYRELU. */
int piper_hip_unary_f32(piper_hip_ctx* ctx, piper_hip_unary_op op, const float* x, size_t count, float alpha,
                        float** out, piper_hip_stream stream);

/* MetalBackend.{add,sub,mul,div,pow}F32 → binaryBroadcastF32 (MetalBackend.swift:2099-2134, 2592-2610);
 * NumPy broadcasting, output rank ≤ 4 (MetalBackend.swift:2065-2067). out_shape has max(ra,rb) entries. */
int piper_hip_binary_broadcast_f32(piper_hip_ctx* ctx, piper_hip_binary_op op, const float* a,
                                   const int64_t* a_shape, int a_rank, const float* b, const int64_t* b_shape,
                                   int b_rank, float** out, int64_t* out_shape, int* out_rank,
                                   piper_hip_stream stream);

/* Layout ops the rel-position skew and the flow coupling are made of (§8a rows a7, a9). */
/* MetalBackend.padConstantF32 (MetalBackend.swift:780-839) → pad_constant_f32_rank4 (pad.metal).
 * pads = [begin_0..begin_{r-1}, end_0..end_{r-1}], all ≥ 0, rank ≤ 4. */
int piper_hip_pad_constant_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank,
                               const int64_t* pads, float value, float** out, int64_t* out_shape,
                               piper_hip_stream stream);
/* Slice arms (GraphExecutor.swift:1322-1426) → slice.metal kernels. One axis, start/end already
 * clamped ONNX-style by the caller, step ≠ 0 (step −1 on axis 1 is VITS `Flip`). rank ≤ 4. */
int piper_hip_slice_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank, int axis, int64_t start,
                        int64_t end, int64_t step, float** out, int64_t* out_shape, piper_hip_stream stream);
/* MetalBackend.transposeF32 (MetalBackend.swift:995-1060) → transpose_f32_rank4 (transpose.metal:16-46). */
int piper_hip_transpose_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank, const int32_t* perm,
                            float** out, int64_t* out_shape, piper_hip_stream stream);
/* concat2_axis1_ncl_f32 / split2_axis1_ncl_f32 (tensorops.metal; GraphExecutor.swift:1107-1124, 2254-2262). */
int piper_hip_concat2_axis1_f32(piper_hip_ctx* ctx, const float* a, const int64_t a_shape[3], const float* b,
                                const int64_t b_shape[3], float** out, int64_t out_shape[3],
                                piper_hip_stream stream);
int piper_hip_split2_axis1_f32(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], int64_t c0,
                               float** out0, float** out1, piper_hip_stream stream);
/* MetalBackend.expandF32 (MetalBackend.swift:2438-2458) → expand_f32_rank4 (expand.metal). */
int piper_hip_expand_f32(piper_hip_ctx* ctx, const float* x, const int64_t* in_shape, const int64_t* out_shape,
                         int rank, float** out, piper_hip_stream stream);
/* reduce_mean_lastdim_f32 (reduce.metal:12-25; MetalBackend.swift:1357-1390) — the LayerNorm building block. */
int piper_hip_reduce_mean_lastdim_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank,
                                      float** out, piper_hip_stream stream);

/* MetalBackend.randomNormalLike(shape:seed:) (MetalBackend.swift:3398-3426) → random_normal_like_f32 (elementwise.metal:139-163):
 * element i draws u0, u1 from xorshift32 seeded with (seed & 0xffffffff) ^ (i·747796405 + 2891336453), maps them to (0, 1] and
 * returns sqrt(−2·ln u0)·cos(2π·u1). The reference calls it with the fixed seed 1234 for both RandomNormalLike nodes of the
 * graph (GraphExecutor.swift:2656-2659). The integer stream is bit-exact; the floats agree to ≈ 1e-6 (Metal's log/cos are not
 * libm's). */
int piper_hip_random_normal_like_f32(piper_hip_ctx* ctx, size_t count, uint64_t seed, float** out, piper_hip_stream stream);
/* The raw draws behind it, for parity checks of the integer stream: out[2i] = u0, out[2i+1] = u1 of element i (device buffer). */
int piper_hip_random_draws_u32(piper_hip_ctx* ctx, size_t count, uint64_t seed, uint32_t** out, piper_hip_stream stream);

/* ---- fused entry points: results equal the unfused composition of the ops above ---- */

/* Text-encoder relative-position self-attention core (§8a rows a5+a6+a7), one call per layer:
 *   scores = (q/√d)·kᵀ + rel→abs( (q/√d)·E_kᵀ );  p = softmax(scores);  out = p·v + abs→rel(p)·E_v
 * q,k,v: [N, H·d, T] channel-major (the k=1 Conv outputs, head h = channels [h·d,(h+1)·d));
 * emb_rel_k/emb_rel_v: [2·window+1, d] shared over heads; out: [N, H·d, T].
 * Replaces the MatMul/Pad/Reshape/Slice/Softmax/Transpose chain of GraphExecutor.swift:1862-1929,
 * 1130-1210, 1371-1426, 901-947 without materialising [N,H,T,2T−1]. */
int piper_hip_rel_attention_f32(piper_hip_ctx* ctx, const float* q, const float* k, const float* v,
                                const float* emb_rel_k, const float* emb_rel_v, int64_t n, int64_t heads,
                                int64_t head_dim, int64_t t, int64_t window, float** out, piper_hip_stream stream);

/* The attention block of an encoder layer in ONE launch: out = LN_c(x + conv_o(rel_attention(q, k, v)))·gamma + beta, with
 * conv_o the k = 1 output projection (w_o [C, C, 1], b_o [C], C = heads·head_dim). Equals rel_attention_f32 → conv1d_f32 →
 * add_layernorm_f32 (GraphExecutor.swift:1862-1929 + Conv :1739-1810 + Add :741-779 + LayerNorm chain :2071-2125) without the
 * two intermediate tensors. Covered: head_dim 96, C ≤ 256, window ≤ 7, T ≤ 2048; otherwise PIPER_HIP_ERR_UNSUPPORTED and the
 * caller composes the three ops. */
int piper_hip_attention_block_f32(piper_hip_ctx* ctx, const float* q, const float* k, const float* v, const float* emb_rel_k,
                                  const float* emb_rel_v, const float* w_o, const float* b_o, const float* x, const float* gamma,
                                  const float* beta, int64_t n, int64_t heads, int64_t head_dim, int64_t t, int64_t window, float eps,
                                  float** out, piper_hip_stream stream);

/* Channel LayerNorm with fused residual: out = LN_c(x + y)·gamma + beta over C for each (n,t); y may be
 * NULL. Replaces Transpose + ReduceMean/Sub/Pow/Sqrt/Div/Mul/Add (GraphExecutor.swift:2071-2125). */
int piper_hip_add_layernorm_f32(piper_hip_ctx* ctx, const float* x, const float* y, const float* gamma,
                                const float* beta, int64_t n, int64_t c, int64_t t, float eps, float** out,
                                piper_hip_stream stream);

/* One WaveNet layer of the flow (§8a rows a3+a8): acts = tanh(a)·sigmoid(b), [a;b] = in_conv(x) (C→2C, k, dilation);
 * rs = res_skip_conv(acts) (k=1; C→2C, or C→C when `last`). If !last: x_out = x + rs[:C], skip_out = skip_in + rs[C:];
 * else skip_out = skip_in + rs. skip_in may be NULL (treated as 0). x_out/skip_out follow the `*out` convention
 * (x_out is not written when `last`). */
int piper_hip_wavenet_layer_f32(piper_hip_ctx* ctx, const float* x, const float* skip_in, const float* w_in,
                                const float* b_in, const float* w_rs, const float* b_rs, int64_t n, int64_t c,
                                int64_t t, int64_t k, int64_t dilation, int last, float** x_out, float** skip_out,
                                piper_hip_stream stream);

/* HiFi-GAN residual blocks (§8a rows a3+a8). type 1 (Piper "high"): for each dilation d_i:
 * x = x + conv2_i(lrelu(conv1_i(lrelu(x)), d=d_i), d=1); type 2 (Piper "medium"): x = x + conv_i(lrelu(x), d=d_i).
 * slope 0.1 (LeakyRelu alpha), `same` padding (k·d−d)/2. weights: n_dil (type 2) or 2·n_dil (type 1, ordered
 * c1_0,c2_0,c1_1,…) tensors [C,C,K] each followed by its bias [C], as arrays of device pointers on the host. */
int piper_hip_hifigan_resblock_f32(piper_hip_ctx* ctx, int type, const float* x, int64_t n, int64_t c, int64_t t,
                                   int64_t k, const int32_t* dilations, int n_dil, const float* const* weights,
                                   const float* const* biases, float lrelu_slope, float** out,
                                   piper_hip_stream stream);

/* ---- whole-utterance path: PiperMetalRuntime.synthesize (PiperMetalRuntime.swift:62-80) ---- */

#define PIPER_HIP_MAX_UPS 4
#define PIPER_HIP_MAX_RB 3
typedef struct {
  int32_t n_vocab;    /* embedding rows (Piper num_symbols, 256) */
  int32_t hidden;     /* 192 */
  int32_t n_heads;    /* 2 */
  int32_t n_layers;   /* 6 */
  int32_t ffn;        /* 768 */
  int32_t ffn_kernel; /* 3 */
  int32_t window;     /* 4 */
  int32_t inter;      /* 192 */
  int32_t n_flows;    /* 4 */
  int32_t wn_layers;  /* 4 */
  int32_t wn_kernel;  /* 5 */
  int32_t up_initial; /* 256 (medium) / 512 (high) */
  int32_t n_ups;      /* 3 / 4 */
  int32_t up_rates[PIPER_HIP_MAX_UPS];
  int32_t up_kernels[PIPER_HIP_MAX_UPS];
  int32_t resblock_type; /* 2 (medium) / 1 (high) */
  int32_t n_rb;          /* resblocks per stage (3) */
  int32_t rb_kernels[PIPER_HIP_MAX_RB];
  int32_t rb_n_dil; /* dilations per resblock: 2 (medium) / 3 (high) */
  int32_t rb_dilations[PIPER_HIP_MAX_RB][3];
  int32_t sample_rate; /* 22050 */
  /* ---- ABI 2: stochastic duration predictor (VITS `dp`; the Softplus / CumSum / GatherElements / … spline arms of
   * GraphExecutor.swift:2379-2645). dp_present = 0 ⇒ the blob carries no `dp.*` tensors and per-id durations must be
   * supplied with every utterance. */
  int32_t dp_present;    /* 1 */
  int32_t dp_kernel;     /* 3: depthwise kernel of the dilated depth-separable convs; dilation kernel^i */
  int32_t dp_dds_layers; /* 3 */
  int32_t dp_n_flows;    /* 4 ConvFlows in the module; inference runs the last dp_n_flows − 1 of them + the ElementwiseAffine */
  int32_t dp_bins;       /* 10 spline bins */
  float dp_tail_bound;   /* 5.0 */
} piper_hip_voice_config;

/* Piper medium / high geometry (SURVEY.md §8a †). quality: 0 = medium, 1 = high. */
int piper_hip_voice_config_preset(int quality, piper_hip_voice_config* out);
/* Number of floats in the packed weight blob for `cfg` (layout: include/piper_hip_voice_layout.h). */
int piper_hip_voice_blob_floats(const piper_hip_voice_config* cfg, size_t* n_floats);
/* One tensor of the blob (order = include/piper_hip_voice_layout.h). kind: 0 weight, 1 bias, 2 gamma, 3 beta, 4 embedding. */
typedef struct {
  char name[96];
  int32_t kind;
  int32_t rank;
  int64_t shape[3];
  int64_t fan_in;
  uint64_t offset; /* floats */
  uint64_t count;  /* floats */
} piper_hip_tensor_info;
int piper_hip_voice_blob_layout(const piper_hip_voice_config* cfg, piper_hip_tensor_info* out, int max_entries,
                                int* n_entries);
/* Fill a HOST blob with synthetic weights: the bench has no real voice offline. Counter-based
 * SplitMix64 → 24-bit uniform, zero-mean with the variance SURVEY.md §8d asks for (weights/embeddings
 * U(±√(3/fan_in)) ⇒ var 1/fan_in; biases U(±0.01·√3); gamma 1+U(±0.1); beta U(±0.1)) — uniform instead of
 * Box-Muller so that C and numpy (tests/katdata.py) produce bit-identical blobs. Host-only, no GPU needed. */
int piper_hip_voice_synthetic_blob(const piper_hip_voice_config* cfg, uint64_t seed, float* host_blob,
                                   size_t n_floats);
/* Load a voice from a packed fp32 blob; `on_device` ≠ 0 when `blob` is a device pointer (e.g. the RCCL-broadcast
 * copy). Packs MFMA weight fragments once; the blob itself is not retained. ⇔ GraphExecutor.init + persistent
 * initializer buffers (GraphExecutor.swift:42-71, 279-283). */
int piper_hip_voice_create(piper_hip_ctx* ctx, const piper_hip_voice_config* cfg, const float* blob, int on_device,
                           piper_hip_voice** out);
void piper_hip_voice_destroy(piper_hip_voice* v);

/* Arithmetic of the HiFi-GAN generator (94 % of the high voice's FLOPs). F32 (default): everything fp32, the parity
 * configuration. BF16: every generator Conv / ConvTranspose takes bf16 operands (weights, and
 * the LeakyReLU'd activations written by the producing conv) with fp32 accumulation; bias, residual stream, MRF mean,
 * conv_post, the text encoder and the flow stay fp32. Stated tolerance: waveform SNR ≥ 35 dB against the fp32 result.
 * Drops every prepared slot (prepare again). UNSUPPORTED when a generator conv is outside the bf16 kernels' geometry. */
#define PIPER_HIP_PRECISION_F32 0
#define PIPER_HIP_PRECISION_BF16 1
int piper_hip_voice_set_precision(piper_hip_voice* v, int precision);
int piper_hip_voice_precision(const piper_hip_voice* v);

/* ---- Piper `.onnx` / `.onnx.json` → voice blob (role of Sources/PiperONNX for this library) ----
 * Host-only (no GPU). Decodes the protobuf wire subset the reference's loader decodes (ONNXLoader.swift:34-37, 94-99,
 * 170-176, 214-223, 321-327): initializers (name, dims, data_type, float_data / raw_data little-endian, the flat order of
 * TensorValue.swift:45-116) and the Conv / ConvTranspose node attributes (strides, dilations). `infer_config` derives the
 * voice geometry from initializer shapes + those attributes; `build_blob` emits the initializers in the order of
 * include/piper_hip_voice_layout.h (names = the ONNX initializer names, e.g. `enc_p.encoder.attn_layers.0.conv_q.weight`,
 * ONNXParsingTests.swift:32), folding `weight_g`/`weight_v` pairs if the export kept weight norm. Tested against ONNX files
 * WRITTEN by the test-suite (no Piper voice exists offline): real-voice naming is unpinned. */
typedef struct piper_hip_onnx piper_hip_onnx;
typedef struct {
  char name[128];
  int32_t data_type; /* ONNX TensorProto.DataType: 1 FLOAT, 6 INT32, 7 INT64, 9 BOOL */
  int32_t rank;
  int64_t dims[8];
  int64_t count;
} piper_hip_onnx_tensor_info;
int piper_hip_onnx_open(const char* path, piper_hip_onnx** out); /* mmap: weights are copied once, by build_blob */
int piper_hip_onnx_open_memory(const void* data, size_t size, piper_hip_onnx** out);
void piper_hip_onnx_close(piper_hip_onnx* m);
/* ir_version, default-domain opset, node and initializer counts (ONNXParsingTests.swift:22-36 pins 15 / 2755 / 401). */
int piper_hip_onnx_counts(const piper_hip_onnx* m, int64_t* ir_version, int64_t* opset, int* n_nodes, int* n_initializers);
int piper_hip_onnx_initializer(const piper_hip_onnx* m, int index, piper_hip_onnx_tensor_info* out);
int piper_hip_onnx_find(const piper_hip_onnx* m, const char* name); /* initializer index or -1 */
int piper_hip_onnx_read_f32(const piper_hip_onnx* m, int index, float* dst, size_t n);
int piper_hip_onnx_infer_config(const piper_hip_onnx* m, piper_hip_voice_config* cfg);
/* Is the node graph the computation this library's fixed launch schedule performs for `cfg`? The reference executes whatever the graph
 * says (GraphExecutor.swift:227-265); this library does not look at the nodes when it runs, so it looks at them HERE: header (opset 15,
 * I/O names, Gather first — ONNXParsingTests.swift:22-36), op census (the 50 arms of GraphExecutor.swift:592-2659), every Conv /
 * ConvTranspose's attributes and effective padding, the attention / skew / softmax / LayerNorm(eps) / FFN structure per encoder layer,
 * Flips and tanh·sigmoid gates of the flow, LeakyRelu slopes, residual Adds and the MRF mean of the generator, the RandomNormalLike count.
 * PIPER_HIP_ERR_UNSUPPORTED + a message naming the first differing node otherwise (csrc/onnx_verify.cpp). */
int piper_hip_onnx_verify_graph(const piper_hip_onnx* m, const piper_hip_voice_config* cfg);
/* build_blob = verify_graph, then the initializers in layout order. _unchecked skips the verification: for weight containers that carry
 * no graph; the caller vouches that the weights belong to a standard Piper VITS. */
int piper_hip_onnx_build_blob(const piper_hip_onnx* m, const piper_hip_voice_config* cfg, float* host_blob, size_t n_floats);
int piper_hip_onnx_build_blob_unchecked(const piper_hip_onnx* m, const piper_hip_voice_config* cfg, float* host_blob, size_t n_floats);
/* The numbers this library needs from the voice's `.onnx.json` (PiperConfig.swift:3-47). */
typedef struct {
  int32_t sample_rate, num_symbols, num_speakers;
  float noise_scale, length_scale, noise_w;
} piper_hip_piper_json_info;
int piper_hip_piper_json(const char* json_text, piper_hip_piper_json_info* out);
/* Is the voice the `.onnx.json` describes one this library renders correctly with geometry `cfg`? ERR_UNSUPPORTED for
 * num_speakers > 1 (speaker conditioning is not implemented: the audio would be wrong without any error), ERR_SHAPE when
 * num_symbols differs from the embedding rows of the graph. */
int piper_hip_voice_check_json(const piper_hip_voice_config* cfg, const piper_hip_piper_json_info* info);

/* Waveform → 16-bit PCM / mono WAV (WavFileWriter.swift:20-30, 44-60): clamp to [−1,1], ×32767, truncate toward zero. */
int piper_hip_pcm16_from_f32(const float* samples, size_t n, int16_t* pcm);
int piper_hip_wav_write(const char* path, const float* samples, size_t n, int32_t sample_rate);

/* Inputs of one utterance ⇔ ExecutionInputs + overrides (GraphExecutor.swift:5-15, 101-104): phoneme ids, the three
 * `scales` of PiperMetalRuntime.synthesize (PiperMetalRuntime.swift:62-80), and the tensors the reference lets a caller
 * pre-seed by name — the duration predictor's output (per-id frame counts) and the two RandomNormalLike tensors
 * (PiperTestVector.swift:24-29: `dp` [1,2,T], `main` [1,inter,F]).
 * ABI 2 appended everything below `noise_scale`; zero-initialising the struct keeps the ABI-1 behaviour. */
#define PIPER_HIP_NOISE_INJECTED 0 /* NULL noise pointers mean zeros (deterministic parity runs) */
#define PIPER_HIP_NOISE_DEVICE 1   /* NULL noise pointers are generated on the device by the reference's RandomNormalLike
                                      stream (piper_hip_random_normal_like_f32) with `seed`; non-NULL pointers still win */
typedef struct {
  const int64_t* phoneme_ids; /* [T] host */
  int32_t t;
  const int32_t* durations; /* [T] host, frames per id (≥0), F = Σ durations — the `overrides` route; NULL: predicted on the device
                               by the voice's stochastic duration predictor from length_scale / noise_w / dp_noise (ABI 2) */
  const float* noise;       /* `main` RandomNormalLike [inter, F] host, may be NULL (see noise_mode). Must be NULL when durations is NULL
                               (F is not known to the caller then: PIPER_HIP_ERR_ARG; predict_durations first, or noise_mode DEVICE) */
  float noise_scale;        /* scales[0] */
  /* ---- ABI 2 ---- */
  int32_t noise_mode;       /* PIPER_HIP_NOISE_INJECTED / PIPER_HIP_NOISE_DEVICE */
  uint32_t seed;            /* PIPER_HIP_NOISE_DEVICE: RandomNormalLike seed; the reference hard-codes 1234 (GraphExecutor.swift:2658) */
  float length_scale;       /* scales[1]: w = exp(logw)·length_scale (only read when durations == NULL; 0 is taken as 1.0) */
  float noise_w;            /* scales[2]: scale of the predictor's latent noise (only read when durations == NULL) */
  const float* dp_noise;    /* `dp` RandomNormalLike [2, T] host, may be NULL (see noise_mode; both RandomNormalLike nodes of the
                               reference draw from the SAME seed, so device mode reproduces that) */
} piper_hip_utterance;

/* Samples `synthesize` will produce for this utterance (F · Π up_rates); −1 on a bad utterance, −2 when the durations are to
 * be predicted (unknown before `prepare`: ask piper_hip_voice_prepared_samples afterwards). */
int64_t piper_hip_voice_num_samples(const piper_hip_voice* v, const piper_hip_utterance* u);
/* After prepare: samples of each batch item of the slot (n_items entries) and/or their sum. */
int piper_hip_voice_prepared_samples(const piper_hip_voice* v, int slot, int64_t* per_item, int max_items, int64_t* total);
/* After prepare: the frames-per-id actually used (supplied or predicted), items back to back (Σ T_b entries). */
int piper_hip_voice_durations(const piper_hip_voice* v, int slot, int32_t* out, int max_entries, int* n_entries);
/* Attach a plan for this utterance's bucket to `slot` (building and caching it if the voice has none idle) and upload the
 * utterance's inputs; returns the slot id ≥ 0. Plans own a stream and an arena, so several prepared slots can be launched
 * back-to-back and overlap on the GPU. */
int piper_hip_voice_prepare(piper_hip_voice* v, const piper_hip_utterance* u, int slot);
/* The same for `n` utterances in ONE schedule whose kernels carry a batch dimension, so a launch costs what one utterance
 * costs in dispatches. The items may differ in length (ragged batch): the schedule is the bucket of the longest item, every
 * length-aware kernel reads each item's true lengths from device memory — taps past an item's end read as zero padding and
 * attention excludes the keys past it, i.e. the reference graph's x_mask / y_mask semantics — so each item's result is what
 * it would be alone, and tiles entirely past an item's end cost (almost) nothing. Group items of similar length to limit the
 * padding. `collect` returns the n waveforms back to back, each at its own length. */
int piper_hip_voice_prepare_batch(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int slot);
/* The same for utterances whose durations are to be PREDICTED (durations = noise = NULL), without the host round trip
 * prepare_batch makes between the duration predictor and the rest (it needs the frame count to pick the plan): the caller states an upper
 * bound on the frames per item, the plan is the bucket of `max_frames`, and the frame counts stay on the device — generate_path
 * (the exported graph's CumSum / Less chain, GraphExecutor.swift:1283-1330 CumSum) runs there too and every kernel masks by its result, so
 * the waveform is what prepare_batch would have produced. Nothing waits for the GPU here. Until `collect`,
 * piper_hip_voice_prepared_samples reports the CAPACITY (n · bucket(max_frames) · hop) — the room `collect` needs — and afterwards the
 * true lengths; piper_hip_voice_durations answers after `collect`. An item whose prediction exceeds `max_frames` makes `collect` fail
 * with PIPER_HIP_ERR_SHAPE (prepare it again with a larger bound or through prepare_batch). */
int piper_hip_voice_prepare_batch_bounded(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int slot, int max_frames);
/* Plans (schedule + arena + HIP graph) are cached per voice by bucket — phonemes rounded up to 16, frames to 16 (64 beyond
 * 1024) — and batch size, least recently used first out, so a (T, F) never seen before usually finds its graph ready and
 * `prepare` is only the input upload ("warm"); a new bucket pays schedule construction + one validation pass + capture +
 * instantiate once ("cold"). Reports the bucket of a prepared slot and the cache's size. */
int piper_hip_voice_plan_info(const piper_hip_voice* v, int slot, int32_t* bucket_t, int32_t* bucket_f, int32_t* cached_plans,
                              size_t* cached_bytes);
/* Bounds of the voice's plan cache: at most `max_plans` plans and `max_bytes` of arenas (0 = the default 24 GiB) are kept; idle plans are
 * evicted least recently used first, plans attached to a slot never. Defaults: 128 plans. */
int piper_hip_voice_set_plan_cache(piper_hip_voice* v, int max_plans, size_t max_bytes);
/* Wall milliseconds of the phases of the LATEST plan build (a "cold" prepare): [0] stream / events, [1] schedule construction + arena
 * allocation, [2] arena initialisation, [3] eager validation pass, [4] graph capture, [5] graph instantiate. */
int piper_hip_voice_last_build_breakdown(const piper_hip_voice* v, double out_ms[6]);
/* Batch size of a prepared slot (0 if the slot is not prepared). */
int piper_hip_voice_batch_size(const piper_hip_voice* v, int slot);
/* Enqueue the prepared slot's forward pass (one hipGraphLaunch). No host sync. */
int piper_hip_voice_launch(piper_hip_voice* v, int slot);
/* Wait for the slot and copy the waveform(s) [batch · num_samples] to host (NULL = just wait). */
int piper_hip_voice_collect(piper_hip_voice* v, int slot, float* host_audio, int64_t max_samples);
/* The duration predictor alone (text encoder + `dp` of the graph): frames per id for `n` utterances (their `durations` fields are
 * ignored), items back to back in `durations_out` (Σ T_b entries); `logw_out` (optional, same length) receives the predictor's
 * log-durations before exp / length_scale / ceil. This is what `prepare` runs first when an utterance has durations == NULL. */
int piper_hip_voice_predict_durations(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int32_t* durations_out, float* logw_out,
                                      int max_entries);
/* Streaming ⇔ PiperMetalRuntime.synthesizeStream (PiperMetalRuntime.swift:82-115) — which "today chunks the final
 * waveform". Here the text encoder and the flow run once (stream_begin) and the HiFi-GAN generator decodes the latent
 * window by window: each stream_next decodes `chunk_frames` frames plus the generator's receptive field on both sides
 * (piper_hip_voice_receptive_field frames; clamped at the utterance's ends) and returns chunk_frames · hop samples, so the
 * first audio is available after encoder + flow + one window instead of the whole utterance. The samples equal
 * synthesize()'s up to fp32 summation order (tile splits depend on the window length). Batch 1.
 * stream_begin returns the number of chunks (≥ 1) or a negative status; stream_next sets *n_samples = 0 at the end. */
int piper_hip_voice_receptive_field(const piper_hip_voice* v);
int piper_hip_voice_stream_begin(piper_hip_voice* v, const piper_hip_utterance* u, int slot, int chunk_frames);
int piper_hip_voice_stream_next(piper_hip_voice* v, int slot, float* host_audio, int64_t max_samples, int64_t* n_samples);
/* prepare + launch + collect: PiperMetalRuntime.synthesize (PiperMetalRuntime.swift:62-80). */
int piper_hip_voice_synthesize(piper_hip_voice* v, const piper_hip_utterance* u, float* host_audio,
                               int64_t max_samples, int64_t* n_samples);
/* Debug taps ⇔ GraphExecutor.execute(maxNodeIndex:) returning intermediates (GraphExecutor.swift:75-152):
 * copy a named intermediate of the slot's last run to host. Names: "enc_out" [H,T], "m_p" [inter,T],
 * "logs_p" [inter,T], "z_p" [inter,F], "z" [inter,F], "dec_pre" [up_initial,F] — per batch item, compacted to the item's
 * true T / F (the bucket's padding is not copied), items back to back. With predicted durations, "logw" [1,T] of the
 * predictor is available through piper_hip_voice_predict_durations. */
int piper_hip_voice_tap(piper_hip_voice* v, int slot, const char* name, float* host, size_t max_floats,
                        size_t* n_floats);
/* GPU milliseconds of the slot's last completed launch (hipEvent pair on the slot's stream) ⇔
 * RunTimings.gpuMs (GraphExecutor.swift:29-38). */
int piper_hip_voice_last_gpu_ms(piper_hip_voice* v, int slot, double* ms);
/* Stream of a slot (for external event timing / profiling). */
piper_hip_stream piper_hip_voice_slot_stream(piper_hip_voice* v, int slot);
/* Per-kernel timing of the slot's schedule: runs the schedule eagerly `iters` times with a hipEvent pair
 * around every launch and reports the median per launch; fills up to `max_entries` entries in schedule order. */
typedef struct {
  char name[48];
  double avg_us;     /* MEDIAN launch duration over the profiled passes (robust against a stalled pass) */
  double flops;      /* algorithmic FLOPs of this launch (SURVEY.md Appendix A recipe) */
  double bytes;      /* algorithmic bytes of this launch (same recipe) */
} piper_hip_kernel_stat;
int piper_hip_voice_profile(piper_hip_voice* v, int slot, int iters, piper_hip_kernel_stat* out, int max_entries,
                            int* n_entries);

/* Time a subset of the slot's schedule the way it runs in production: the launches whose name contains `name_filter`
 * ("" = all; "a|b|c" = the launches named exactly a, b or c) are captured into their own HIP graph and replayed `iters` times between two hipEvents on the slot's
 * stream. avg_launch_us = elapsed / (iters · n_launches) — kernel time plus the dispatch boundary, no per-kernel event
 * overhead; flops/bytes are the algorithmic totals of the subset (SURVEY.md Appendix A recipe). */
int piper_hip_voice_time_subset(piper_hip_voice* v, int slot, const char* name_filter, int iters, double* avg_launch_us,
                                int* n_launches, double* flops, double* bytes);

/* ---- multi-GPU: the path's one collective --------------------------------------------------------------------
 * Utterances are independent, so N GPUs are N replicas of the voice, one process per GPU; the only exchange is the one-shot
 * broadcast of the voice blob from the rank that parsed the .onnx (PiperMetalRuntime.init(modelPath:) does that parse once per
 * process, PiperMetalRuntime.swift:37-41 → ONNXModel.init). These wrap rccl.h so that a host without torch.distributed (the Swift CLI, a C++
 * server) can do it: rank 0 calls comm_unique_id and hands the 128 bytes to the other ranks by any side channel (a file, a
 * socket, an environment variable), every rank calls comm_create (collective), uploads / allocates n_floats on its GPU,
 * calls comm_broadcast_f32 (collective, in place, returns when the data is there) and piper_hip_voice_create(on_device = 1).
 * comm_max_f64 / comm_barrier are what a bench needs for the "MAX over ranks between two barriers" timing rule.
 * RCCL is bound at run time: without librccl.so.1 these return PIPER_HIP_ERR_UNAVAILABLE and the rest of the library works. */
#define PIPER_HIP_COMM_ID_BYTES 128
int piper_hip_comm_unique_id(void* id_out /* PIPER_HIP_COMM_ID_BYTES */);
int piper_hip_comm_create(piper_hip_ctx* ctx, const void* id, int rank, int world, piper_hip_comm** out);
void piper_hip_comm_destroy(piper_hip_comm* c);
int piper_hip_comm_rank(const piper_hip_comm* c);
int piper_hip_comm_world(const piper_hip_comm* c); /* ncclCommCount: what the communicator says */
int piper_hip_comm_broadcast_f32(piper_hip_comm* c, float* device_buf, size_t count, int root);
int piper_hip_comm_max_f64(piper_hip_comm* c, double* value); /* in/out: all-reduce MAX of one double */
int piper_hip_comm_barrier(piper_hip_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* PIPER_HIP_H */
