// voice.hip — whole-utterance VITS forward pass as a static schedule of fused launches, replayed as a HIP graph.
//
// Stands where PiperMetalRuntime.synthesize → GraphExecutor.executeOutput stands in the reference
// (PiperMetalRuntime.swift:62-80, GraphExecutor.swift:156-327), but instead of interpreting 2 755 ONNX nodes with a
// fresh buffer, a string-keyed table lookup and (in the unbatched mode) a blocking commit per node, the voice is
// compiled once into ≈110 launches over a preplanned arena:
//   encoder layer = qkv conv · rel-attention · o conv · add+LayerNorm · ffn1(+ReLU) · ffn2 · add+LayerNorm
//   flow coupling = pre conv (Flip/Split folded into channel maps) · 4×[in conv + tanh·sigmoid gate, res/skip conv
//                   writing x and skip in place] · post conv with x1 ← x1 − m in place (Concat folded)
//   generator     = conv_pre · per stage [LeakyReLU(+MRF mean)→ConvTranspose, 3 ResBlocks with LeakyReLU and residual
//                   fused into each conv] · LeakyReLU+MRF mean→conv_post→tanh
// Weights stay resident (packed once into MFMA fragment order); nothing is decoded or uploaded per call
// (the reference re-decodes 401 initializers and re-uploads every Conv weight per synthesize: GraphExecutor.swift:187-189,
// 1774-1780).  Utterances are independent, so a voice has several slots (stream + arena + graph) that overlap on the GPU.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <functional>
#include <memory>

#include "../../include/piper_hip_voice_layout.h"
#include "conv.h"
#include "conv_bf16.h"
#include "conv_win.h"
#include "rng.h"

namespace ph {
int validate_config(const piper_hip_voice_config* c);
int launch_rel_attention(piper_hip_ctx* ctx, hipStream_t s, const float* q, const float* k, const float* v, const float* ek,
                         const float* ev, float* out, int N, int H, int d, int T, int w, int64_t in_batch_stride,
                         int64_t out_batch_stride, const int* len_ptr);
int rel_attention_split_parts(piper_hip_ctx* ctx, int N, int H, int d, int T, int w);
int launch_rel_attention_split(piper_hip_ctx* ctx, hipStream_t s, const float* q, const float* k, const float* v, const float* ek,
                               const float* ev, float* out, int N, int H, int d, int T, int w, int64_t in_batch_stride,
                               int64_t out_batch_stride, const int* len_ptr, int nsplit, float* part_o, float* part_ml);
size_t dp_scalars_bytes(int n);
void dp_scalars_fill(void* host, int i, float noise_w, float length_scale, unsigned gen, unsigned seed);
bool flow_seam_eligible(int H, int half);
int launch_flow_seam(hipStream_t s, const float* skip, float* zp, float* h, const float* post16, const float* post_b, const float* pre16,
                     const float* pre_b, int N, int H, int half, int F, int post_steps, int pre_steps, int ob, int os, const int* len_ptr);
bool dds_layer_eligible(int H, int K);
int launch_dds_layer(piper_hip_ctx* ctx, hipStream_t s, const float* x, const float* dw_w, const float* dw_b, const float* g1, const float* b1,
                     const float* pw16, const float* pw_b, const float* g2, const float* b2, float* out, int N, int H, int T, int K, int dil,
                     int pw_steps, const int* len_ptr, float eps);
int launch_dp_init(hipStream_t s, const float* noise, const void* scalars, float* z, int N, int T, const int* len_ptr);
int launch_dp_spline(hipStream_t s, const float* h, float* z, int N, int T, int bins, float tail_bound, float filter_channels, const int* len_ptr);
int launch_dp_final(hipStream_t s, const float* z, const float* m, const float* logs, const void* scalars, float* logw, int32_t* dur, int N, int T,
                    const int* len_ptr);
bool attention_block_eligible(int H, int d, int w, int T);
bool attention_block_wanted();
int launch_attention_block(piper_hip_ctx* ctx, hipStream_t s, const float* q, const float* k, const float* v, const float* ek, const float* ev,
                           const float* wo16, const float* bo, const float* xres, const float* gamma, const float* beta, float* out, int N,
                           int H, int d, int T, int w, int64_t in_batch_stride, int64_t x_batch_stride, const int* len_ptr, int o_nsteps,
                           float eps);
}  // namespace ph

using namespace ph;

namespace {

constexpr int kBlock = 256;
constexpr int kMaxSlots = 16;

// x[c][t] = emb[ids[t]][c] * sqrt(H): Gather + Mul + Transpose of the graph head (GraphExecutor.swift:653-666)
__global__ __launch_bounds__(kBlock) void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb,
                                                       float* __restrict__ x, int H, int T, int n_vocab, float scale) {
  ids += (int64_t)blockIdx.y * T;  // batch item
  x += (int64_t)blockIdx.y * H * T;
  const int total = H * T;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const int c = i / T, t = i - c * T;
    int64_t id = ids[t];
    if (id < 0) id += n_vocab;  // gather_axis0_f32_2d (gather.metal:54-57): negative ids wrap once, …
    x[i] = (id < 0 || id >= n_vocab) ? 0.0f * scale : emb[id * H + c] * scale;  // … what is still out of range gathers 0.0
  }
}

// z_p[c][f] = m_p[c][t(f)] + (noise[c][f] * exp(logs_p[c][t(f)])) * noise_scale.
// m_p/logs_p expansion by the one-hot path matrix (MatMul [1,F,T]×[1,T,I], GraphExecutor.swift:1862-1915) is a row
// gather: frame f copies phoneme t(f); bit-identical to the matmul (every other product is an exact ±0 add).
__global__ __launch_bounds__(kBlock) void expand_noise_kernel(const float* __restrict__ stats, const int32_t* __restrict__ frame2id,
                                                              const float* __restrict__ noise, float* __restrict__ zp, float* __restrict__ zp_tap,
                                                              int I, int T, int F, const float* __restrict__ noise_scale_dev,
                                                              const unsigned* __restrict__ rng_dev, const int* __restrict__ lensF) {
  const int nb = blockIdx.y;  // batch item
  stats += (int64_t)nb * 2 * I * T;
  frame2id += (int64_t)nb * F;
  noise += (int64_t)nb * I * F;
  zp += (int64_t)nb * I * F;
  zp_tap += (int64_t)nb * I * F;
  const float noise_scale = noise_scale_dev[nb];  // per-utterance scalar lives in device memory so a replayed graph sees it
  // RandomNormalLike on the device (random_normal_like_f32, elementwise.metal:139-163): element index = flat index of the
  // item's [1, I, F] tensor; rng_dev[2nb] = generate?, rng_dev[2nb+1] = seed. Otherwise the injected tensor is read.
  const bool gen = rng_dev[2 * nb] != 0u;
  const unsigned seed = rng_dev[2 * nb + 1];
  const int Fv = lensF ? min(lensF[nb], F) : F;  // RandomNormalLike mirrors the item's TRUE [1, I, Fv] shape, not the bucket
  const int64_t total = (int64_t)I * F;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(i / F), f = (int)(i - (int64_t)c * F);
    const int t = frame2id[f];
    const float m = stats[(int64_t)c * T + t];
    const float lg = stats[(int64_t)(I + c) * T + t];
    const float nz = gen ? (f < Fv ? rnl_normal(seed, (unsigned)(c * Fv + f)) : 0.0f) : noise[i];
    const float r = m + (nz * expf(lg)) * noise_scale;
    zp[i] = r;      // updated in place by the flow couplings
    zp_tap[i] = r;  // pristine copy for the "z_p" debug tap
  }
}

// HiFi-GAN multi-receptive-field mean + the LeakyReLU that follows it: y = lrelu(((a + b) + c) / 3, alpha).
// Done once per element here instead of inside the consumer conv, which re-reads every input element
// (row tiles × taps) times — the in-conv form paid 3 loads and one IEEE division per re-read.
__global__ __launch_bounds__(kBlock) void mrf_mean_lrelu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                const float* __restrict__ c, float* __restrict__ y, int64_t n,
                                                                float alpha) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    const float4 va = reinterpret_cast<const float4*>(a)[i], vb = reinterpret_cast<const float4*>(b)[i],
                 vc = reinterpret_cast<const float4*>(c)[i];
    float4 r;
    r.x = ((va.x + vb.x) + vc.x) / 3.0f; r.y = ((va.y + vb.y) + vc.y) / 3.0f;
    r.z = ((va.z + vb.z) + vc.z) / 3.0f; r.w = ((va.w + vb.w) + vc.w) / 3.0f;
    r.x = r.x >= 0.0f ? r.x : alpha * r.x; r.y = r.y >= 0.0f ? r.y : alpha * r.y;
    r.z = r.z >= 0.0f ? r.z : alpha * r.z; r.w = r.w >= 0.0f ? r.w : alpha * r.w;
    reinterpret_cast<float4*>(y)[i] = r;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float v = ((a[i] + b[i]) + c[i]) / 3.0f;
    y[i] = v >= 0.0f ? v : alpha * v;
  }
}

__global__ __launch_bounds__(kBlock) void flip_channels_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int L) {
  x += (int64_t)blockIdx.y * C * L;  // batch item
  y += (int64_t)blockIdx.y * C * L;
  const int64_t total = (int64_t)C * L;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(i / L), l = (int)(i - (int64_t)c * L);
    y[i] = x[(int64_t)(C - 1 - c) * L + l];
  }
}

// keeps the GPU busy for `ticks` of the 100 MHz realtime counter: lets the host queue a whole profiled pass ahead of the
// GPU, so the per-launch event deltas contain the ~1.7 µs kernel boundary but not host launch latency
__global__ void empty_kernel() {}
// lensT / lensF of a fresh plan = the bucket's own lengths (a kernel, not a copy from a host vector: see stream_wait below — nothing on the
// request path hands PAGEABLE host memory to an asynchronous copy)
__global__ void fill_lens_kernel(int* __restrict__ lensT, int* __restrict__ lensF, int T, int F, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { lensT[i] = T; lensF[i] = F; }
}

__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// Wait for a stream on the request path. hipStreamSynchronize parks the thread on an interrupt; on the GPU boxes of this pool a wake-up is
// now and then 25 … 35 ms late (r3, tools/probe/request_max.py: the same ~26 ms in whichever synchronisation of a request it hits — arena
// initialisation, prepare, collect — on requests whose GPU work is 1 ms). A request is short: poll the stream for up to 5 ms, then park.
hipError_t stream_wait(hipStream_t q) {
  static const bool park = getenv("PIPER_HIP_NO_SPIN_WAIT") != nullptr;
  if (!park) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(q);
      if (e != hipErrorNotReady) return e;
      (void)hipGetLastError();  // "not ready" must not show up as the sticky error of a later launch check
      if (clk::now() - t0 > std::chrono::milliseconds(5)) break;
      for (int i = 0; i < 16; i++) __builtin_ia32_pause();
    }
  }
  return hipStreamSynchronize(q);
}

struct Step {
  std::string name;
  std::function<int(hipStream_t)> run;
  double flops = 0, bytes = 0;
  std::string tag;  // kernel family ("conv_mfma", "conv_small", "rel_attention", …) for piper_hip_voice_time_subset
  int lane = 0;  // 0 = the slot's stream; 1, 2 = side streams between a FORK and a JOIN (independent ResBlocks of one stage)
  enum Kind { LAUNCH, FORK, JOIN } kind = LAUNCH;
};

struct ConvW {  // one resident conv: packed (MFMA) or raw (direct) weights + bias pointer into the resident blob
  const float* w = nullptr;
  const float* w16 = nullptr;  // 16-wide fragment image (short-utterance geometry)
  const float* w16g = nullptr; // the same for a gated conv: 8 tanh rows + their 8 sigmoid rows per tile
  const float* w8 = nullptr;   // 8-row fragment image (the FFN's second conv: 768 → 192)
  const float* w4 = nullptr;   // conv_win_kernel fragment image (generator convs: long rows)
  const float* w5 = nullptr;   // conv_pipe_kernel fragment image (chunk-major step order; Cin % 32 == 0)
  const float* bias = nullptr;
  int Cout = 0, Cin = 0, K = 1;
  bool mfma = false;
};

struct Slot {
  bool inited = false;
  hipStream_t stream = nullptr;
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // A Slot is a PLAN: schedule + arena + graph for one bucket (kind, T, F, NB). T and F are the bucket's row lengths; the
  // true lengths of the NB batch items live in device memory (lensT / lensF) where every length-aware kernel reads them, so
  // one captured graph serves every utterance that fits the bucket — exactly (positions past a true length read as zero
  // padding, attention excludes keys past it), not approximately.
  int kind = 0;  // 0 = whole utterance, 1 = generator only (streaming window)
  int T = -1, F = -1, NB = 1;
  int prec = 0;            // generator precision the schedule was built for
  bool in_use = false;     // attached to a user slot id
  bool built = false;      // schedule + arena exist; `exec` (the captured graph) follows after the plan's FIRST run, which goes out eagerly
  uint64_t last_use = 0;   // voice-wide clock value of the last attach (LRU)
  size_t arena_bytes = 0;
  int* lensT = nullptr;    // [NB] device: true phoneme count per item
  int* lensF = nullptr;    // [NB] device: true frame count per item
  std::vector<int> h_T, h_F;   // the same on the host (collect / tap / streaming)
  // duration-predictor plan (kind 2): encoder + predictor → frames per id
  float* dp_noise = nullptr;   // [NB][2][T] injected `dp` RandomNormalLike tensor
  void* dp_scalars = nullptr;  // [NB] DpScalars (device)
  int32_t* dp_dur = nullptr;   // [NB][T] predicted frames per id
  std::vector<int32_t> h_dur;  // predicted durations of the attached request, per item back to back (host)
  int* h_lens = nullptr;       // pinned staging [2·NB]
  // kind 3 prepared by piper_hip_voice_prepare_batch_bounded: the frame counts are decided on the device and reach the host with the
  // waveform (collect). Until then h_F holds the bucket's capacity.
  bool bounded_pending = false;
  int bounded_cap = 0;         // max_frames the caller allowed
  hipEvent_t ev_in = nullptr;  // kind 3: "the copy of the predictor plan's projection has been read" (that plan's stream waits for it)
  float* stats = nullptr;      // [NB][2·inter][T] encoder projection (m_p ; logs_p): output of kinds 0 / 2, INPUT of kind 3
  float* h_audio = nullptr;    // pinned landing buffer of the waveform (collect: device → pinned DMA, then a host memcpy)
  size_t h_audio_cap = 0;
  size_t h_cap_lens = 0;
  // device buffers
  std::vector<void*> owned;
  int64_t* ids = nullptr;
  int32_t* frame2id = nullptr;
  float* noise = nullptr;
  float* noise_scale = nullptr;  // [NB] device
  unsigned* rng = nullptr;       // [NB][2] device: {generate noise on the device?, seed}
  std::vector<float> h_noise_scale;
  std::vector<unsigned> h_rng;
  float* audio = nullptr;
  int64_t n_samples = 0;
  std::vector<Step> steps;
  struct Tap {  // a named intermediate: [NB] items of [C][row] floats, of which the first len_b (T or F of item b) are real
    const float* p;
    int C, row, unit;  // unit: 0 = phonemes (T), 1 = frames (F)
    size_t batch_stride;
  };
  std::map<std::string, Tap> taps;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool timed = false;
  bool parallel = false;  // capture FORK/JOIN lanes as parallel graph branches (set by the schedule builder)
  float* zin = nullptr;   // generator-only schedule (streaming): the latent window [inter, F] it decodes
  const float* z_out = nullptr;  // full schedule: where the flow leaves z
  // streaming state of a full-schedule slot (piper_hip_voice_stream_*)
  hipGraph_t front_graph = nullptr;
  hipGraphExec_t front_exec = nullptr;  // encoder + flow only
  int st_chunk = 0, st_next = -1, st_halo = 0;
  int cur_lane = 0;  // lane given to steps added by add_conv
  // host staging (pinned so the H2D copies are truly async)
  int64_t* h_ids = nullptr;
  int32_t* h_f2i = nullptr;
  size_t h_cap_t = 0, h_cap_f = 0;
};

}  // namespace

struct piper_hip_voice {
  piper_hip_ctx* ctx = nullptr;
  piper_hip_voice_config cfg{};
  float* blob = nullptr;    // resident copy of the fp32 blob (biases, embeddings, LayerNorm, raw weights)
  float* packed = nullptr;  // MFMA fragment images
  size_t blob_floats = 0, packed_floats = 0;
  std::map<std::string, piper_tensor_desc> index;
  // compiled weights
  struct EncLayer {
    ConvW qkv, o, f1, f2;
    const float *ek, *ev, *g1, *b1, *g2, *b2;
    float* qkv_bias;
  };
  std::vector<EncLayer> enc;
  ConvW proj;
  struct Coupling {
    ConvW pre, post;
    std::vector<ConvW> in, rs;
  };
  std::vector<Coupling> flows;
  ConvW conv_pre, conv_post;
  struct DdsLayer {
    const float *dw_w, *dw_b, *g1, *b1, *g2, *b2;
    ConvW pw;  // 1×1 conv: the 16-wide fragment image (w16) is what dds_layer_kernel reads
  };
  struct DpBlock {  // the predictor's own stack (flow = 0) or one ConvFlow
    int flow = 0;
    ConvW pre, proj;
    std::vector<DdsLayer> dds;
  };
  std::vector<DpBlock> dp;  // [0] main, then the ConvFlows in EXECUTION order (module indices 7, 5, 3)
  const float *dp_m = nullptr, *dp_logs = nullptr;
  struct Stage {
    ConvW up;  // packed ConvTranspose (rows = Cout·stride)
    int Cin, Cout, K, stride, pad;
    std::vector<std::vector<ConvW>> rb;  // [n_rb][n_dil or 2·n_dil]
  };
  std::vector<Stage> stages;
  // bf16 generator (PIPER_HIP_PRECISION_BF16): fragment images of the same decoder convs, built by set_precision
  struct ConvWB {
    const uint16_t* w = nullptr;
    const float* bias = nullptr;
    int Cout = 0, Cin = 0, K = 0;
  };
  int precision = PIPER_HIP_PRECISION_F32;
  ConvWB conv_pre_b;
  std::vector<ConvWB> up_b;                             // [stage]
  std::vector<std::vector<std::vector<ConvWB>>> rb_b;   // [stage][rb][conv]
  std::vector<void*> owned;
  // Plans are cached voice-wide, least-recently-used first out; a user slot id is a handle on one of them. A TTS server sees
  // a new (T, F) almost every call: with buckets the plan for it usually exists already (prepare = input upload only).
  std::vector<std::unique_ptr<Slot>> plans;
  struct StreamSet { hipStream_t stream, side[2]; hipEvent_t ev0, ev1, ev_fork, ev_join[2]; };
  std::vector<StreamSet> free_sets;  // streams / events of evicted plans, reused by the next build (a HIP stream costs ≈ 3 ms to create)
  double last_build_ms[6] = {0, 0, 0, 0, 0, 0};  // the latest plan build: stream/events, schedule + arena, arena init, eager pass, capture, instantiate
  size_t plan_cache_max = 128, plan_cache_bytes = (size_t)24 << 30;
  // Page-locked staging of a slot id's inputs and of its (short) waveform: it belongs to the SLOT ID, not to the plan attached to it —
  // a new bucket then costs no hipHostMalloc (≈ 0.3–1 ms each, four per plan until round 3). Grown in powers of two, freed with the voice.
  struct Staging {
    int64_t* h_ids = nullptr; int32_t* h_f2i = nullptr; int* h_lens = nullptr; float* h_audio = nullptr;
    size_t cap_t = 0, cap_f = 0, cap_lens = 0, audio_cap = 0;
    // bounded prepare (durations predicted, no host round trip): scalars of the request, the predictor's noise, and what the device reports
    // back (frames per item, durations) — the last two written by a kernel through the host mapping
    char* h_misc = nullptr; float* h_dpn = nullptr; int32_t* h_res = nullptr;
    size_t cap_misc = 0, cap_dpn = 0, cap_res = 0;
    float* h_noise = nullptr;  // the caller's noise tensors, laid out in bucket rows (prepare_batch)
    size_t cap_noise = 0;
    std::vector<hipEvent_t> chunk_ev;  // collect, 1 … 16 MB waveforms: one event per 1 MB chunk landed in h_audio
  } staging[kMaxSlots];
  Slot* attached[kMaxSlots] = {};
  Slot* attached_dp[kMaxSlots] = {};  // bounded prepare: the encoder + predictor plan this slot id holds until its next prepare / detach
  uint64_t use_clock = 0;
  int hop = 1;
};

namespace {

struct IndexOut {
  std::map<std::string, piper_tensor_desc>* m;
};
void index_visit(const piper_tensor_desc* d, void* user) { (*((IndexOut*)user)->m)[d->name] = *d; }

const float* tensor(const piper_hip_voice* v, const std::string& name) {
  auto it = v->index.find(name);
  if (it == v->index.end()) return nullptr;
  return v->blob + it->second.offset;
}

struct Packer {  // bump allocator over the packed-weights allocation
  piper_hip_voice* v;
  size_t off = 0;
  hipStream_t s;
  float* take(size_t n) {
    float* p = v->packed ? v->packed + off : nullptr;
    off += (n + 63) & ~(size_t)63;
    return p;
  }
};

// Registers one conv. dry = true only measures the packed size.
ConvW make_conv(Packer& pk, bool dry, const std::string& prefix, int Cout, int Cin, int K, bool has_bias = true, bool win = false, bool gated = false, bool rows8 = false) {
  ConvW c;
  c.Cout = Cout; c.Cin = Cin; c.K = K;
  c.mfma = conv_mfma_eligible(Cout, Cin, K, 1, 1);
  const float* w = dry ? nullptr : tensor(pk.v, prefix + ".weight");
  c.bias = (dry || !has_bias) ? nullptr : tensor(pk.v, prefix + ".bias");
  if (c.mfma) {
    float* p = pk.take(packed_conv_floats(Cout, Cin, K));
    if (!dry) pack_conv_weights(pk.s, w, Cout, Cin, K, p);
    c.w = p;
    float* p16 = pk.take(packed_conv_floats(Cout, Cin, K, 16));
    if (!dry) pack_conv_weights(pk.s, w, Cout, Cin, K, p16, 16);
    c.w16 = p16;
    if (rows8 && Cin % 32 == 0) {  // 8-row tiles for a conv with many input channels and few output rows (conv_lean.hip)
      float* p8 = pk.take(packed_conv_rows8_floats(Cout, Cin, K));
      if (!dry) pack_conv_weights_rows8(pk.s, w, Cout, Cin, K, p8);
      c.w8 = p8;
    }
    if (gated && Cout % 16 == 0) {  // tanh / sigmoid rows interleaved per 16-row tile (conv_short.hip)
      float* pg = pk.take(packed_conv_floats(Cout, Cin, K, 16));
      if (!dry) pack_conv_weights_gate16(pk.s, w, Cout, Cin, K, pg);
      c.w16g = pg;
    }
    if (win) {
      float* p4 = pk.take(packed_conv_win_floats(Cout, Cin, K));
      if (!dry) pack_conv_weights_win(pk.s, w, Cout, Cin, K, p4);
      c.w4 = p4;
      if (Cin % 32 == 0 && K * (Cin / 32) >= 2) {
        float* p5 = pk.take(packed_conv_pipe_floats(Cout, Cin, K));
        if (!dry) pack_conv_weights_pipe(pk.s, w, Cout, Cin, K, p5);
        c.w5 = p5;
      }
    }
  } else {
    c.w = w;
  }
  return c;
}

int compile_weights(piper_hip_voice* v, Packer& pk, bool dry, const std::vector<float*>* qkv_bias) {
  const piper_hip_voice_config& c = v->cfg;
  const int H = c.hidden, I = c.inter;
  char nm[128];
  v->enc.clear(); v->flows.clear(); v->stages.clear();
  for (int l = 0; l < c.n_layers; l++) {
    piper_hip_voice::EncLayer L{};
    L.qkv_bias = qkv_bias ? (*qkv_bias)[l] : nullptr;
    // q,k,v as one 3H-row conv: the packed image is [row tile][step][64], so three H-row images concatenate
    L.qkv.Cout = 3 * H; L.qkv.Cin = H; L.qkv.K = 1; L.qkv.mfma = true;
    float* p = pk.take(3 * packed_conv_floats(H, H, 1));
    float* p16 = pk.take(3 * packed_conv_floats(H, H, 1, 16));
    L.qkv.w = p;
    L.qkv.w16 = p16;
    if (!dry) {
      static const char* qkv[3] = {"conv_q", "conv_k", "conv_v"};
      for (int j = 0; j < 3; j++) {
        snprintf(nm, sizeof nm, "enc_p.encoder.attn_layers.%d.%s", l, qkv[j]);
        pack_conv_weights(pk.s, tensor(v, std::string(nm) + ".weight"), H, H, 1, p + j * packed_conv_floats(H, H, 1));
        pack_conv_weights(pk.s, tensor(v, std::string(nm) + ".weight"), H, H, 1, p16 + j * packed_conv_floats(H, H, 1, 16), 16);
        PH_HIP(hipMemcpyAsync(L.qkv_bias + j * H, tensor(v, std::string(nm) + ".bias"), H * sizeof(float),
                              hipMemcpyDeviceToDevice, pk.s), PIPER_HIP_ERR_LAUNCH);
      }
      L.qkv.bias = L.qkv_bias;
    }
    snprintf(nm, sizeof nm, "enc_p.encoder.attn_layers.%d.conv_o", l);
    L.o = make_conv(pk, dry, nm, H, H, 1);
    snprintf(nm, sizeof nm, "enc_p.encoder.ffn_layers.%d.conv_1", l);
    L.f1 = make_conv(pk, dry, nm, c.ffn, H, c.ffn_kernel);
    snprintf(nm, sizeof nm, "enc_p.encoder.ffn_layers.%d.conv_2", l);
    L.f2 = make_conv(pk, dry, nm, H, c.ffn, c.ffn_kernel, true, false, false, true);
    if (!dry) {
      snprintf(nm, sizeof nm, "enc_p.encoder.attn_layers.%d.emb_rel_k", l); L.ek = tensor(v, nm);
      snprintf(nm, sizeof nm, "enc_p.encoder.attn_layers.%d.emb_rel_v", l); L.ev = tensor(v, nm);
      snprintf(nm, sizeof nm, "enc_p.encoder.norm_layers_1.%d.gamma", l); L.g1 = tensor(v, nm);
      snprintf(nm, sizeof nm, "enc_p.encoder.norm_layers_1.%d.beta", l); L.b1 = tensor(v, nm);
      snprintf(nm, sizeof nm, "enc_p.encoder.norm_layers_2.%d.gamma", l); L.g2 = tensor(v, nm);
      snprintf(nm, sizeof nm, "enc_p.encoder.norm_layers_2.%d.beta", l); L.b2 = tensor(v, nm);
    }
    v->enc.push_back(L);
  }
  v->proj = make_conv(pk, dry, "enc_p.proj", 2 * I, H, 1);
  for (int f = 0; f < c.n_flows; f++) {
    piper_hip_voice::Coupling C;
    snprintf(nm, sizeof nm, "flow.flows.%d.pre", 2 * f);
    C.pre = make_conv(pk, dry, nm, H, I / 2, 1);
    for (int i = 0; i < c.wn_layers; i++) {
      snprintf(nm, sizeof nm, "flow.flows.%d.enc.in_layers.%d", 2 * f, i);
      C.in.push_back(make_conv(pk, dry, nm, 2 * H, H, c.wn_kernel, true, false, true));
      snprintf(nm, sizeof nm, "flow.flows.%d.enc.res_skip_layers.%d", 2 * f, i);
      C.rs.push_back(make_conv(pk, dry, nm, (i + 1 < c.wn_layers) ? 2 * H : H, H, 1));
    }
    snprintf(nm, sizeof nm, "flow.flows.%d.post", 2 * f);
    C.post = make_conv(pk, dry, nm, I / 2, H, 1);
    v->flows.push_back(C);
  }
  v->conv_pre = make_conv(pk, dry, "dec.conv_pre", c.up_initial, I, 7);
  int ch = c.up_initial;
  for (int u = 0; u < c.n_ups; u++) {
    piper_hip_voice::Stage S;
    S.Cin = ch; S.Cout = ch / 2; S.K = c.up_kernels[u]; S.stride = c.up_rates[u]; S.pad = (S.K - S.stride) / 2;
    const int J = (S.K + S.stride - 1) / S.stride;
    S.up.Cout = S.Cout * S.stride; S.up.Cin = S.Cin; S.up.K = J; S.up.mfma = true;
    float* p = pk.take(packed_convt_floats(S.Cin, S.Cout, S.K, S.stride));
    float* p16 = pk.take(packed_convt_floats(S.Cin, S.Cout, S.K, S.stride, 16));
    S.up.w = p;
    S.up.w16 = p16;
    const bool ct_win = convt_win_eligible(S.Cin, S.Cout, S.K, S.stride, S.pad, 4);
    float* p4 = ct_win ? pk.take(packed_convt_win_floats(S.Cin, S.Cout, S.K, S.stride)) : nullptr;
    S.up.w4 = ct_win ? p4 : nullptr;
    const bool ct_pipe = convt_pipe_eligible(S.Cin, S.Cout, S.K, S.stride, S.pad, 4);
    float* p5 = ct_pipe ? pk.take(packed_convt_pipe_floats(S.Cin, S.Cout, S.K, S.stride)) : nullptr;
    S.up.w5 = p5;
    if (!dry) {
      snprintf(nm, sizeof nm, "dec.ups.%d", u);
      if (ct_win) pack_convt_weights_win(pk.s, tensor(v, std::string(nm) + ".weight"), S.Cin, S.Cout, S.K, S.stride, S.pad, p4);
      if (ct_pipe) pack_convt_weights_pipe(pk.s, tensor(v, std::string(nm) + ".weight"), S.Cin, S.Cout, S.K, S.stride, S.pad, p5);
      pack_convt_weights(pk.s, tensor(v, std::string(nm) + ".weight"), S.Cin, S.Cout, S.K, S.stride, p);
      pack_convt_weights(pk.s, tensor(v, std::string(nm) + ".weight"), S.Cin, S.Cout, S.K, S.stride, p16, 16);
      S.up.bias = tensor(v, std::string(nm) + ".bias");
    }
    ch /= 2;
    for (int j = 0; j < c.n_rb; j++) {
      std::vector<ConvW> convs;
      const int rb = u * c.n_rb + j;
      for (int d = 0; d < c.rb_n_dil; d++) {
        if (c.resblock_type == 1) {
          snprintf(nm, sizeof nm, "dec.resblocks.%d.convs1.%d", rb, d);
          convs.push_back(make_conv(pk, dry, nm, ch, ch, c.rb_kernels[j], true, true));
          snprintf(nm, sizeof nm, "dec.resblocks.%d.convs2.%d", rb, d);
          convs.push_back(make_conv(pk, dry, nm, ch, ch, c.rb_kernels[j], true, true));
        } else {
          snprintf(nm, sizeof nm, "dec.resblocks.%d.convs.%d", rb, d);
          convs.push_back(make_conv(pk, dry, nm, ch, ch, c.rb_kernels[j], true, true));
        }
      }
      S.rb.push_back(convs);
    }
    v->stages.push_back(S);
  }
  v->conv_post = make_conv(pk, dry, "dec.conv_post", 1, ch, 7, false);
  v->dp.clear();
  if (c.dp_present) {
    auto dds_of = [&](const std::string& base) {
      std::vector<piper_hip_voice::DdsLayer> out;
      for (int i = 0; i < c.dp_dds_layers; i++) {
        piper_hip_voice::DdsLayer d{};
        const std::string sep = base + ".convs.convs_sep." + std::to_string(i), n1 = base + ".convs.norms_1." + std::to_string(i),
                          n2 = base + ".convs.norms_2." + std::to_string(i);
        d.pw = make_conv(pk, dry, base + ".convs.convs_1x1." + std::to_string(i), H, H, 1);
        if (!dry) {
          d.dw_w = tensor(v, sep + ".weight"); d.dw_b = tensor(v, sep + ".bias");
          d.g1 = tensor(v, n1 + ".gamma"); d.b1 = tensor(v, n1 + ".beta");
          d.g2 = tensor(v, n2 + ".gamma"); d.b2 = tensor(v, n2 + ".beta");
        }
        out.push_back(d);
      }
      return out;
    };
    piper_hip_voice::DpBlock m;
    m.flow = 0;
    m.pre = make_conv(pk, dry, "dp.pre", H, H, 1);
    m.proj = make_conv(pk, dry, "dp.proj", H, H, 1);
    m.dds = dds_of("dp");
    v->dp.push_back(m);
    for (int f = 2 * c.dp_n_flows - 1; f > 1; f -= 2) {
      piper_hip_voice::DpBlock b;
      const std::string base = "dp.flows." + std::to_string(f);
      b.flow = f;
      b.pre = make_conv(pk, dry, base + ".pre", H, 1, 1);             // 1 → H: direct kernel
      b.proj = make_conv(pk, dry, base + ".proj", 3 * c.dp_bins - 1, H, 1);
      b.dds = dds_of(base);
      v->dp.push_back(b);
    }
    if (!dry) { v->dp_m = tensor(v, "dp.flows.0.m"); v->dp_logs = tensor(v, "dp.flows.0.logs"); }
  }
  return PIPER_HIP_OK;
}

void slot_release(piper_hip_voice* v, Slot& s, bool all) {
  if (s.exec) { (void)hipGraphExecDestroy(s.exec); s.exec = nullptr; }
  if (s.graph) { (void)hipGraphDestroy(s.graph); s.graph = nullptr; }
  if (s.front_exec) { (void)hipGraphExecDestroy(s.front_exec); s.front_exec = nullptr; }
  if (s.front_graph) { (void)hipGraphDestroy(s.front_graph); s.front_graph = nullptr; }
  s.st_next = -1; s.zin = nullptr; s.z_out = nullptr;
  s.dp_noise = nullptr; s.dp_scalars = nullptr; s.dp_dur = nullptr;
  for (void* p : s.owned) (void)v->ctx->pool.release(p);
  s.owned.clear();
  s.steps.clear();
  s.taps.clear();
  s.T = s.F = -1;
  s.built = false;
  if (all) {  // (host buffers belong to the plan whether or not it currently holds a stream set)
    // (the pinned staging buffers are the slot id's: piper_hip_voice::staging)
    if (s.ev_in) (void)hipEventDestroy(s.ev_in);
    s.ev_in = nullptr;
    s.h_ids = nullptr; s.h_f2i = nullptr; s.h_lens = nullptr; s.h_audio = nullptr; s.h_cap_t = s.h_cap_f = s.h_cap_lens = 0; s.h_audio_cap = 0;
    // streams and events go back to the voice (the caller has synchronised them); piper_hip_voice_destroy destroys them
    if (s.stream) v->free_sets.push_back({s.stream, {s.side[0], s.side[1]}, s.ev0, s.ev1, s.ev_fork, {s.ev_join[0], s.ev_join[1]}});
    s.stream = nullptr; s.side[0] = s.side[1] = nullptr; s.ev0 = s.ev1 = s.ev_fork = nullptr; s.ev_join[0] = s.ev_join[1] = nullptr;
    s.inited = false;
  }
}

struct Arena {
  piper_hip_voice* v;
  Slot* s;
  int rc = PIPER_HIP_OK;
  float* f32(size_t n) {
    void* p = nullptr;
    if (rc) return nullptr;
    rc = v->ctx->pool.alloc((n ? n : 1) * sizeof(float), &p);
    if (!rc) { s->owned.push_back(p); s->arena_bytes += (n ? n : 1) * sizeof(float); }
    return (float*)p;
  }
  void* raw(size_t bytes) {
    void* p = nullptr;
    if (rc) return nullptr;
    rc = v->ctx->pool.alloc(bytes ? bytes : 1, &p);
    if (!rc) { s->owned.push_back(p); s->arena_bytes += bytes ? bytes : 1; }
    return p;
  }
};

double conv_bytes(int Cin, int Cout, int K, int64_t L) { return 4.0 * ((double)Cin * L + (double)Cout * L + (double)Cout * Cin * K + Cout); }

// conv step over a resident ConvW
void add_conv(piper_hip_voice* v, Slot& s, const std::string& name, const ConvW& w, ConvArgs a, int64_t Lout_for_work) {
  a.w = w.w;
  a.w16 = w.w16;
  a.w16g = w.w16g;
  a.w8 = w.w8;
  a.bias = w.bias;
  a.Cin = w.Cin; a.Cout = w.Cout; a.K = w.K;
  piper_hip_ctx* ctx = v->ctx;
  Step st;
  st.name = name;
  const bool mfma = w.mfma;
  st.run = [ctx, a, mfma](hipStream_t q) { return mfma ? launch_conv_mfma(ctx, q, a) : launch_conv_direct(ctx, q, a); };
  st.flops = a.N * conv_flops(w.Cout, w.Cin, w.K, Lout_for_work);
  st.bytes = a.N * conv_bytes(w.Cin, a.gate ? w.Cout / 2 : w.Cout, w.K, Lout_for_work);
  st.lane = s.cur_lane;
  st.tag = mfma ? "conv_mfma" : "conv_small";
  s.steps.push_back(std::move(st));
}

// conv step over a bf16 fragment image
void add_conv_bf16(piper_hip_voice* v, Slot& s, const std::string& name, const piper_hip_voice::ConvWB& w, ConvBf16Args a,
                   double flops) {
  a.w = w.w; a.bias = w.bias; a.Cin = w.Cin; a.Cout = w.Cout; a.K = w.K;
  piper_hip_ctx* ctx = v->ctx;
  Step st;
  st.name = name;
  st.run = [ctx, a](hipStream_t q) { return launch_conv_bf16(ctx, q, a); };
  st.flops = flops;
  const double cols = (double)a.N * a.Lout, out_cols = a.ct_stride > 0 ? cols * a.ct_stride : cols;
  st.bytes = 2.0 * (w.Cin * cols + (double)w.Cout * w.Cin * w.K) + (a.act ? 2.0 : 0.0) * w.Cout * out_cols +
             ((a.y ? 4.0 : 0.0) + (a.res ? 4.0 : 0.0) + (a.mrf_a ? 8.0 : 0.0)) * w.Cout * out_cols;
  st.lane = s.cur_lane;
  st.tag = "conv_bf16";
  s.steps.push_back(std::move(st));
}

// several same-shape bf16 convs (the stage's ResBlocks) as one launch
void add_conv_bf16_multi(piper_hip_voice* v, Slot& s, const std::string& name, const piper_hip_voice::ConvWB* const* ws,
                         const ConvBf16Args* args, int count, double flops) {
  struct Pack { ConvBf16Args a[kBf16Multi]; } pk;
  double bytes = 0;
  for (int i = 0; i < count; i++) {
    ConvBf16Args a = args[i];
    const auto& w = *ws[i];
    a.w = w.w; a.bias = w.bias; a.Cin = w.Cin; a.Cout = w.Cout; a.K = w.K;
    pk.a[i] = a;
    const double cols = (double)a.N * a.Lout;
    bytes += 2.0 * (w.Cin * cols + (double)w.Cout * w.Cin * w.K) + (a.act ? 2.0 : 0.0) * w.Cout * cols +
             ((a.y ? 4.0 : 0.0) + (a.res ? 4.0 : 0.0) + (a.mrf_a ? 8.0 : 0.0)) * w.Cout * cols;
  }
  piper_hip_ctx* ctx = v->ctx;
  Step st;
  st.name = name;
  st.run = [ctx, pk, count](hipStream_t q) { return launch_conv_bf16_multi(ctx, q, pk.a, count); };
  st.flops = flops; st.bytes = bytes;
  st.lane = 0;
  st.tag = "conv_bf16";
  s.steps.push_back(std::move(st));
}

// HiFi-GAN generator with bf16 contraction operands. Same graph as the fp32 generator below;
// what changes is the data each conv READS: the C8 bf16 image of LeakyReLU(x) written by its producer's epilogue
// (conv_bf16.h). The residual stream, the bias adds and the MRF mean stay fp32, so rounding enters only through the
// operands of each contraction and does not accumulate along the residual chain.
int build_generator_bf16(piper_hip_voice* v, Slot& s, Arena& ar, const float* z, float* dec0, int F, int NB) {
  const piper_hip_voice_config& c = v->cfg;
  const int I = c.inter;
  const size_t B = (size_t)NB;
  hipStream_t zs = s.stream;
  static const bool no_par = getenv("PIPER_HIP_BF16_SERIAL_RB") != nullptr;
  static const bool no_merge = getenv("PIPER_HIP_NO_MERGED_RB") != nullptr;
  // short utterances / small batches: the three ResBlocks advance in one launch; otherwise one launch per conv, as parallel
  // graph branches for the 18-conv ResBlock1 stages (+10 %; the 6-conv ResBlock2 stages do not gain)
  const bool merged = !no_merge && c.n_rb == 3 && (int64_t)NB * F <= 1536;
  s.parallel = !merged && !no_par && c.resblock_type == 1;
  auto image = [&](int C, int L) -> uint16_t* {  // zeroed once per build: kernels write the interior only
    const size_t bytes = (size_t)c8_elems(NB, C, L) * 2;
    void* p = ar.raw(bytes);
    if (p && hipMemsetAsync(p, 0, bytes, zs) != hipSuccess) ar.rc = PIPER_HIP_ERR_LAUNCH;
    return (uint16_t*)p;
  };
  uint16_t* zc8 = image(I, F);
  uint16_t* a_in = image(c.up_initial, F);
  if (ar.rc) return ar.rc;
  {
    Step st;
    st.name = "dec.z_to_bf16";
    const int* lf = s.lensF;
    st.run = [=](hipStream_t q) { return pack_act_c8(q, z, NB, I, F, 1.0f, zc8, 0, lf); };
    s.steps.push_back(st);
  }
  {
    ConvBf16Args a;
    a.x = zc8; a.y = dec0; a.act = a_in; a.act_alpha = 0.1f;  // the first stage's ConvTranspose reads lrelu(conv_pre)
    a.N = NB; a.dil = 1; a.padL = 3; a.Lout = F; a.x_row = (int)c8_row_len(F); a.act_row = (int)c8_row_len(F); a.y_len = F;
    a.len_ptr = s.lensF; a.len_mul = 1;
    add_conv_bf16(v, s, "dec.conv_pre", v->conv_pre_b, a, NB * conv_flops(c.up_initial, I, 7, F));
  }
  s.taps["dec_pre"] = {dec0, c.up_initial, F, 1, (size_t)c.up_initial * F};
  int L = F;
  const float* mean = nullptr;
  for (int u = 0; u < c.n_ups; u++) {
    const auto& S = v->stages[u];
    const bool last_stage = u + 1 == c.n_ups;
    const int Lo = L * S.stride;
    const int row = (int)c8_row_len(Lo);
    float* up = ar.f32(B * S.Cout * Lo);
    uint16_t* a_up = image(S.Cout, Lo);
    uint16_t* a_next = last_stage ? nullptr : image(S.Cout, Lo);  // lrelu(MRF mean): the next stage's input
    float* m = last_stage ? ar.f32(B * S.Cout * Lo) : nullptr;    // conv_post reads the fp32 mean
    float* r[PIPER_HIP_MAX_RB];
    float* tmp[PIPER_HIP_MAX_RB][2];
    uint16_t* act[PIPER_HIP_MAX_RB][2];
    uint16_t* mid[PIPER_HIP_MAX_RB];
    for (int j = 0; j < c.n_rb; j++) {
      r[j] = ar.f32(B * S.Cout * Lo);
      for (int i = 0; i < 2; i++) {
        tmp[j][i] = ar.f32(B * S.Cout * Lo);
        act[j][i] = image(S.Cout, Lo);
      }
      mid[j] = c.resblock_type == 1 ? image(S.Cout, Lo) : nullptr;
    }
    if (ar.rc) return ar.rc;
    const std::string p = "dec.s" + std::to_string(u) + ".";
    {
      ConvBf16Args a;
      a.x = a_in; a.y = up; a.act = a_up; a.act_alpha = 0.1f;
      a.N = NB; a.Lout = L; a.x_row = (int)c8_row_len(L); a.act_row = row; a.y_len = Lo;
      a.ct_stride = S.stride; a.ct_pad = S.pad;
      a.len_ptr = s.lensF; a.len_mul = Lo / F;
      add_conv_bf16(v, s, p + "lrelu_convT", v->up_b[u], a, NB * 2.0 * S.Cin * S.Cout * (double)S.K * L);
    }
    if (merged) {
      // conv i of rb0, rb1, rb2 have the same shape and no dependence on each other: one launch advances all three. Only the
      // very last conv of rb2 runs alone, after the others, because its epilogue folds the MRF mean over r0, r1.
      const float* src[3] = {up, up, up};
      const uint16_t* src_act[3] = {a_up, a_up, a_up};
      for (int di = 0; di < c.rb_n_dil; di++) {
        const bool lastd = di + 1 == c.rb_n_dil;
        const std::string nm = p + "rb012.c" + std::to_string(di);
        ConvBf16Args aa[3], bb[3];
        const piper_hip_voice::ConvWB *wa[3], *wb[3];
        double fl[3];
        for (int j = 0; j < 3; j++) {
          const int K = c.rb_kernels[j], dl = c.rb_dilations[j][di];
          fl[j] = NB * conv_flops(S.Cout, S.Cout, K, Lo);
          auto base = [&](const uint16_t* in, int d2) {
            ConvBf16Args a;
            a.x = in; a.N = NB; a.dil = d2; a.padL = (K * d2 - d2) / 2; a.Lout = Lo; a.x_row = row; a.act_row = row; a.y_len = Lo;
            a.act_alpha = 0.1f;
            a.len_ptr = s.lensF; a.len_mul = Lo / F;
            return a;
          };
          float* dst = lastd ? (j == 2 ? m : r[j]) : tmp[j][di & 1];
          uint16_t* dst_act = lastd ? (j == 2 ? a_next : nullptr) : act[j][di & 1];
          ConvBf16Args fin = base(c.resblock_type == 1 ? mid[j] : src_act[j], c.resblock_type == 1 ? 1 : dl);
          fin.res = src[j]; fin.y = dst; fin.act = dst_act;
          if (lastd && j == 2) { fin.mrf_a = r[0]; fin.mrf_b = r[1]; }
          bb[j] = fin;
          wb[j] = c.resblock_type == 1 ? &v->rb_b[u][j][2 * di + 1] : &v->rb_b[u][j][di];
          if (c.resblock_type == 1) {
            aa[j] = base(src_act[j], dl);
            aa[j].act = mid[j];
            wa[j] = &v->rb_b[u][j][2 * di];
          }
          src[j] = dst;
          src_act[j] = dst_act;
        }
        // ResBlock1 pairs that are not the stage's last: both convs in one launch, intermediate in LDS (rb_pair_bf16.hip)
        static const bool no_pair = getenv("PIPER_HIP_NO_RB_PAIR") != nullptr;
        if (c.resblock_type == 1 && !no_pair) {
          struct Pack { RbPairBf16Args a[3]; } pk;
          bool ok = true;
          for (int j = 0; j < 3 && ok; j++) {
            RbPairBf16Args& q = pk.a[j];
            q.x = bb[j].res; q.y = bb[j].y;
            // fused pairs stage from the fp32 stream themselves; the stage's last pair writes the next stage's input image, and a
            // pair in front of an UNFUSED one writes the image that one reads
            const bool next_fused = !lastd && rb_pair_bf16_eligible(S.Cout, c.rb_kernels[j], c.rb_dilations[j][di + 1], c.rb_kernels[j], 1, Lo);
            q.act = (lastd || !next_fused) ? bb[j].act : nullptr;
            q.mrf_a = bb[j].mrf_a; q.mrf_b = bb[j].mrf_b;
            q.act_row = row;
            q.wa = wa[j]->w; q.ba = wa[j]->bias; q.wb = wb[j]->w; q.bb = wb[j]->bias;
            q.Ka = wa[j]->K; q.dila = aa[j].dil; q.Kb = wb[j]->K; q.dilb = 1; q.alpha = 0.1f;
            q.N = NB; q.C = S.Cout; q.L = Lo; q.len_ptr = s.lensF; q.len_mul = Lo / F;
            ok = q.x && (q.y || q.act) && q.wa && q.wb && q.ba && q.bb && wa[j]->Cin == S.Cout && wa[j]->Cout == S.Cout && wb[j]->Cin == S.Cout && wb[j]->Cout == S.Cout &&
                 rb_pair_bf16_eligible(S.Cout, q.Ka, q.dila, q.Kb, q.dilb, Lo);
          }
          if (ok) {
            piper_hip_ctx* ctx = v->ctx;
            auto add_pairs = [&](const std::string& name, int first, int count, double flops) {
              struct P2 { RbPairBf16Args a[3]; } p2;
              for (int i = 0; i < 3; i++) p2.a[i] = pk.a[first + (i < count ? i : 0)];
              Step st;
              st.name = name;
              st.tag = "conv_bf16";
              st.flops = 2.0 * flops;
              st.bytes = NB * count * (2.0 * 4.0 * S.Cout * (double)Lo);
              st.run = [ctx, p2, count](hipStream_t q) { return launch_rb_pair_bf16_multi(ctx, q, p2.a, count); };
              s.steps.push_back(std::move(st));
            };
            if (!lastd) add_pairs(nm + "ab_lrelu_conv_lrelu_conv_res_x3", 0, 3, fl[0] + fl[1] + fl[2]);
            else {
              // all three last pairs in one launch (rb2 writes its own fp32 output), then the MRF mean as a small elementwise
              // launch: a single pair of a 128-channel stage is 96 blocks for 256 CUs (52 µs; r3h), the mean 8 µs
              pk.a[2].y = r[2]; pk.a[2].act = nullptr; pk.a[2].mrf_a = nullptr; pk.a[2].mrf_b = nullptr;
              add_pairs(nm + "ab_lrelu_conv_lrelu_conv_res_x3", 0, 3, fl[0] + fl[1] + fl[2]);
              Step st;
              st.name = p + "mrf_mean" + (last_stage ? "" : "_lrelu_to_bf16");
              const float *r0 = r[0], *r1 = r[1], *r2 = r[2];
              const int* lf = s.lensF;
              const int Cc = S.Cout, lm = Lo / F;
              if (last_stage) {  // conv_post applies its LeakyReLU(0.01) itself: slope 1 here
                const int64_t cnt = (int64_t)NB * S.Cout * Lo;
                st.run = [=](hipStream_t q) {
                  const int grid = (int)std::min<int64_t>(ceil_div(cnt, (int64_t)kBlock * 4), 2048);
                  hipLaunchKernelGGL(mrf_mean_lrelu_kernel, dim3(grid), dim3(kBlock), 0, q, r0, r1, r2, m, cnt, 1.0f);
                  return PIPER_HIP_OK;
                };
              } else {
                st.run = [=](hipStream_t q) { return pack_mean3_c8(q, r0, r1, r2, NB, Cc, Lo, 0.1f, a_next, row, lf, lm); };
              }
              s.steps.push_back(st);
            }
            continue;
          }
        }
        if (c.resblock_type == 1) add_conv_bf16_multi(v, s, nm + "a_lrelu_conv_x3", wa, aa, 3, fl[0] + fl[1] + fl[2]);
        const std::string bn = c.resblock_type == 1 ? "b" : "";
        if (!lastd) {
          add_conv_bf16_multi(v, s, nm + bn + "_lrelu_conv_res_x3", wb, bb, 3, fl[0] + fl[1] + fl[2]);
        } else {
          add_conv_bf16_multi(v, s, nm + bn + "_lrelu_conv_res_x2", wb, bb, 2, fl[0] + fl[1]);
          add_conv_bf16_multi(v, s, nm + bn + "_lrelu_conv_res_mrfmean", wb + 2, bb + 2, 1, fl[2]);
        }
      }
      a_in = a_next;
      mean = m;
      L = Lo;
      continue;
    }
    // The stage's three ResBlocks are independent chains of short, latency-bound launches that leave most CUs idle:
    // rb0 / rb1 run as side branches of the graph, rb2 on the main lane, joined before rb2's last conv (which folds the
    // MRF mean over all three).
    if (s.parallel) {
      Step f;
      f.name = p + "fork";
      f.kind = Step::FORK;
      s.steps.push_back(f);
    }
    for (int j = 0; j < c.n_rb; j++) {
      const int K = c.rb_kernels[j];
      const float* src = up;
      const uint16_t* src_act = a_up;
      s.cur_lane = (j + 1 == c.n_rb) ? 0 : j + 1;
      for (int di = 0; di < c.rb_n_dil; di++) {
        const int dil = c.rb_dilations[j][di];
        const bool lastd = di + 1 == c.rb_n_dil;
        const bool fuse_mean = lastd && j + 1 == c.n_rb;  // r0, r1 are complete: fold (r0+r1+r2)/3 into this epilogue
        float* dst = lastd ? (fuse_mean ? m : r[j]) : tmp[j][di & 1];
        uint16_t* dst_act = lastd ? (fuse_mean ? a_next : nullptr) : act[j][di & 1];
        const std::string nm = p + "rb" + std::to_string(j) + ".c" + std::to_string(di);
        auto rbconv = [&](const uint16_t* in, int dl) {
          ConvBf16Args a;
          a.x = in; a.N = NB; a.dil = dl; a.padL = (K * dl - dl) / 2; a.Lout = Lo; a.x_row = row; a.act_row = row; a.y_len = Lo;
          a.act_alpha = 0.1f;
          a.len_ptr = s.lensF; a.len_mul = Lo / F;
          return a;
        };
        auto finish = [&](ConvBf16Args a) {  // the conv that closes the residual: x ← x + conv(…)
          a.res = src; a.y = dst; a.act = dst_act;
          if (fuse_mean) { a.mrf_a = r[0]; a.mrf_b = r[1]; }
          return a;
        };
        const double fl = NB * conv_flops(S.Cout, S.Cout, K, Lo);
        auto join = [&]() {
          if (!(fuse_mean && s.parallel)) return;
          Step jn;
          jn.name = p + "join";
          jn.kind = Step::JOIN;
          s.steps.push_back(jn);
        };
        if (c.resblock_type == 1) {
          ConvBf16Args a1 = rbconv(src_act, dil);
          a1.act = mid[j];
          add_conv_bf16(v, s, nm + "a_lrelu_conv", v->rb_b[u][j][2 * di], a1, fl);
          join();
          add_conv_bf16(v, s, nm + (fuse_mean ? "b_lrelu_conv_res_mrfmean" : "b_lrelu_conv_res"), v->rb_b[u][j][2 * di + 1],
                        finish(rbconv(mid[j], 1)), fl);
        } else {
          join();
          add_conv_bf16(v, s, nm + (fuse_mean ? "_lrelu_conv_res_mrfmean" : "_lrelu_conv_res"), v->rb_b[u][j][di],
                        finish(rbconv(src_act, dil)), fl);
        }
        src = dst;
        src_act = dst_act;
      }
    }
    s.cur_lane = 0;
    a_in = a_next;
    mean = m;
    L = Lo;
  }
  s.n_samples = L;
  s.audio = ar.f32(B * L);
  if (ar.rc) return ar.rc;
  {
    ConvArgs a;  // 32→1 k7: thread-per-output fp32 kernel on the fp32 mean, LeakyReLU(0.01) as its prologue
    a.x = mean;
    a.prologue = PRO_LRELU; a.alpha = 0.01f;
    a.y = s.audio; a.N = NB; a.padL = 3; a.Lin = L; a.Lout = L;
    a.len_ptr = s.lensF; a.len_mul = L / F;
    a.x_batch_stride = (int64_t)v->conv_post.Cin * L; a.y_batch_stride = L; a.y_len = L;
    a.epilogue = EPI_TANH;
    add_conv(v, s, "dec.conv_post_tanh", v->conv_post, a, L);
  }
  return PIPER_HIP_OK;
}

// fp32 HiFi-GAN generator with the stage's three ResBlocks advanced together: conv i of rb0, rb1, rb2 are independent and
// have the same shape (they differ in kernel size, dilation, weights, buffers), so they run as ONE window-kernel launch
// (launch_conv_win_multi). The MRF mean (r0+r1+r2)/3 is folded into the consumer's staging (ConvTranspose of the next
// stage / conv_post). Medium voice: 23 → 11 generator launches; high: 77 → 29. Returns UNSUPPORTED (nothing scheduled)
// when a conv falls outside the window kernel's geometry, and the caller schedules the per-conv path instead.
int build_generator_merged(piper_hip_voice* v, Slot& s, Arena& ar, float* dec0, int F, int NB) {
  const piper_hip_voice_config& c = v->cfg;
  piper_hip_ctx* ctx = v->ctx;
  const size_t B = (size_t)NB;
  if (c.n_rb != kWinMulti) return PIPER_HIP_ERR_UNSUPPORTED;
  // A/B switches: PIPER_HIP_NO_PIPE=1 keeps round 1's one-tile-per-block window kernel; PIPER_HIP_PIPE_MIN_F (frames × batch)
  // is the size below which the window kernel's in-block K-split still wins (very short utterances: fewer tiles than CUs)
  // conv_pipe (persistent, chunk-pipelined) wins once a launch holds enough work to amortise its prologue and tail — the
  // high voice's 128/256-channel stages: 89 vs 73 TFLOP/s — and loses to the one-tile-per-block window kernel on the medium
  // voice at factor 8 (48 vs 46 µs per merged launch). Threshold on the launch's FLOPs; PIPER_HIP_PIPE_MIN_GFLOP overrides.
  static const bool no_pipe = getenv("PIPER_HIP_NO_PIPE") != nullptr;
  static const double pipe_min_flops = [] { const char* e = getenv("PIPER_HIP_PIPE_MIN_GFLOP"); return (e ? atof(e) : 5.0) * 1e9; }();
  // r2f (factor 64): as a single-conv launch it also loses at 32 / 64 channels (135 vs 114 µs, 105 vs 100 µs: one chunk per
  // tile leaves nothing to pipeline, and 2 blocks per CU hide less than the window kernel's 4) and wins from 128 channels up.
  // ConvTranspose: it won (112 vs 145 µs) until the window kernel staged narrow windows with a flat index and kept the phases
  // of a column range in one block; since then the window kernel wins at every size measured (factor 64: 126 / 121 vs
  // 187 / 204 µs, 8 × factor 8: 111 / 119 vs 181 / 201 µs, high voice factor 32: 215 vs 312 µs) — never by default,
  // PIPER_HIP_PIPE_CT_MIN_GFLOP=g brings it back for launches of ≥ g GFLOP.
  static const double pipe_ct_min_flops = [] { const char* e = getenv("PIPER_HIP_PIPE_CT_MIN_GFLOP"); return e ? atof(e) * 1e9 : 1e30; }();
  auto pipe_pays = [&](double launch_flops, int Cin, bool ct) { return !no_pipe && launch_flops >= (ct ? pipe_ct_min_flops : pipe_min_flops) && (ct || Cin >= 128); };
  {
    int L = F;
    for (int u = 0; u < c.n_ups; u++) {
      const auto& S = v->stages[u];
      const int Lo = L * S.stride;
      for (int j = 0; j < c.n_rb; j++)
        for (const ConvW& w : S.rb[j]) {
          int dmax = 1;
          for (int di = 0; di < c.rb_n_dil; di++) dmax = std::max(dmax, (int)c.rb_dilations[j][di]);
          if (!w.w4 || !conv_win_eligible(w.Cout, w.Cin, w.K, dmax, (w.K * dmax - dmax) / 2, Lo, Lo)) return PIPER_HIP_ERR_UNSUPPORTED;
        }
      L = Lo;
    }
  }
  const float* cur[3] = {dec0, nullptr, nullptr};  // stage input: one tensor, or the three ResBlock outputs to average
  int L = F;
  for (int u = 0; u < c.n_ups; u++) {
    const auto& S = v->stages[u];
    const int Lo = L * S.stride;
    float* up = ar.f32(B * S.Cout * Lo);
    float* buf[kWinMulti][2];
    float* mid[kWinMulti];
    for (int j = 0; j < c.n_rb; j++) {
      buf[j][0] = ar.f32(B * S.Cout * Lo);
      buf[j][1] = ar.f32(B * S.Cout * Lo);
      mid[j] = c.resblock_type == 1 ? ar.f32(B * S.Cout * Lo) : nullptr;
    }
    if (ar.rc) return ar.rc;
    const std::string p = "dec.s" + std::to_string(u) + ".";
    {  // ConvTranspose on lrelu(input) — input = conv_pre output, or the mean of the previous stage's ResBlocks
      Step st;
      st.name = p + (cur[1] ? "mrfmean_lrelu_convT" : "lrelu_convT");
      st.tag = "conv_mfma";
      st.flops = NB * 2.0 * S.Cin * S.Cout * (double)S.K * L;
      st.bytes = NB * 4.0 * ((double)S.Cin * L * (cur[1] ? 3 : 1) + (double)S.Cout * Lo + (double)S.Cin * S.Cout * S.K + S.Cout);
      const bool ct_pipe = pipe_pays(st.flops, S.Cin, true) && S.up.w5 && convt_pipe_eligible(S.Cin, S.Cout, S.K, S.stride, S.pad, L);
      if (ct_pipe || (S.up.w4 && convt_win_eligible(S.Cin, S.Cout, S.K, S.stride, S.pad, L))) {
        ConvWinArgs wa;
        wa.x = cur[0]; wa.x2 = cur[1]; wa.x3 = cur[2]; wa.w4 = ct_pipe ? S.up.w5 : S.up.w4; wa.bias = S.up.bias; wa.y = up;
        wa.pro_alpha = 0.1f;
        wa.N = NB; wa.Cin = S.Cin; wa.Cout = S.Cout; wa.K = S.K; wa.Lin = L; wa.Lout = L; wa.y_len = Lo;
        wa.ct_stride = S.stride; wa.ct_pad = S.pad;
        wa.len_ptr = s.lensF; wa.len_mul = L / F;
        if (ct_pipe) st.run = [ctx, wa](hipStream_t q) { return launch_conv_pipe_multi(ctx, q, &wa, 1); };
        else st.run = [ctx, wa](hipStream_t q) { return launch_conv_win(ctx, q, wa); };
      } else {
        ConvArgs a;
        a.x = cur[0]; a.x2 = cur[1]; a.x3 = cur[2];
        a.prologue = cur[1] ? PRO_AVG3_LRELU : PRO_LRELU;
        a.alpha = 0.1f;
        a.y = up; a.N = NB; a.dil = -1; a.padL = 0; a.Lin = L; a.Lout = (Lo - 1 + S.pad) / S.stride + 1;
        a.len_ptr = s.lensF; a.len_mul = L / F;
        a.x_batch_stride = (int64_t)S.Cin * L; a.y_batch_stride = (int64_t)S.Cout * Lo; a.y_len = Lo;
        a.epilogue = EPI_CONVT; a.ct_stride = S.stride; a.ct_padL = S.pad; a.ct_Lout = Lo;
        a.w = S.up.w; a.w16 = S.up.w16; a.bias = S.up.bias; a.Cin = S.up.Cin; a.Cout = S.up.Cout; a.K = S.up.K;
        st.run = [ctx, a](hipStream_t q) { return launch_conv_mfma(ctx, q, a); };
      }
      s.steps.push_back(st);
    }
    const float* src[kWinMulti] = {up, up, up};
    auto add_multi = [&](const std::string& name, const ConvW* ws[kWinMulti], const float* const x[kWinMulti],
                         const float* const res[kWinMulti], float* const y[kWinMulti], const int dil[kWinMulti]) {
      struct Pack { ConvWinArgs a[kWinMulti]; } pk;
      double fl = 0, by = 0;
      double launch_fl = 0;
      for (int j = 0; j < kWinMulti; j++) launch_fl += NB * conv_flops(ws[j]->Cout, ws[j]->Cin, ws[j]->K, Lo);
      bool pipe = pipe_pays(launch_fl, ws[0]->Cin, false);
      for (int j = 0; j < kWinMulti; j++)
        pipe = pipe && ws[j]->w5 && conv_pipe_eligible(ws[j]->Cout, ws[j]->Cin, ws[j]->K, dil[j], (ws[j]->K * dil[j] - dil[j]) / 2, Lo, Lo);
      for (int j = 0; j < kWinMulti; j++) {
        ConvWinArgs& wa = pk.a[j];
        const ConvW& w = *ws[j];
        wa.x = x[j]; wa.w4 = pipe ? w.w5 : w.w4; wa.bias = w.bias; wa.res = res[j]; wa.y = y[j];
        wa.pro_alpha = 0.1f;
        wa.N = NB; wa.Cin = w.Cin; wa.Cout = w.Cout; wa.K = w.K; wa.dil = dil[j]; wa.padL = (w.K * dil[j] - dil[j]) / 2;
        wa.Lin = Lo; wa.Lout = Lo; wa.y_len = Lo;
        wa.len_ptr = s.lensF; wa.len_mul = Lo / F;
        fl += NB * conv_flops(w.Cout, w.Cin, w.K, Lo);
        by += NB * conv_bytes(w.Cin, w.Cout, w.K, Lo);
      }
      Step st;
      st.name = name;
      if (pipe) st.run = [ctx, pk](hipStream_t q) { return launch_conv_pipe_multi(ctx, q, pk.a, kWinMulti); };
      else st.run = [ctx, pk](hipStream_t q) { return launch_conv_win_multi(ctx, q, pk.a, kWinMulti); };
      st.flops = fl; st.bytes = by;
      st.tag = "conv_mfma";
      s.steps.push_back(std::move(st));
    };
    // two chained convs per launch, intermediate in LDS (rb_pair.hip): ResBlock1 — (convs1[di], convs2[di]); ResBlock2 —
    // steps (di, di+1). PIPER_HIP_NO_RB_PAIR=1 keeps the conv-by-conv schedule (A/B). Very short utterances (under 128 frames in the
    // launch: factors 1 and 2) leave the pair kernel's 256-column tiles too few blocks — 41 at factor 1 — and run conv by conv
    // (r2: factor 1 0.649 → 0.620 ms, factor 2 0.670 → 0.659; from factor 4 on the pair kernel wins). PIPER_HIP_RB_PAIR_MIN_F moves it.
    static const bool no_pair_env = getenv("PIPER_HIP_NO_RB_PAIR") != nullptr;
    static const int64_t pair_min_f = [] { const char* e = getenv("PIPER_HIP_RB_PAIR_MIN_F"); return e ? atoll(e) : 128ll; }();
    const bool no_pair = no_pair_env || (int64_t)F * NB < pair_min_f;
    auto add_pair = [&](const std::string& name, int ia, int ib, const int da[kWinMulti], const int db[kWinMulti], bool res_a, bool res_b_x,
                        const float* const x[kWinMulti], float* const y[kWinMulti]) {
      struct Pack { RbPairArgs a[kWinMulti]; } pk;
      double fl = 0, by = 0;
      for (int j = 0; j < kWinMulti; j++) {
        const ConvW &wa = S.rb[j][ia], &wb = S.rb[j][ib];
        if (no_pair || !wa.w4 || !wb.w4 || !wa.bias || !wb.bias || wa.Cin != wa.Cout || wa.Cin != wb.Cin || wb.Cin != wb.Cout ||
            !rb_pair_eligible(wa.Cin, wa.K, da[j], wb.K, db[j], Lo))
          return false;
        RbPairArgs& a = pk.a[j];
        a.x = x[j]; a.y = y[j]; a.wa4 = wa.w4; a.ba = wa.bias; a.wb4 = wb.w4; a.bb = wb.bias;
        a.Ka = wa.K; a.dila = da[j]; a.Kb = wb.K; a.dilb = db[j]; a.res_a = res_a; a.res_b_x = res_b_x; a.alpha = 0.1f;
        a.N = NB; a.C = wa.Cin; a.L = Lo; a.len_ptr = s.lensF; a.len_mul = Lo / F;
        fl += NB * (conv_flops(wa.Cout, wa.Cin, wa.K, Lo) + conv_flops(wb.Cout, wb.Cin, wb.K, Lo));
        by += NB * 4.0 * (2.0 * wa.Cin * (double)Lo + (double)wa.Cin * wa.Cin * (wa.K + wb.K) + 2.0 * wa.Cin);  // x in, y out, weights
      }
      Step st;
      st.name = name;
      st.run = [ctx, pk](hipStream_t q) { return launch_rb_pair_multi(ctx, q, pk.a, kWinMulti); };
      st.flops = fl; st.bytes = by;
      st.tag = "conv_mfma";
      s.steps.push_back(std::move(st));
      return true;
    };
    for (int di = 0; di < c.rb_n_dil; di++) {
      float* dst[kWinMulti];
      int dil[kWinMulti], one[kWinMulti] = {1, 1, 1};
      const float* none[kWinMulti] = {nullptr, nullptr, nullptr};
      for (int j = 0; j < kWinMulti; j++) { dst[j] = src[j] == buf[j][0] ? buf[j][1] : buf[j][0]; dil[j] = c.rb_dilations[j][di]; }  // never the buffer being read
      const std::string nm = p + "rb012.c" + std::to_string(di);
      if (c.resblock_type == 1 && add_pair(nm + "ab_lrelu_conv_lrelu_conv_res_x3", 2 * di, 2 * di + 1, dil, one, false, true, src, dst)) {
        for (int j = 0; j < kWinMulti; j++) src[j] = dst[j];
        continue;
      }
      if (c.resblock_type == 2 && di + 1 < c.rb_n_dil) {
        int dil2[kWinMulti];
        for (int j = 0; j < kWinMulti; j++) dil2[j] = c.rb_dilations[j][di + 1];
        if (add_pair(p + "rb012.c" + std::to_string(di) + std::to_string(di + 1) + "_lrelu_conv_res_pair_x3", di, di + 1, dil, dil2, true, false, src, dst)) {
          for (int j = 0; j < kWinMulti; j++) src[j] = dst[j];
          di++;
          continue;
        }
      }
      if (c.resblock_type == 1) {
        const ConvW* wa[kWinMulti] = {&S.rb[0][2 * di], &S.rb[1][2 * di], &S.rb[2][2 * di]};
        const ConvW* wb[kWinMulti] = {&S.rb[0][2 * di + 1], &S.rb[1][2 * di + 1], &S.rb[2][2 * di + 1]};
        const float* midc[kWinMulti] = {mid[0], mid[1], mid[2]};
        add_multi(nm + "a_lrelu_conv_x3", wa, src, none, mid, dil);
        add_multi(nm + "b_lrelu_conv_res_x3", wb, midc, src, dst, one);
      } else {
        const ConvW* w[kWinMulti] = {&S.rb[0][di], &S.rb[1][di], &S.rb[2][di]};
        add_multi(nm + "_lrelu_conv_res_x3", w, src, src, dst, dil);
      }
      for (int j = 0; j < kWinMulti; j++) src[j] = dst[j];
    }
    for (int j = 0; j < kWinMulti; j++) cur[j] = src[j];
    L = Lo;
  }
  s.n_samples = L;
  s.audio = ar.f32(B * L);
  if (ar.rc) return ar.rc;
  {
    ConvArgs a;
    a.x = cur[0]; a.x2 = cur[1]; a.x3 = cur[2];
    a.prologue = PRO_AVG3_LRELU; a.alpha = 0.01f;  // F.leaky_relu default slope before conv_post, on the MRF mean
    a.y = s.audio; a.N = NB; a.padL = 3; a.Lin = L; a.Lout = L;
    a.len_ptr = s.lensF; a.len_mul = L / F;
    a.x_batch_stride = (int64_t)v->conv_post.Cin * L; a.y_batch_stride = L; a.y_len = L;
    a.epilogue = EPI_TANH;
    add_conv(v, s, "dec.mrfmean_conv_post_tanh", v->conv_post, a, L);
  }
  return PIPER_HIP_OK;
}

// the duration predictor on the encoder output x [NB][H][T]: → s.dp_dur / "logw" tap
int build_duration_predictor(piper_hip_voice* v, Slot& s, Arena& ar, const float* x, int T, int NB);

// mode 0: whole utterance; 1: generator only (streaming window); 2: text encoder + projection + duration predictor;
// 3: everything AFTER the text encoder (expansion, flow, generator) from an m_p / logs_p tensor copied in from a mode-2 plan —
// the pair (2, 3) is a whole utterance with predicted durations that runs the encoder once
int build_schedule(piper_hip_voice* v, Slot& s, int T, int F, int NB, int mode = 0) {
  const bool gen_only = mode == 1;
  const bool from_stats = mode == 3;
  const piper_hip_voice_config& c = v->cfg;
  piper_hip_ctx* ctx = v->ctx;
  const int H = c.hidden, I = c.inter, d = H / c.n_heads;
  slot_release(v, s, false);
  static const bool parallel_rb = getenv("PIPER_HIP_PARALLEL_RB") != nullptr;
  static const bool use_win = getenv("PIPER_HIP_NO_WIN") == nullptr;  // window kernel for the generator's long rows
  static const bool no_pipe1 = getenv("PIPER_HIP_NO_PIPE") != nullptr;
  static const double pipe_min_flops1 = [] { const char* e = getenv("PIPER_HIP_PIPE_MIN_GFLOP"); return (e ? atof(e) : 5.0) * 1e9; }();
  static const double pipe_ct_min_flops1 = [] { const char* e = getenv("PIPER_HIP_PIPE_CT_MIN_GFLOP"); return e ? atof(e) * 1e9 : 1e30; }();
  auto pipe_pays1 = [&](double launch_flops, int Cin, bool ct) { return !no_pipe1 && launch_flops >= (ct ? pipe_ct_min_flops1 : pipe_min_flops1) && (ct || Cin >= 128); };
  Arena ar{v, &s};
  if (c.n_rb != 3) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "voice: n_rb=%d (only the 3-kernel MRF of Piper voices is scheduled)", c.n_rb);
  s.T = T; s.F = F; s.NB = NB;
  s.kind = mode;
  s.prec = v->precision;
  s.arena_bytes = 0;
  s.parallel = parallel_rb;
  const size_t B = (size_t)NB;
  s.lensT = (int*)ar.raw(B * sizeof(int));
  s.lensF = (int*)ar.raw(B * sizeof(int));
  if (ar.rc) return ar.rc;
  const int* lensT = s.lensT;
  const int* lensF = s.lensF;
  // lens = the per-item true lengths the rows of this conv are measured in (phonemes or frames), mul = positions per unit
  auto plain = [&](const float* in, float* out, int Cin_, int Cout_, int L, const int* lens, int mul = 1) {
    ConvArgs a;
    a.x = in; a.y = out; a.N = NB; a.Lin = L; a.Lout = L; a.x_batch_stride = (int64_t)Cin_ * L; a.y_batch_stride = (int64_t)Cout_ * L;
    a.y_len = L;
    a.len_ptr = lens; a.len_mul = mul;
    return a;
  };
  const float* z = nullptr;
  float* dec0 = nullptr;
  if (gen_only) {  // streaming: only the generator, over a window of the latent that the caller copies into zin
    s.zin = ar.f32(B * (size_t)I * F);
    dec0 = ar.f32(B * (size_t)c.up_initial * F);
    if (ar.rc) return ar.rc;
    z = s.zin;
  } else {
  s.ids = (int64_t*)ar.raw(B * T * sizeof(int64_t));
  s.frame2id = (int32_t*)ar.raw(B * F * sizeof(int32_t));
  s.noise = ar.f32(B * I * F);
  s.noise_scale = ar.f32(B);
  s.rng = (unsigned*)ar.raw(B * 2 * sizeof(unsigned));
  // ---------------- text encoder
  float* x = ar.f32(B * (size_t)H * T);
  float* x1 = ar.f32(B * (size_t)H * T);
  float* qkv = ar.f32(B * (size_t)3 * H * T);
  float* att = ar.f32(B * (size_t)H * T);
  // long rows: the attention core runs key-split in two parts plus a merge (attention.hip); scratch for the parts, shared by the layers
  const int att_parts = rel_attention_split_parts(ctx, NB, c.n_heads, H / std::max(1, c.n_heads), T, c.window);
  float* att_po = att_parts > 1 ? ar.f32(B * (size_t)att_parts * H * T) : nullptr;
  float* att_pml = att_parts > 1 ? ar.f32(B * (size_t)c.n_heads * att_parts * 2 * T) : nullptr;
  float* y = ar.f32(B * (size_t)H * T);
  float* ff = ar.f32(B * (size_t)c.ffn * T);
  float* stats = ar.f32(B * (size_t)2 * I * T);
  float* zp = ar.f32(B * (size_t)I * F);
  float* zflip = ar.f32(B * (size_t)I * F);
  float* zp_tap = ar.f32(B * (size_t)I * F);
  float* h = ar.f32(B * (size_t)H * F);
  float* acts = ar.f32(B * (size_t)H * F);
  float* skip = ar.f32(B * (size_t)H * F);
  dec0 = ar.f32(B * (size_t)c.up_initial * F);
  if (ar.rc) return ar.rc;
  s.stats = stats;
  if (!from_stats) {
  {
    Step st;
    st.name = "embed";
    const int64_t* ids = s.ids;
    const float* emb = tensor(v, "enc_p.emb.weight");
    const int nv = c.n_vocab;
    const float scale = sqrtf((float)H);
    st.run = [=](hipStream_t q) {
      const int grid = (int)std::min<int64_t>(ceil_div((int64_t)H * T, kBlock), 2048);
      hipLaunchKernelGGL(embed_kernel, dim3(grid, NB), dim3(kBlock), 0, q, ids, emb, x, H, T, nv, scale);
      return PIPER_HIP_OK;
    };
    s.steps.push_back(st);
  }
  const int kf = c.ffn_kernel;
  // LayerNorms without launches of their own: the conv BEFORE a LayerNorm adds the residual and leaves per-column partial
  // sums (stats_out), the conv AFTER it normalises its input on load (PRO_LN) and writes the normalised tensor once.
  // PIPER_HIP_NO_LN_FUSE=1 keeps the add+LayerNorm kernels (A/B).
  // Above ≈ 640 columns (r2: factor 64 and 8 × factor 8, T·NB = 896: 3.07–3.09 vs 3.10 ms, 2.87 vs 2.90 ms) the normalisation
  // inside the consumers' K loops costs more than the twelve launches it saves; up to T·NB = 448 the fused form wins (factor 8:
  // 0.853 vs 0.879 ms). PIPER_HIP_LN_FUSE_MAX_T moves the crossover.
  static const bool ln_fuse = getenv("PIPER_HIP_NO_LN_FUSE") == nullptr;
  static const int64_t ln_fuse_max_t = [] { const char* e = getenv("PIPER_HIP_LN_FUSE_MAX_T"); return e ? atoll(e) : 640ll; }();
  const bool ln_ok = ln_fuse && (kf == 1 || kf == 3) && v->proj.mfma && H <= 256 && (int64_t)T * NB <= ln_fuse_max_t;
  float* st1 = ln_ok ? ar.f32(B * (size_t)ceil_div(H, 16) * T * 2) : nullptr;
  float* st2 = ln_ok ? ar.f32(B * (size_t)ceil_div(H, 16) * T * 2) : nullptr;
  if (ar.rc) return ar.rc;
  // Round 3: where conv_lean.hip takes all three consumers (qkv, ffn1, proj of one utterance: a block holds every channel of its
  // columns), the CONSUMER computes the LayerNorm statistics of its own operand (ConvArgs::ln_self) and the producers are plain
  // residual adds — no statistics tensor crosses the kernel boundary.
  const bool ln_self = ln_ok && mode != 2 && conv_lean_ln_self_ok(ctx, H, 3 * H, 1, 0, T, NB) && conv_lean_ln_self_ok(ctx, H, c.ffn, kf, (kf - 1) / 2, T, NB) &&
                       conv_lean_ln_self_ok(ctx, H, 2 * I, 1, 0, T, NB);
  auto with_ln = [&](ConvArgs a, const float* stats, const float* g, const float* be, float* normalised) {
    a.prologue = PRO_LN;
    a.ln_stats = ln_self ? nullptr : stats; a.ln_self = ln_self ? 1 : 0;
    a.ln_gamma = g; a.ln_beta = be; a.ln_out = normalised; a.ln_eps = 1e-5f;
    return a;
  };
  for (int l = 0; l < c.n_layers; l++) {
    const auto& L = v->enc[l];
    const std::string p = "enc" + std::to_string(l) + ".";
    if (ln_ok && l > 0)  // x = LN2 of the previous layer, applied to y = x1 + ffn2(…) on load; materialised into x
      add_conv(v, s, p + "ln2_qkv", L.qkv, with_ln(plain(y, qkv, H, 3 * H, T, lensT), st2, v->enc[l - 1].g2, v->enc[l - 1].b2, x), T);
    else
    add_conv(v, s, p + "qkv", L.qkv, plain(x, qkv, H, 3 * H, T, lensT), T);
    auto add_ln = [&](const std::string& nm, const float* a, const float* b, const float* g, const float* be, float* out) {
      Step st;
      st.name = nm;
      st.tag = "add_layernorm";
      st.run = [=](hipStream_t q) {
        float* o = out;
        return piper_hip_add_layernorm_f32(ctx, a, b, g, be, NB, H, T, 1e-5f, &o, (piper_hip_stream)q);
      };
      s.steps.push_back(st);
    };
    bool ln1_pending = false;  // LN1 still to be applied by ffn1's prologue (y holds x + conv_o(att), st1 its statistics)
    // mm(2,T,T,96) ×2 + mm(2,T,2T−1,96) ×2
    const double att_flops = NB * 2.0 * c.n_heads * ((double)T * T * d * 2 + (double)T * (2 * T - 1) * d * 2);
    const double att_bytes = NB * 4.0 * c.n_heads * (2.0 * ((double)T * d + (double)d * T + (double)T * T) + 2.0 * ((double)T * d + (double)d * (2 * T - 1) + (double)T * (2 * T - 1)));
    if (L.o.w16 && attention_block_wanted() && attention_block_eligible(c.n_heads, d, c.window, T)) {
      // attention + conv_o + Add + LayerNorm in one launch: the block owns every channel of its 16 columns
      Step st;
      st.name = p + "attention_o_add_ln1";
      st.tag = "attention_block";
      const float *ek = L.ek, *ev = L.ev, *wo = L.o.w16, *bo = L.o.bias, *g1 = L.g1, *b1 = L.b1;
      const int nh = c.n_heads, w = c.window;
      const int o_nsteps = (int)(packed_conv_floats(H, H, 1, 16) / ((size_t)ceil_div(H, 16) * 64));
      st.run = [=](hipStream_t q) {
        return launch_attention_block(ctx, q, qkv, qkv + (size_t)H * T, qkv + (size_t)2 * H * T, ek, ev, wo, bo, x, g1, b1, x1, NB, nh, d, T, w,
                                      (int64_t)3 * H * T, (int64_t)H * T, lensT, o_nsteps, 1e-5f);
      };
      st.flops = att_flops + NB * conv_flops(H, H, 1, T);
      st.bytes = att_bytes + NB * conv_bytes(H, H, 1, T);
      s.steps.push_back(st);
    } else {
    {
      Step st;
      st.name = p + "rel_attention";
      st.tag = "rel_attention";
      const float *ek = L.ek, *ev = L.ev;
      const int nh = c.n_heads, w = c.window;
      st.run = [=](hipStream_t q) {
        if (att_parts > 1)
          return launch_rel_attention_split(ctx, q, qkv, qkv + (size_t)H * T, qkv + (size_t)2 * H * T, ek, ev, att, NB, nh, d, T, w,
                                            (int64_t)3 * H * T, (int64_t)H * T, lensT, att_parts, att_po, att_pml);
        return launch_rel_attention(ctx, q, qkv, qkv + (size_t)H * T, qkv + (size_t)2 * H * T, ek, ev, att, NB, nh, d, T, w,
                                    (int64_t)3 * H * T, (int64_t)H * T, lensT);
      };
      st.flops = att_flops;
      st.bytes = att_bytes;
      s.steps.push_back(st);
    }
    if (ln_ok) {  // y = x + conv_o(att) with its LayerNorm statistics; the normalisation itself happens in ffn1's prologue
      ConvArgs a = plain(att, y, H, H, T, lensT);
      a.res = x; a.stats_out = ln_self ? nullptr : st1;
      add_conv(v, s, p + (ln_self ? "o_add" : "o_add_stats"), L.o, a, T);
      ln1_pending = true;
    } else {
    add_conv(v, s, p + "o", L.o, plain(att, y, H, H, T, lensT), T);
    add_ln(p + "add_ln1", x, y, L.g1, L.b1, x1);
    }
    }
    {
      ConvArgs a = plain(ln1_pending ? y : x1, ff, H, c.ffn, T, lensT);
      a.padL = (kf - 1) / 2;
      if (ln1_pending) a = with_ln(a, st1, L.g1, L.b1, x1);
      a.epilogue = EPI_RELU;
      add_conv(v, s, p + (ln1_pending ? "ln1_ffn1_relu" : "ffn1_relu"), L.f1, a, T);
      ConvArgs b = plain(ff, y, c.ffn, H, T, lensT);
      b.padL = (kf - 1) / 2;
      if (ln_ok) { b.res = x1; b.stats_out = ln_self ? nullptr : st2; }
      add_conv(v, s, p + (ln_self ? "ffn2_add" : ln_ok ? "ffn2_add_stats" : "ffn2"), L.f2, b, T);
    }
    if (!ln_ok) add_ln(p + "add_ln2", x1, y, L.g2, L.b2, x);
  }
  s.taps["enc_out"] = {x, H, T, 0, (size_t)H * T};
  if (mode == 2) {  // x must exist in memory for the predictor: with the LayerNorm folded into its consumer, materialise it
    if (ln_ok && c.n_layers > 0) {
      Step st;
      st.name = "enc.ln2_final";
      st.tag = "add_layernorm";
      const float *g2 = v->enc[c.n_layers - 1].g2, *b2 = v->enc[c.n_layers - 1].b2;
      st.run = [=](hipStream_t q) {
        float* o = x;
        return piper_hip_add_layernorm_f32(ctx, y, nullptr, g2, b2, NB, H, T, 1e-5f, &o, (piper_hip_stream)q);
      };
      s.steps.push_back(st);
    }
    add_conv(v, s, "enc.proj", v->proj, plain(x, stats, H, 2 * I, T, lensT), T);  // kept for the mode-3 plan that continues from here
    return build_duration_predictor(v, s, ar, x, T, NB);
  }
  if (ln_ok && c.n_layers > 0)
    add_conv(v, s, "enc.ln2_proj", v->proj, with_ln(plain(y, stats, H, 2 * I, T, lensT), st2, v->enc[c.n_layers - 1].g2, v->enc[c.n_layers - 1].b2, x), T);
  else
  add_conv(v, s, "enc.proj", v->proj, plain(x, stats, H, 2 * I, T, lensT), T);
  }  // !from_stats
  s.taps["m_p"] = {stats, I, T, 0, (size_t)2 * I * T};  // halves of the [2I, T] projection
  s.taps["logs_p"] = {stats + (size_t)I * T, I, T, 0, (size_t)2 * I * T};
  {
    Step st;
    st.name = "expand_noise";
    const int32_t* f2i = s.frame2id;
    const float* nz = s.noise;
    const float* nsd = s.noise_scale;
    const unsigned* rngd = s.rng;
    st.run = [=](hipStream_t q) {
      const int grid = (int)std::min<int64_t>(ceil_div((int64_t)I * F, kBlock), 4096);
      hipLaunchKernelGGL(expand_noise_kernel, dim3(grid, NB), dim3(kBlock), 0, q, stats, f2i, nz, zp, zp_tap, I, T, F, nsd, rngd, lensF);
      return PIPER_HIP_OK;
    };
    // path expansion counted as the reference's two MatMuls mm(1,F,192,T)
    st.flops = NB * 2.0 * 2.0 * F * (double)I * T;
    st.bytes = NB * 2.0 * 4.0 * ((double)F * T + (double)T * I + (double)F * I);
    s.steps.push_back(st);
  }
  s.taps["z_p"] = {zp_tap, I, F, 1, (size_t)I * F};
  // ---------------- flow (reverse)
  bool flipped = false;
  const int half = I / 2;
  for (int f = c.n_flows - 1; f >= 0; f--) {
    flipped = !flipped;
    const auto& C = v->flows[f];
    const std::string p = "flow" + std::to_string(f) + ".";
    // post of this coupling + x1 − m + Flip + pre of the next one in ONE launch (flow_seam.hip); PIPER_HIP_NO_FLOW_SEAM=1 keeps them apart
    static const bool no_seam = getenv("PIPER_HIP_NO_FLOW_SEAM") != nullptr;
    auto seam_ok = [&](const piper_hip_voice::Coupling& A, const piper_hip_voice::Coupling& B) {
      return !no_seam && I == 2 * half && flow_seam_eligible(H, half) && A.post.w16 && B.pre.w16 && A.post.bias && B.pre.bias && A.post.K == 1 && B.pre.K == 1 &&
             A.post.Cin == H && A.post.Cout == half && B.pre.Cin == half && B.pre.Cout == H;
    };
    const bool pre_done = f + 1 < c.n_flows && seam_ok(v->flows[f + 1], C);  // the previous (f + 1) coupling's seam already wrote h
    if (!pre_done) {
      ConvArgs a = plain(zp, h, I, H, F, lensF);
      a.in_ch_base = flipped ? I - 1 : 0;
      a.in_ch_sign = flipped ? -1 : 1;
      add_conv(v, s, p + "pre", C.pre, a, F);
    }
    for (int i = 0; i < c.wn_layers; i++) {
      const bool last = i + 1 == c.wn_layers;
      ConvArgs a = plain(h, acts, H, H, F, lensF);
      a.padL = (c.wn_kernel - 1) / 2;
      a.gate = 1;
      add_conv(v, s, p + "wn" + std::to_string(i) + ".in_gate", C.in[i], a, F);
      ConvArgs b = plain(acts, h, H, H, F, lensF);
      b.y2 = skip; b.y2_batch_stride = (int64_t)H * F;
      b.skip = i == 0 ? nullptr : skip;
      if (last) b.epilogue = EPI_WN_SKIP_LAST;
      else { b.epilogue = EPI_WN_RES_SKIP; b.wn_c = H; b.res = h; }
      add_conv(v, s, p + "wn" + std::to_string(i) + ".res_skip", C.rs[i], b, F);
    }
    if (f > 0 && seam_ok(C, v->flows[f - 1])) {
      const auto& Nx = v->flows[f - 1];
      Step st;
      st.name = p + "post_sub_flip_pre" + std::to_string(f - 1);
      st.tag = "conv_mfma";
      const float *p16 = C.post.w16, *pb = C.post.bias, *q16 = Nx.pre.w16, *qb = Nx.pre.bias;
      const int ps = (int)(packed_conv_floats(half, H, 1, 16) / ((size_t)ceil_div(half, 16) * 64));
      const int qs = (int)(packed_conv_floats(H, half, 1, 16) / ((size_t)ceil_div(H, 16) * 64));
      const int ob = flipped ? I - 1 - half : half, os = flipped ? -1 : 1;
      st.run = [=](hipStream_t q) { return launch_flow_seam(q, skip, zp, h, p16, pb, q16, qb, NB, H, half, F, ps, qs, ob, os, lensF); };
      st.flops = NB * (conv_flops(half, H, 1, F) + conv_flops(H, half, 1, F));
      st.bytes = NB * 4.0 * ((double)H * F + 2.0 * half * F + (double)H * F + 2.0 * half * H);
      s.steps.push_back(st);
    } else {
      ConvArgs a = plain(skip, zp, H, I, F, lensF);
      a.epilogue = EPI_RSUB;
      a.res = zp;
      a.out_ch_base = flipped ? I - 1 - half : half;
      a.out_ch_sign = flipped ? -1 : 1;
      add_conv(v, s, p + "post_sub", C.post, a, F);
    }
  }
  z = zp;
  if (flipped) {  // odd number of couplings: materialise the last Flip once
    Step st;
    st.name = "flow.final_flip";
    st.run = [=](hipStream_t q) {
      const int grid = (int)std::min<int64_t>(ceil_div((int64_t)I * F, kBlock), 4096);
      hipLaunchKernelGGL(flip_channels_kernel, dim3(grid, NB), dim3(kBlock), 0, q, zp, zflip, I, F);
      return PIPER_HIP_OK;
    };
    s.steps.push_back(st);
    z = zflip;
  }
  s.z_out = z;
  }  // !gen_only
  s.taps["z"] = {z, I, F, 1, (size_t)I * F};
  if (v->precision == PIPER_HIP_PRECISION_BF16) return build_generator_bf16(v, s, ar, z, dec0, F, NB);
  // ---------------- HiFi-GAN generator
  {
    ConvArgs a = plain(z, dec0, I, c.up_initial, F, lensF);
    a.padL = 3;
    add_conv(v, s, "dec.conv_pre", v->conv_pre, a, F);
  }
  s.taps["dec_pre"] = {dec0, c.up_initial, F, 1, (size_t)c.up_initial * F};
  static const bool no_merge = getenv("PIPER_HIP_NO_MERGED_RB") != nullptr;
  // Advancing the three ResBlocks in one launch pays while a single conv cannot fill the chip (short utterances, small
  // batches); with many tiles per conv (NB·F large) the per-conv schedule with the mean fused into its producer is faster
  // (measured at 8 × factor 8: 2 650 vs 2 840 utterances/s).
  // r2t: with two chained convs per launch (rb_pair.hip) the merged schedule is also the faster one for long rows;
  // PIPER_HIP_MERGED_MAX_F restores a frames × batch limit for A/B runs.
  static const int64_t merged_max = [] { const char* e = getenv("PIPER_HIP_MERGED_MAX_F"); return e ? (int64_t)atoll(e) : (int64_t)1 << 40; }();
  if (use_win && !no_merge && !parallel_rb && (int64_t)NB * F <= merged_max) {
    const size_t mark = s.steps.size();
    const int rcm = build_generator_merged(v, s, ar, dec0, F, NB);
    if (rcm != PIPER_HIP_ERR_UNSUPPORTED) return rcm;
    s.steps.resize(mark);  // geometry outside the window kernel: schedule conv by conv below
  }
  const float* cur[3] = {dec0, nullptr, nullptr};
  bool cur_is_mrf = false;
  int L = F;
  for (int u = 0; u < c.n_ups; u++) {
    const auto& S = v->stages[u];
    const int Lo = L * S.stride;  // (L−1)s − 2·pad + K = L·s for the even (K−s) the config check enforces
    float* up = ar.f32(B * S.Cout * Lo);
    float* r[PIPER_HIP_MAX_RB];
    float* tmp[PIPER_HIP_MAX_RB];
    float* tmp2[PIPER_HIP_MAX_RB];
    float* mid[PIPER_HIP_MAX_RB];
    for (int j = 0; j < c.n_rb; j++) {
      r[j] = ar.f32(B * S.Cout * Lo);
      tmp[j] = ar.f32(B * S.Cout * Lo);
      tmp2[j] = c.rb_n_dil > 2 ? ar.f32(B * S.Cout * Lo) : nullptr;
      mid[j] = c.resblock_type == 1 ? ar.f32(B * S.Cout * Lo) : nullptr;
    }
    if (ar.rc) return ar.rc;
    const std::string p = "dec.s" + std::to_string(u) + ".";
    {
      ConvArgs a;
      a.x = cur[0];
      a.prologue = cur_is_mrf ? PRO_NONE : PRO_LRELU;  // the MRF mean kernel already applied LeakyReLU(0.1)
      a.alpha = 0.1f;
      a.y = up; a.N = NB; a.dil = -1; a.padL = 0; a.Lin = L; a.Lout = (Lo - 1 + S.pad) / S.stride + 1;
      a.len_ptr = s.lensF; a.len_mul = L / F;
      a.x_batch_stride = (int64_t)S.Cin * L; a.y_batch_stride = (int64_t)S.Cout * Lo; a.y_len = Lo;
      a.epilogue = EPI_CONVT; a.ct_stride = S.stride; a.ct_padL = S.pad; a.ct_Lout = Lo;
      a.w = S.up.w; a.w16 = S.up.w16; a.bias = S.up.bias; a.Cin = S.up.Cin; a.Cout = S.up.Cout; a.K = S.up.K;
      Step st;
      st.name = p + "lrelu_convT";
      st.tag = "conv_mfma";
      const double ct_flops = NB * 2.0 * S.Cin * S.Cout * (double)S.K * L;
      const bool ct_pipe1 = use_win && pipe_pays1(ct_flops, S.Cin, true) && S.up.w5 && convt_pipe_eligible(S.Cin, S.Cout, S.K, S.stride, S.pad, L);
      if (ct_pipe1 || (use_win && S.up.w4 && convt_win_eligible(S.Cin, S.Cout, S.K, S.stride, S.pad, L))) {
        ConvWinArgs wa;
        wa.x = cur[0]; wa.w4 = ct_pipe1 ? S.up.w5 : S.up.w4; wa.bias = S.up.bias; wa.y = up;
        wa.pro_alpha = cur_is_mrf ? 1.0f : 0.1f;
        wa.N = NB; wa.Cin = S.Cin; wa.Cout = S.Cout; wa.K = S.K; wa.Lin = L; wa.Lout = L; wa.y_len = Lo;
        wa.ct_stride = S.stride; wa.ct_pad = S.pad;
        wa.len_ptr = s.lensF; wa.len_mul = L / F;
        if (ct_pipe1) st.run = [ctx, wa](hipStream_t q) { return launch_conv_pipe_multi(ctx, q, &wa, 1); };
        else st.run = [ctx, wa](hipStream_t q) { return launch_conv_win(ctx, q, wa); };
      } else
      st.run = [ctx, a](hipStream_t q) { return launch_conv_mfma(ctx, q, a); };
      st.flops = NB * 2.0 * S.Cin * S.Cout * (double)S.K * L;  // convT(Cin,Cout,K,s,Lin)
      st.bytes = NB * 4.0 * ((double)S.Cin * L + (double)S.Cout * Lo + (double)S.Cin * S.Cout * S.K + S.Cout);
      s.steps.push_back(st);
    }
    {  // the stage's ResBlocks read the same `up` and write disjoint buffers: run them as parallel graph branches
      Step f;
      f.name = p + "fork";
      f.kind = Step::FORK;
      s.steps.push_back(f);
    }
    float* m = ar.f32(B * S.Cout * Lo);  // lrelu(mean of the three ResBlock outputs): input of the next stage
    if (ar.rc) return ar.rc;
    const float mean_alpha = (u + 1 == c.n_ups) ? 0.01f : 0.1f;  // F.leaky_relu default slope before conv_post
    for (int j = 0; j < c.n_rb; j++) {
      s.cur_lane = j < 3 ? j : 0;
      const int K = c.rb_kernels[j];
      const float* src = up;
      for (int di = 0; di < c.rb_n_dil; di++) {
        const int dil = c.rb_dilations[j][di];
        const bool lastd = di + 1 == c.rb_n_dil;
        // the very last conv of the stage folds the MRF mean + LeakyReLU into its epilogue (r0, r1 are complete by then)
        const bool fuse_mean = lastd && j + 1 == c.n_rb && !parallel_rb;
        float* dst = lastd ? (fuse_mean ? m : r[j]) : ((di & 1) ? tmp2[j] : tmp[j]);
        const std::string nm = p + "rb" + std::to_string(j) + ".c" + std::to_string(di);
        auto rbconv = [&](const float* in, const float* res, float* out, int dl) {
          ConvArgs a = plain(in, out, S.Cout, S.Cout, Lo, lensF, Lo / F);
          a.dil = dl; a.padL = (K * dl - dl) / 2; a.prologue = PRO_LRELU; a.alpha = 0.1f; a.res = res;
          return a;
        };
        auto with_mean = [&](ConvArgs a) {
          if (fuse_mean) {
            a.epilogue = EPI_MRF_MEAN;
            a.mrf_a = r[0]; a.mrf_b = r[1]; a.alpha2 = mean_alpha;
          }
          return a;
        };
        // long rows: the window kernel (conv_win.hip); otherwise the streaming kernel
        auto add_rb = [&](const std::string& name, const ConvW& w, const ConvArgs& a) {
          if (!(use_win && w.w4 && conv_win_eligible(w.Cout, w.Cin, w.K, a.dil, a.padL, Lo, Lo))) {
            add_conv(v, s, name, w, a, Lo);
            return;
          }
          ConvWinArgs wa;
          wa.x = a.x; wa.w4 = w.w4; wa.bias = w.bias; wa.res = a.res; wa.y = a.y;
          wa.pro_alpha = a.alpha;
          if (a.epilogue == EPI_MRF_MEAN) { wa.mrf_a = a.mrf_a; wa.mrf_b = a.mrf_b; wa.out_alpha = a.alpha2; }
          wa.N = NB; wa.Cin = w.Cin; wa.Cout = w.Cout; wa.K = w.K; wa.dil = a.dil; wa.padL = a.padL;
          wa.Lin = Lo; wa.Lout = Lo; wa.y_len = Lo;
          wa.len_ptr = s.lensF; wa.len_mul = Lo / F;
          Step st;
          st.name = name;
          st.flops = NB * conv_flops(w.Cout, w.Cin, w.K, Lo);
          if (pipe_pays1(st.flops, w.Cin, false) && w.w5 && conv_pipe_eligible(w.Cout, w.Cin, w.K, a.dil, a.padL, Lo, Lo)) {
            wa.w4 = w.w5;
            st.run = [ctx, wa](hipStream_t q) { return launch_conv_pipe_multi(ctx, q, &wa, 1); };
          } else
          st.run = [ctx, wa](hipStream_t q) { return launch_conv_win(ctx, q, wa); };
          st.bytes = NB * conv_bytes(w.Cin, w.Cout, w.K, Lo);
          st.lane = s.cur_lane;
          st.tag = "conv_mfma";
          s.steps.push_back(std::move(st));
        };
        if (c.resblock_type == 1) {
          add_rb(nm + "a_lrelu_conv", S.rb[j][2 * di], rbconv(src, nullptr, mid[j], dil));
          add_rb(nm + (fuse_mean ? "b_lrelu_conv_res_mrfmean" : "b_lrelu_conv_res"), S.rb[j][2 * di + 1],
                 with_mean(rbconv(mid[j], src, dst, 1)));
        } else {
          add_rb(nm + (fuse_mean ? "_lrelu_conv_res_mrfmean" : "_lrelu_conv_res"), S.rb[j][di],
                 with_mean(rbconv(src, src, dst, dil)));
        }
        src = dst;
      }
    }
    s.cur_lane = 0;
    {
      Step jn;
      jn.name = p + "join";
      jn.kind = Step::JOIN;
      s.steps.push_back(jn);
    }
    if (parallel_rb) {  // branches finish independently: the mean needs its own launch after the join
      Step st;
      st.name = p + "mrf_mean_lrelu";
      const float *r0 = r[0], *r1 = r[1], *r2 = r[2];
      const int64_t cnt = (int64_t)NB * S.Cout * Lo;
      st.run = [=](hipStream_t q) {
        const int grid = (int)std::min<int64_t>(ceil_div(cnt, (int64_t)kBlock * 4), 2048);
        hipLaunchKernelGGL(mrf_mean_lrelu_kernel, dim3(grid), dim3(kBlock), 0, q, r0, r1, r2, m, cnt, mean_alpha);
        return PIPER_HIP_OK;
      };
      s.steps.push_back(st);
    }
    cur[0] = m; cur[1] = nullptr; cur[2] = nullptr;
    cur_is_mrf = true;
    L = Lo;
  }
  s.n_samples = L;
  s.audio = ar.f32(B * L);
  if (ar.rc) return ar.rc;
  {
    ConvArgs a;
    a.x = cur[0];
    a.prologue = PRO_NONE;  // LeakyReLU(0.01) of the MRF mean was applied by the mean kernel
    a.y = s.audio; a.N = NB; a.padL = 3; a.Lin = L; a.Lout = L;
    a.len_ptr = s.lensF; a.len_mul = L / F;
    a.x_batch_stride = (int64_t)v->conv_post.Cin * L; a.y_batch_stride = L; a.y_len = L;
    a.epilogue = EPI_TANH;
    add_conv(v, s, "dec.conv_post_tanh", v->conv_post, a, L);
  }
  return PIPER_HIP_OK;
}

int build_duration_predictor(piper_hip_voice* v, Slot& s, Arena& ar, const float* x, int T, int NB) {
  const piper_hip_voice_config& c = v->cfg;
  piper_hip_ctx* ctx = v->ctx;
  const int H = c.hidden, nbins = c.dp_bins, K = c.dp_kernel;
  const size_t B = (size_t)NB;
  if (!c.dp_present || v->dp.empty()) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "voice has no duration predictor (dp_present = 0): supply durations");
  if (!dds_layer_eligible(H, K)) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "duration predictor: hidden %d / kernel %d not covered", H, K);
  float* a0 = ar.f32(B * H * T);
  float* a1 = ar.f32(B * H * T);
  float* cond = ar.f32(B * H * T);                  // the predictor's conditioning x for the flows
  float* hsp = ar.f32(B * (size_t)(3 * nbins - 1) * T);
  float* z = ar.f32(B * 2 * T);
  float* logw = ar.f32(B * T);
  s.dp_noise = ar.f32(B * 2 * T);
  s.dp_scalars = ar.raw(dp_scalars_bytes(NB));
  s.dp_dur = (int32_t*)ar.raw(B * T * sizeof(int32_t));
  if (ar.rc) return ar.rc;
  const int* lensT = s.lensT;
  auto conv_k1 = [&](const std::string& name, const ConvW& w, const float* in, int64_t in_bs, float* out, const float* res) {
    ConvArgs a;
    a.x = in; a.y = out; a.N = NB; a.Lin = T; a.Lout = T; a.x_batch_stride = in_bs; a.y_batch_stride = (int64_t)w.Cout * T; a.y_len = T;
    a.len_ptr = lensT; a.res = res;
    add_conv(v, s, name, w, a, T);
  };
  auto dds_stack = [&](const std::string& name, const std::vector<piper_hip_voice::DdsLayer>& layers, float* cur, float* other) {
    // layer i: cur → other, then swap; returns where the result lives
    int dil = 1;
    for (size_t i = 0; i < layers.size(); i++) {
      const auto& L = layers[i];
      Step st;
      st.name = name + ".dds" + std::to_string(i);
      st.tag = "dds_layer";
      const float *src = cur, *dw_w = L.dw_w, *dw_b = L.dw_b, *g1 = L.g1, *b1 = L.b1, *pw = L.pw.w16, *pwb = L.pw.bias, *g2 = L.g2, *b2 = L.b2;
      float* dst = other;
      const int steps = (int)(packed_conv_floats(H, H, 1, 16) / ((size_t)ceil_div(H, 16) * 64));
      const int d2 = dil;
      st.run = [=](hipStream_t q) { return launch_dds_layer(ctx, q, src, dw_w, dw_b, g1, b1, pw, pwb, g2, b2, dst, NB, H, T, K, d2, steps, lensT, 1e-5f); };
      st.flops = NB * (conv_flops(H, H, 1, T) + conv_flops(H, 1, K, T));
      st.bytes = NB * 4.0 * (2.0 * H * T + (double)H * H);
      s.steps.push_back(st);
      std::swap(cur, other);
      dil *= K;
    }
    return cur;
  };
  // x → pre → DDSConv → proj = the conditioning of every flow
  conv_k1("dp.pre", v->dp[0].pre, x, (int64_t)H * T, a0, nullptr);
  float* r = dds_stack("dp", v->dp[0].dds, a0, a1);
  conv_k1("dp.proj", v->dp[0].proj, r, (int64_t)H * T, cond, nullptr);
  {
    Step st;
    st.name = "dp.init_latent";
    const float* nz = s.dp_noise;
    const void* sc = s.dp_scalars;
    st.run = [=](hipStream_t q) { return launch_dp_init(q, nz, sc, z, NB, T, lensT); };
    s.steps.push_back(st);
  }
  for (size_t b = 1; b < v->dp.size(); b++) {
    const auto& Bk = v->dp[b];
    const std::string p = "dp.flow" + std::to_string(Bk.flow);
    conv_k1(p + ".pre_add_cond", Bk.pre, z, (int64_t)2 * T, a0, cond);  // h = pre(z0) + g: the DDSConv's `x + g`
    float* hr = dds_stack(p, Bk.dds, a0, a1);
    conv_k1(p + ".proj", Bk.proj, hr, (int64_t)H * T, hsp, nullptr);
    Step st;
    st.name = p + ".spline_flip";
    const float tb = c.dp_tail_bound, fc = (float)H;
    st.run = [=](hipStream_t q) { return launch_dp_spline(q, hsp, z, NB, T, nbins, tb, fc, lensT); };
    s.steps.push_back(st);
  }
  {
    Step st;
    st.name = "dp.affine_exp_ceil";
    const float *m = v->dp_m, *lg = v->dp_logs;
    const void* sc = s.dp_scalars;
    int32_t* dur = s.dp_dur;
    st.run = [=](hipStream_t q) { return launch_dp_final(q, z, m, lg, sc, logw, dur, NB, T, lensT); };
    s.steps.push_back(st);
  }
  s.taps["logw"] = {logw, 1, T, 0, (size_t)T};
  return PIPER_HIP_OK;
}

int ensure_side_streams(Slot& s);

int run_schedule(Slot& s, hipStream_t q, bool parallel) {
  if (parallel) {
    const int rc0 = ensure_side_streams(s);
    if (rc0) return rc0;
  }
  for (auto& st : s.steps) {
    if (st.kind == Step::FORK) {
      if (parallel) {
        PH_HIP(hipEventRecord(s.ev_fork, q), PIPER_HIP_ERR_LAUNCH);
        for (int i = 0; i < 2; i++) PH_HIP(hipStreamWaitEvent(s.side[i], s.ev_fork, 0), PIPER_HIP_ERR_LAUNCH);
      }
      continue;
    }
    if (st.kind == Step::JOIN) {
      if (parallel)
        for (int i = 0; i < 2; i++) {
          PH_HIP(hipEventRecord(s.ev_join[i], s.side[i]), PIPER_HIP_ERR_LAUNCH);
          PH_HIP(hipStreamWaitEvent(q, s.ev_join[i], 0), PIPER_HIP_ERR_LAUNCH);
        }
      continue;
    }
    hipStream_t target = (parallel && st.lane > 0) ? s.side[st.lane - 1] : q;
    int rc = st.run(target);
    if (rc) return rc;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "schedule launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

// The first device → host copy a STREAM hands to the copy engine costs 7 … 17 ms (r3, tools/probe/first_run.py with PIPER_HIP_COLLECT_DMA=1:
// 8.1 ms in collect for 1.0 ms of GPU work; warming another stream of the process did not help). Every stream a plan will use gets that copy
// out of the way when it is created — while the voice loads for the four it pre-creates.
void warm_stream_copies(piper_hip_voice* v, hipStream_t q) {
  void* hp = nullptr;
  const size_t nb = std::min<size_t>((size_t)2 << 20, v->blob_floats * sizeof(float));  // large enough for the copy ENGINE (small ones are blitted)
  if (!v->blob || hipHostMalloc(&hp, nb) != hipSuccess) { (void)hipGetLastError(); return; }
  hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, q);  // as in a request: the copy waits for a kernel of the same stream
  (void)hipMemcpyAsync(hp, v->blob, nb, hipMemcpyDeviceToHost, q);
  (void)hipStreamSynchronize(q);
  (void)hipHostFree(hp);
  (void)hipGetLastError();
}

int slot_init(piper_hip_voice* v, Slot& s) {
  if (s.inited) return PIPER_HIP_OK;
  if (!v->free_sets.empty()) {  // a set an evicted plan left behind
    const auto st = v->free_sets.back();
    v->free_sets.pop_back();
    s.stream = st.stream; s.side[0] = st.side[0]; s.side[1] = st.side[1];
    s.ev0 = st.ev0; s.ev1 = st.ev1; s.ev_fork = st.ev_fork; s.ev_join[0] = st.ev_join[0]; s.ev_join[1] = st.ev_join[1];
    s.inited = true;
    return PIPER_HIP_OK;
  }
  // r3 (tools/probe/cold_prepare.py): creating the three streams of a plan was 8.6–10 ms of an 11 ms plan build. The two side
  // streams are only for schedules with parallel branches (ensure_side_streams), so a plan now creates one.
  PH_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipEventCreate(&s.ev0), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipEventCreate(&s.ev1), PIPER_HIP_ERR_LAUNCH);
  warm_stream_copies(v, s.stream);
  s.inited = true;
  return PIPER_HIP_OK;
}

int ensure_side_streams(Slot& s) {
  for (int i = 0; i < 2; i++) {
    if (!s.side[i]) PH_HIP(hipStreamCreateWithFlags(&s.side[i], hipStreamNonBlocking), PIPER_HIP_ERR_LAUNCH);
    if (!s.ev_join[i]) PH_HIP(hipEventCreateWithFlags(&s.ev_join[i], hipEventDisableTiming), PIPER_HIP_ERR_LAUNCH);
  }
  if (!s.ev_fork) PH_HIP(hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming), PIPER_HIP_ERR_LAUNCH);
  return PIPER_HIP_OK;
}

int check_utt(const piper_hip_voice* v, const piper_hip_utterance* u, int64_t* F_out) {
  if (!v) PH_FAIL(PIPER_HIP_ERR_ARG, "null voice");
  if (!u || !u->phoneme_ids) PH_FAIL(PIPER_HIP_ERR_ARG, "utterance: null ids");
  if (u->t < 1) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance: need at least one phoneme id");
  if (u->t > 4096) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance: %d ids exceeds the 4096 cap (PiperCLI.swift:394)", u->t);
  if (!u->durations) {  // to be predicted: the frame count is not known yet
    if (!v->cfg.dp_present) PH_FAIL(PIPER_HIP_ERR_ARG, "utterance: durations are NULL and the voice has no duration predictor");
    // `noise` is [inter, F] and F is what the predictor is about to decide: the caller cannot have sized it, and the ABI carries no
    // size to check it against (in the reference an override tensor brings its shape, TensorValue.swift:4-43)
    if (u->noise)
      PH_FAIL(PIPER_HIP_ERR_ARG, "utterance: noise given but durations are NULL — its [inter, F] shape depends on the predicted durations: call "
                                 "piper_hip_voice_predict_durations first and pass durations + noise, or use noise_mode = DEVICE");
    *F_out = -1;
    return PIPER_HIP_OK;
  }
  int64_t F = 0;
  for (int i = 0; i < u->t; i++) {
    if (u->durations[i] < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance: negative duration");
    F += u->durations[i];
  }
  if (F < 1) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance: zero frames");
  if (F * v->hop > 0x3fffffff / 64) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance: %lld frames too long", (long long)F);
  *F_out = F;
  return PIPER_HIP_OK;
}

}  // namespace

PH_EXPORT int piper_hip_voice_create(piper_hip_ctx* ctx, const piper_hip_voice_config* cfg, const float* blob, int on_device,
                                     piper_hip_voice** out) {
  PH_CHECK_CTX(ctx);
  if (!out || !blob) PH_FAIL(PIPER_HIP_ERR_ARG, "voice_create: null argument");
  *out = nullptr;
  int rc = validate_config(cfg);
  if (rc) return rc;
  if (cfg->hidden % 32 || cfg->inter % 64) PH_FAIL(PIPER_HIP_ERR_SHAPE, "voice: hidden must be a multiple of 32 and inter of 64");
  PH_HIP(hipSetDevice(ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  std::unique_ptr<piper_hip_voice> v(new piper_hip_voice());
  v->ctx = ctx;
  v->cfg = *cfg;
  v->hop = 1;
  for (int u = 0; u < cfg->n_ups; u++) v->hop *= cfg->up_rates[u];
  IndexOut io{&v->index};
  v->blob_floats = piper_hip_layout_walk(cfg, index_visit, &io);
  void* p = nullptr;
  if ((rc = ctx->pool.alloc(v->blob_floats * sizeof(float), &p))) return rc;
  v->blob = (float*)p;
  v->owned.push_back(p);
  hipStream_t s = ctx->default_stream;
  PH_HIP(hipMemcpyAsync(v->blob, blob, v->blob_floats * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s),
         PIPER_HIP_ERR_LAUNCH);
  // measure, allocate, pack
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(s);  // pack kernels / the blob copy may still be running on these blocks
    for (void* q : v->owned) (void)ctx->pool.release(q);
    return code;
  };
  Packer dry{v.get(), 0, s};
  compile_weights(v.get(), dry, true, nullptr);
  v->packed_floats = dry.off;
  if ((rc = ctx->pool.alloc(v->packed_floats * sizeof(float), &p))) return fail(rc);
  v->packed = (float*)p;
  v->owned.push_back(p);
  std::vector<float*> qkvb;
  for (int l = 0; l < cfg->n_layers; l++) {
    if ((rc = ctx->pool.alloc((size_t)3 * cfg->hidden * sizeof(float), &p))) return fail(rc);
    v->owned.push_back(p);
    qkvb.push_back((float*)p);
  }
  Packer pk{v.get(), 0, s};
  if ((rc = compile_weights(v.get(), pk, false, &qkvb))) return fail(rc);
  hipError_t e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    fail(0);
    PH_FAIL(PIPER_HIP_ERR_LAUNCH, "voice_create: packing failed: %s", hipGetErrorString(e));
  }
  ph::warm_all_modules();  // every translation unit's code object, now instead of under the first request that needs it (common.h)
  // plan memory up front (Pool::reserve): a no-op when the host reserved its own amount before creating the voice
  {
    static const size_t reserve_mb = [] { const char* e = getenv("PIPER_HIP_RESERVE_MB"); return e ? (size_t)atoll(e) : (size_t)8192; }();
    (void)ctx->pool.reserve(reserve_mb << 20);
  }
  // Four stream sets up front: creating a HIP stream costs ≈ 3 ms (8 ms for the first of a process, tools/probe/cold_prepare.py) — paid here,
  // while the voice loads, it is off the first request (first_request_ms 21 → 13 in bench.py). Plans take sets from this list (slot_init)
  // and give them back when they go idle (detach); a voice serving more than four slot ids at once creates the rest on demand.
  for (int i = 0; i < 4; i++) {
    piper_hip_voice::StreamSet st = {};
    if (hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&st.ev0) != hipSuccess || hipEventCreate(&st.ev1) != hipSuccess) {
      (void)hipGetLastError();
      if (st.ev0) (void)hipEventDestroy(st.ev0);
      if (st.stream) (void)hipStreamDestroy(st.stream);
      break;  // not fatal: slot_init creates what is missing
    }
    v->free_sets.push_back(st);
  }
  // … and the process's first graph capture + instantiate (≈ 8 ms against ≈ 1 ms for later ones: the runtime sets its graph machinery up
  // on first use) on a one-kernel graph, for the same reason. Failure is not fatal.
  if (!v->free_sets.empty()) {
    const hipStream_t q = v->free_sets[0].stream;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    if (hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, q);
      if (hipStreamEndCapture(q, &g) == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess && ge) {
        (void)hipGraphLaunch(ge, q);
        (void)hipStreamSynchronize(q);
      }
    }
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
  }
  for (auto& st : v->free_sets) warm_stream_copies(v.get(), st.stream);
  *out = v.release();
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_set_precision(piper_hip_voice* v, int precision) {
  if (!v) PH_FAIL(PIPER_HIP_ERR_ARG, "null voice");
  if (precision != PIPER_HIP_PRECISION_F32 && precision != PIPER_HIP_PRECISION_BF16)
    PH_FAIL(PIPER_HIP_ERR_ARG, "voice_set_precision: unknown precision %d", precision);
  if (precision == v->precision) return PIPER_HIP_OK;
  piper_hip_ctx* ctx = v->ctx;
  const piper_hip_voice_config& c = v->cfg;
  PH_HIP(hipSetDevice(ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  PH_HIP(hipDeviceSynchronize(), PIPER_HIP_ERR_LAUNCH);
  if (precision == PIPER_HIP_PRECISION_BF16 && v->up_b.empty()) {
    // every generator conv must fall inside the bf16 kernels' geometry; otherwise the voice stays fp32
    if (!conv_bf16_eligible(c.up_initial, c.inter, 7, 1, 3, 3))
      PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "bf16 generator: conv_pre %d→%d not covered", c.inter, c.up_initial);
    size_t elems = packed_conv_bf16_elems(c.up_initial, c.inter, 7);
    for (const auto& S : v->stages) {
      if (!convt_bf16_eligible(S.Cin, S.Cout, S.K, S.stride, S.pad, S.pad, 1, 0))
        PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "bf16 generator: ConvTranspose %d→%d k%d s%d not covered", S.Cin, S.Cout, S.K, S.stride);
      elems += packed_convt_bf16_elems(S.Cin, S.Cout, S.K, S.stride);
      for (int j = 0; j < c.n_rb; j++)
        for (int di = 0; di < c.rb_n_dil; di++) {
          const int K = c.rb_kernels[j], dl = c.rb_dilations[j][di], pad = (K * dl - dl) / 2;
          if (!conv_bf16_eligible(S.Cout, S.Cout, K, dl, pad, pad))
            PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "bf16 generator: ResBlock conv C=%d k%d d%d not covered", S.Cout, K, dl);
          elems += (c.resblock_type == 1 ? 2 : 1) * packed_conv_bf16_elems(S.Cout, S.Cout, K);
        }
    }
    void* p = nullptr;
    int rc = ctx->pool.alloc(elems * 2 + 256 * (size_t)(2 + c.n_ups * (1 + c.n_rb * c.rb_n_dil * 2)), &p);
    if (rc) return rc;
    v->owned.push_back(p);
    uint16_t* cur = (uint16_t*)p;
    hipStream_t q = ctx->default_stream;
    auto take = [&](size_t n) { uint16_t* r = cur; cur += (n + 127) & ~(size_t)127; return r; };
    auto conv = [&](const std::string& prefix, int Cout, int Cin, int K) {
      piper_hip_voice::ConvWB w;
      w.Cout = Cout; w.Cin = Cin; w.K = K;
      uint16_t* img = take(packed_conv_bf16_elems(Cout, Cin, K));
      pack_conv_weights_bf16(q, tensor(v, prefix + ".weight"), Cout, Cin, K, img);
      w.w = img;
      w.bias = tensor(v, prefix + ".bias");
      return w;
    };
    v->conv_pre_b = conv("dec.conv_pre", c.up_initial, c.inter, 7);
    char nm[128];
    for (int u = 0; u < c.n_ups; u++) {
      const auto& S = v->stages[u];
      piper_hip_voice::ConvWB w;
      w.Cout = S.Cout; w.Cin = S.Cin; w.K = S.K;
      uint16_t* img = take(packed_convt_bf16_elems(S.Cin, S.Cout, S.K, S.stride));
      snprintf(nm, sizeof nm, "dec.ups.%d", u);
      pack_convt_weights_bf16(q, tensor(v, std::string(nm) + ".weight"), S.Cin, S.Cout, S.K, S.stride, S.pad, img);
      w.w = img;
      w.bias = tensor(v, std::string(nm) + ".bias");
      v->up_b.push_back(w);
      std::vector<std::vector<piper_hip_voice::ConvWB>> rbs;
      for (int j = 0; j < c.n_rb; j++) {
        std::vector<piper_hip_voice::ConvWB> convs;
        const int rb = u * c.n_rb + j;
        for (int di = 0; di < c.rb_n_dil; di++) {
          if (c.resblock_type == 1) {
            snprintf(nm, sizeof nm, "dec.resblocks.%d.convs1.%d", rb, di);
            convs.push_back(conv(nm, S.Cout, S.Cout, c.rb_kernels[j]));
            snprintf(nm, sizeof nm, "dec.resblocks.%d.convs2.%d", rb, di);
            convs.push_back(conv(nm, S.Cout, S.Cout, c.rb_kernels[j]));
          } else {
            snprintf(nm, sizeof nm, "dec.resblocks.%d.convs.%d", rb, di);
            convs.push_back(conv(nm, S.Cout, S.Cout, c.rb_kernels[j]));
          }
        }
        rbs.push_back(convs);
      }
      v->rb_b.push_back(rbs);
    }
    hipError_t e = hipStreamSynchronize(q);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "voice_set_precision: packing failed: %s", hipGetErrorString(e));
  }
  for (auto& pl : v->plans) slot_release(v, *pl, true);  // every plan was built for the old precision: the next prepare rebuilds
  v->plans.clear();
  for (auto& at : v->attached) at = nullptr;
  v->precision = precision;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_precision(const piper_hip_voice* v) { return v ? v->precision : -1; }

PH_EXPORT void piper_hip_voice_destroy(piper_hip_voice* v) {
  if (!v) return;
  (void)hipSetDevice(v->ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& pl : v->plans) slot_release(v, *pl, true);
  for (auto& sg : v->staging) {
    if (sg.h_ids) (void)hipHostFree(sg.h_ids);
    if (sg.h_f2i) (void)hipHostFree(sg.h_f2i);
    if (sg.h_lens) (void)hipHostFree(sg.h_lens);
    if (sg.h_audio) (void)hipHostFree(sg.h_audio);
    if (sg.h_misc) (void)hipHostFree(sg.h_misc);
    if (sg.h_dpn) (void)hipHostFree(sg.h_dpn);
    if (sg.h_res) (void)hipHostFree(sg.h_res);
    if (sg.h_noise) (void)hipHostFree(sg.h_noise);
    for (hipEvent_t e : sg.chunk_ev) (void)hipEventDestroy(e);
  }
  for (auto& st : v->free_sets) {
    for (hipEvent_t e : {st.ev0, st.ev1, st.ev_fork, st.ev_join[0], st.ev_join[1]})
      if (e) (void)hipEventDestroy(e);
    for (hipStream_t q : {st.side[0], st.side[1], st.stream})
      if (q) (void)hipStreamDestroy(q);
  }
  for (void* p : v->owned) (void)v->ctx->pool.release(p);
  delete v;
}

PH_EXPORT int64_t piper_hip_voice_num_samples(const piper_hip_voice* v, const piper_hip_utterance* u) {
  int64_t F = 0;
  if (check_utt(v, u, &F)) return -1;
  if (F < 0) return -2;
  return F * v->hop;
}

namespace {

// Buckets: phoneme rows in steps of 16 (the tile width of the short-row kernels), frame rows in steps of 16 up to 1024 frames
// and 64 beyond (long utterances: ≤ 6 % padding, far fewer distinct graphs).
int bucket_t(int T) { return (int)ceil_div(T, 16) * 16; }
int bucket_f(int F) { return F <= 1024 ? (int)ceil_div(F, 16) * 16 : (int)ceil_div(F, 64) * 64; }

constexpr size_t kPlanCacheMax = 128;                   // default plans kept per voice (r3: 48 → 128; an idle plan holds an arena, no stream) …
constexpr size_t kPlanCacheBytes = (size_t)24 << 30;    // … and arena bytes (of 288 GB): least recently used idle plans go first
// (piper_hip_voice_set_plan_cache changes both per voice)

Slot* slot_plan(const piper_hip_voice* v, int slot) {
  if (!v || slot < 0 || slot >= kMaxSlots) return nullptr;
  Slot* p = v->attached[slot];
  return (p && p->built) ? p : nullptr;
}

void evict_idle_plans(piper_hip_voice* v) {
  auto total = [&]() { size_t b = 0; for (auto& p : v->plans) b += p->arena_bytes; return b; };
  while (v->plans.size() > v->plan_cache_max || total() > v->plan_cache_bytes) {
    int victim = -1;
    for (int i = 0; i < (int)v->plans.size(); i++)
      if (!v->plans[i]->in_use && (victim < 0 || v->plans[i]->last_use < v->plans[victim]->last_use)) victim = i;
    if (victim < 0) return;  // everything is attached: nothing to evict
    Slot& d = *v->plans[victim];
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    slot_release(v, d, true);
    v->plans.erase(v->plans.begin() + victim);
  }
}

// Run a plan once on its stream. A plan that has a graph replays it; a freshly built one runs its schedule eagerly (the answer of
// the request that missed the cache) and captures + instantiates the graph behind that run, so that the next request of the bucket
// replays. The capture enqueues nothing; it and the instantiate are host work that overlaps the eager pass on the GPU.
// `on`: the stream the plan runs on (default: its own). A plan without parallel lanes may run on any stream — the bounded prepare puts
// the predictor plan on the stream of the plan that continues from it, so the two need no event between them.
int launch_plan(piper_hip_voice* v, Slot& s, hipStream_t on = nullptr) {
  const hipStream_t q = on ? on : s.stream;
  if (s.exec) {
    PH_HIP(hipGraphLaunch(s.exec, q), PIPER_HIP_ERR_LAUNCH);
    return PIPER_HIP_OK;
  }
  int rc = run_schedule(s, q, false);
  if (rc) return rc;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  hipError_t e = hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "plan: begin capture failed: %s", hipGetErrorString(e));
  // fp32 generator: parallel ResBlock branches measured SLOWER on hipGraph (fork/join edges cost more than the three
  // short kernels gain), so that graph stays a single chain unless asked otherwise. The bf16 generator's builder decides
  // for itself (s.parallel).
  rc = run_schedule(s, q, s.parallel && !on);
  hipError_t ce = hipStreamEndCapture(q, &s.graph);
  if (rc) return rc;
  if (ce != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "plan: graph capture failed: %s", hipGetErrorString(ce));
  const auto t1 = clk::now();
  ce = hipGraphInstantiate(&s.exec, s.graph, nullptr, nullptr, 0);
  if (ce != hipSuccess) { s.exec = nullptr; PH_FAIL(PIPER_HIP_ERR_LAUNCH, "plan: graph instantiate failed: %s", hipGetErrorString(ce)); }
  v->last_build_ms[4] = std::chrono::duration<double, std::milli>(t1 - t0).count();
  v->last_build_ms[5] = std::chrono::duration<double, std::milli>(clk::now() - t1).count();
  return PIPER_HIP_OK;
}

// An idle plan for (kind, Tb, Fb, NB) at the voice's precision, built (schedule + arena; its graph follows its first run) if
// the cache has none. *built reports whether this call paid for a build ("cold" prepare).
int acquire_plan(piper_hip_voice* v, int kind, int Tb, int Fb, int NB, Slot** out, bool* built) {
  *built = false;
  for (auto& p : v->plans)
    if (!p->in_use && p->built && p->kind == kind && p->T == Tb && p->F == Fb && p->NB == NB && p->prec == v->precision) {
      const int rc0 = slot_init(v, *p);  // a plan that went idle gave its stream set back (detach)
      if (rc0) return rc0;
      *out = p.get();
      return PIPER_HIP_OK;
    }
  std::unique_ptr<Slot> np(new Slot());
  using clk = std::chrono::steady_clock;
  auto t_prev = clk::now();
  int phase = 0;
  auto lap = [&]() {  // wall time of the build's phases → piper_hip_voice_last_build_breakdown
    const auto now = clk::now();
    if (phase < 6) v->last_build_ms[phase++] = std::chrono::duration<double, std::milli>(now - t_prev).count();
    t_prev = now;
  };
  int rc = slot_init(v, *np);
  if (rc) { slot_release(v, *np, true); return rc; }
  lap();
  if ((rc = build_schedule(v, *np, Tb, Fb, NB, kind))) { slot_release(v, *np, true); return rc; }
  lap();
  Slot& s = *np;
  // a plan may be captured / profiled before every input has been uploaded: give the length arrays and index inputs legal values
  {
    hipLaunchKernelGGL(fill_lens_kernel, dim3((unsigned)ceil_div(NB, 64)), dim3(64), 0, s.stream, s.lensT, s.lensF, Tb, Fb, NB);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && s.ids) e = hipMemsetAsync(s.ids, 0, (size_t)NB * Tb * sizeof(int64_t), s.stream);
    if (e == hipSuccess && s.frame2id) e = hipMemsetAsync(s.frame2id, 0, (size_t)NB * Fb * sizeof(int32_t), s.stream);
    if (e == hipSuccess && s.rng) e = hipMemsetAsync(s.rng, 0, (size_t)NB * 2 * sizeof(unsigned), s.stream);
    if (e == hipSuccess && s.dp_scalars) e = hipMemsetAsync(s.dp_scalars, 0, dp_scalars_bytes(NB), s.stream);
    if (e == hipSuccess && s.dp_noise) e = hipMemsetAsync(s.dp_noise, 0, (size_t)NB * 2 * Tb * sizeof(float), s.stream);
    if (e == hipSuccess) e = stream_wait(s.stream);
    if (e != hipSuccess) { slot_release(v, s, true); PH_FAIL(PIPER_HIP_ERR_LAUNCH, "voice_prepare: arena initialisation failed: %s", hipGetErrorString(e)); }
  }
  lap();
  // No validation pass and no capture here (until round 3 a build ran the schedule once eagerly, waited for it, captured it and
  // instantiated the graph before the caller's inputs were even uploaded: 1–6 ms of GPU time + 0.3 ms on the critical path of the
  // first request of a bucket). The plan's FIRST launch goes out eagerly — that run IS the request's answer — and the graph is
  // captured and instantiated right behind it, host work that overlaps the GPU's (launch_plan).
  s.built = true;
  lap(); lap(); lap();
  *out = np.get();
  v->plans.push_back(std::move(np));
  *built = true;
  return PIPER_HIP_OK;
}

void detach(piper_hip_voice* v, int slot) {
  Slot* p = v->attached[slot];
  if (!p) return;
  if (p->stream) (void)hipStreamSynchronize(p->stream);  // its last launch may still be running / reading the inputs
  p->in_use = false;
  p->st_next = -1;
  p->bounded_pending = false;
  v->attached[slot] = nullptr;
  if (v->attached_dp[slot]) {  // ran on p's stream (bounded prepare): idle now
    v->attached_dp[slot]->in_use = false;
    v->attached_dp[slot] = nullptr;
  }
  // An idle plan needs no stream: hand the set to the next plan that is attached (usually the one replacing this plan on the same
  // slot id), so that after a slot id's first request no prepare ever creates a stream again (≈ 3 ms each, r3).
  if (p->inited && p->stream) {
    v->free_sets.push_back({p->stream, {p->side[0], p->side[1]}, p->ev0, p->ev1, p->ev_fork, {p->ev_join[0], p->ev_join[1]}});
    p->stream = nullptr; p->side[0] = p->side[1] = nullptr; p->ev0 = p->ev1 = p->ev_fork = nullptr; p->ev_join[0] = p->ev_join[1] = nullptr;
    p->inited = false;
  }
}

}  // namespace

namespace {
int predict_impl(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int32_t* durations_out, float* logw_out, int max_entries, Slot** plan_out);
template <typename Tp>
int grow_pinned(Tp*& p, size_t& cap, size_t need) {
  if (cap >= need) return PIPER_HIP_OK;
  if (p) (void)hipHostFree(p);
  p = nullptr; cap = 0;
  size_t c = 1024;
  while (c < need) c <<= 1;
  PH_HIP(hipHostMalloc((void**)&p, c * sizeof(Tp)), PIPER_HIP_ERR_ALLOC);
  cap = c;
  return PIPER_HIP_OK;
}
}

PH_EXPORT int piper_hip_voice_prepare_batch(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int slot) {
  if (!v || !utts) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (n < 1 || n > 256) PH_FAIL(PIPER_HIP_ERR_SHAPE, "batch size %d outside [1,256]", n);
  if (slot < 0 || slot >= kMaxSlots) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d out of range [0,%d)", slot, kMaxSlots);
  // the batch items may differ in length: the plan is the bucket of the longest, each item carries its own true lengths
  int Tmax = 0, rc;
  int64_t Fmax = 0;
  std::vector<int> hT(n), hF(n);
  // utterances without durations: run the text encoder + duration predictor first (its own cached plan), then continue with
  // the predicted frames per id exactly as if the caller had supplied them
  std::vector<piper_hip_utterance> resolved;
  std::vector<int32_t> predicted;
  Slot* dp_plan = nullptr;  // the encoder + predictor plan whose m_p / logs_p the main plan continues from (kind 3: no second encoder pass)
  struct DpRelease {
    Slot*& p;
    ~DpRelease() { if (p) p->in_use = false; }
  } dp_release{dp_plan};
  {
    bool any_null = false;
    for (int b = 0; b < n; b++) any_null = any_null || (utts[b].phoneme_ids && !utts[b].durations);
    if (any_null) {
      int64_t total = 0;
      for (int b = 0; b < n; b++) {
        if (utts[b].t < 1 || utts[b].t > 4096) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance %d: bad phoneme count", b);
        if (!utts[b].durations && utts[b].noise)  // the refusal check_utt makes: here the durations are still NULL, below they are the predicted ones
          PH_FAIL(PIPER_HIP_ERR_ARG, "utterance %d: noise given but durations are NULL — its [inter, F] shape depends on the predicted durations: call "
                                     "piper_hip_voice_predict_durations first and pass durations + noise, or use noise_mode = DEVICE", b);
        total += utts[b].t;
      }
      predicted.resize((size_t)total);
      if ((rc = predict_impl(v, utts, n, predicted.data(), nullptr, (int)total, &dp_plan))) return rc;
      resolved.assign(utts, utts + n);
      int64_t off = 0;
      for (int b = 0; b < n; b++) {
        if (!resolved[b].durations) {
          int64_t F = 0;
          for (int t = 0; t < utts[b].t; t++) F += predicted[off + t];
          if (F < 1) predicted[off] = 1;  // Piper: y_lengths = clamp_min(Σ w_ceil, 1)
          resolved[b].durations = predicted.data() + off;
        }
        off += utts[b].t;
      }
      utts = resolved.data();
    }
  }
  for (int b = 0; b < n; b++) {
    int64_t Fb = 0;
    if ((rc = check_utt(v, &utts[b], &Fb))) return rc;
    if (utts[b].noise_mode != PIPER_HIP_NOISE_INJECTED && utts[b].noise_mode != PIPER_HIP_NOISE_DEVICE)
      PH_FAIL(PIPER_HIP_ERR_ARG, "utterance %d: unknown noise_mode %d", b, utts[b].noise_mode);
    hT[b] = utts[b].t;
    hF[b] = (int)Fb;
    Tmax = std::max(Tmax, utts[b].t);
    Fmax = std::max(Fmax, Fb);
  }
  const int T = bucket_t(Tmax), F = bucket_f((int)Fmax), I = v->cfg.inter;
  if ((int64_t)F * v->hop * n > 0x3fffffff) PH_FAIL(PIPER_HIP_ERR_SHAPE, "batch too large");
  PH_HIP(hipSetDevice(v->ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  Slot* cur = v->attached[slot];
  const int kind = (dp_plan && dp_plan->stats && dp_plan->T == T && dp_plan->NB == n) ? 3 : 0;
  const bool same = cur && cur->built && cur->kind == kind && cur->T == T && cur->F == F && cur->NB == n && cur->prec == v->precision;
  if (cur && !same) detach(v, slot);
  if (!same) {
    bool built = false;
    if ((rc = acquire_plan(v, kind, T, F, n, &cur, &built))) return rc;
    cur->in_use = true;
    v->attached[slot] = cur;
    evict_idle_plans(v);
  }
  Slot& s = *cur;
  s.last_use = ++v->use_clock;
  // the previous launch on this plan may still be reading the inputs
  PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
  s.bounded_pending = false;
  if (v->attached_dp[slot]) { v->attached_dp[slot]->in_use = false; v->attached_dp[slot] = nullptr; }
  if (kind == 3)  // the predictor's plan has finished (predict synchronises): its projection becomes this plan's input
  {
    PH_HIP(hipMemcpyAsync(s.stats, dp_plan->stats, (size_t)n * 2 * I * T * sizeof(float), hipMemcpyDeviceToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
    // the predictor plan goes back to the cache when this function returns: whatever runs on it next must not overwrite the
    // projection before this copy has read it
    if (!s.ev_in) PH_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming), PIPER_HIP_ERR_LAUNCH);
    PH_HIP(hipEventRecord(s.ev_in, s.stream), PIPER_HIP_ERR_LAUNCH);
    PH_HIP(hipStreamWaitEvent(dp_plan->stream, s.ev_in, 0), PIPER_HIP_ERR_LAUNCH);
  }
  s.st_next = -1;
  s.h_T = hT;
  s.h_F = hF;
  {
    auto& sg = v->staging[slot];
    auto grow = [](size_t need) { size_t c = 1024; while (c < need) c <<= 1; return c; };
    if (sg.cap_t < (size_t)T * n) {
      if (sg.h_ids) (void)hipHostFree(sg.h_ids);
      sg.h_ids = nullptr; sg.cap_t = 0;
      const size_t c = grow((size_t)T * n);
      PH_HIP(hipHostMalloc((void**)&sg.h_ids, c * sizeof(int64_t)), PIPER_HIP_ERR_ALLOC);
      sg.cap_t = c;
    }
    if (sg.cap_f < (size_t)F * n) {
      if (sg.h_f2i) (void)hipHostFree(sg.h_f2i);
      sg.h_f2i = nullptr; sg.cap_f = 0;
      const size_t c = grow((size_t)F * n);
      PH_HIP(hipHostMalloc((void**)&sg.h_f2i, c * sizeof(int32_t)), PIPER_HIP_ERR_ALLOC);
      sg.cap_f = c;
    }
    if (sg.cap_lens < (size_t)2 * n) {
      if (sg.h_lens) (void)hipHostFree(sg.h_lens);
      sg.h_lens = nullptr; sg.cap_lens = 0;
      const size_t c = grow((size_t)2 * n);
      PH_HIP(hipHostMalloc((void**)&sg.h_lens, c * sizeof(int)), PIPER_HIP_ERR_ALLOC);
      sg.cap_lens = c;
    }
    s.h_ids = sg.h_ids; s.h_f2i = sg.h_f2i; s.h_lens = sg.h_lens;  // borrowed for this request (the plan's stream was synchronised above)
  }
  s.h_noise_scale.resize(n);
  s.h_rng.resize(2 * (size_t)n);
  s.h_dur.clear();
  // Everything that goes to the device leaves from page-locked memory of the slot id: an asynchronous copy from PAGEABLE memory makes the
  // runtime pin (or stage) the caller's pages on the spot. The noise tensor is copied into bucket rows here, on the host.
  auto& sgp = v->staging[slot];
  {
    size_t noise_floats = 0;
    for (int b = 0; b < n; b++) if (utts[b].noise) noise_floats = (size_t)n * I * F;
    if (noise_floats && (rc = grow_pinned(sgp.h_noise, sgp.cap_noise, noise_floats))) return rc;
    if ((rc = grow_pinned(sgp.h_misc, sgp.cap_misc, (size_t)n * (2 * sizeof(unsigned) + sizeof(float)) + 64))) return rc;
  }
  unsigned* p_rng = (unsigned*)sgp.h_misc;
  float* p_ns = (float*)(p_rng + 2 * (size_t)n);
  for (int b = 0; b < n; b++) {
    const piper_hip_utterance* u = &utts[b];
    const int Tb = hT[b], Fb = hF[b];
    s.h_dur.insert(s.h_dur.end(), u->durations, u->durations + Tb);
    s.h_lens[b] = Tb;
    s.h_lens[n + b] = Fb;
    s.h_rng[2 * b] = (!u->noise && u->noise_mode == PIPER_HIP_NOISE_DEVICE) ? 1u : 0u;
    s.h_rng[2 * b + 1] = u->seed;
    memcpy(s.h_ids + (size_t)b * T, u->phoneme_ids, (size_t)Tb * sizeof(int64_t));
    for (int t = Tb; t < T; t++) s.h_ids[(size_t)b * T + t] = 0;  // rows of the bucket beyond the utterance: any legal id
    int f = 0;  // generate_path: frame f belongs to the phoneme whose cumulative duration covers it
    for (int t = 0; t < Tb; t++)
      for (int j = 0; j < u->durations[t]; j++) s.h_f2i[(size_t)b * F + f++] = t;
    for (; f < F; f++) s.h_f2i[(size_t)b * F + f] = 0;
    s.h_noise_scale[b] = u->noise_scale;
    // noise [I, Fb] → rows of the bucket [I, F]
    p_rng[2 * b] = s.h_rng[2 * b]; p_rng[2 * b + 1] = s.h_rng[2 * b + 1];
    p_ns[b] = u->noise_scale;
    if (u->noise) {
      float* hb = sgp.h_noise + (size_t)b * I * F;
      for (int c = 0; c < I; c++) {
        memcpy(hb + (size_t)c * F, u->noise + (size_t)c * Fb, (size_t)Fb * sizeof(float));
        if (Fb < F) memset(hb + (size_t)c * F + Fb, 0, (size_t)(F - Fb) * sizeof(float));
      }
      PH_HIP(hipMemcpyAsync(s.noise + (size_t)b * I * F, hb, (size_t)I * F * sizeof(float), hipMemcpyHostToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
    } else if (!s.h_rng[2 * b])
      PH_HIP(hipMemsetAsync(s.noise + (size_t)b * I * F, 0, (size_t)I * F * sizeof(float), s.stream), PIPER_HIP_ERR_LAUNCH);
  }
  PH_HIP(hipMemcpyAsync(s.rng, p_rng, 2 * (size_t)n * sizeof(unsigned), hipMemcpyHostToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.noise_scale, p_ns, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.ids, s.h_ids, (size_t)T * n * sizeof(int64_t), hipMemcpyHostToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.frame2id, s.h_f2i, (size_t)F * n * sizeof(int32_t), hipMemcpyHostToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.lensT, s.h_lens, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.lensF, s.h_lens + n, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);  // noise / scalars come from caller memory
  s.timed = false;
  return slot;
}

namespace {
// generate_path on the device (GraphExecutor runs the exported graph's CumSum / Less / Cast chain): per item, frames per id → the frame
// count (Piper: clamp_min(Σ w_ceil, 1)), clamped to the plan's capacity, and the id every frame belongs to. One block per item; the
// cumulative sums live in LDS (T ≤ 4096), every frame finds its id by bisection. `rep` (host-mapped, may be null) receives
// [n] predicted frame counts BEFORE clamping, then the items' durations at rep + n + b·T.
__global__ __launch_bounds__(256) void dp_paths_kernel(const int32_t* __restrict__ dur, const int* __restrict__ lensT, int T, int F, int cap,
                                                        int32_t* __restrict__ frame2id, int* __restrict__ lensF, int32_t* __restrict__ rep, int n) {
  __shared__ int cum[4096];
  __shared__ int part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Tb = min(lensT[b], T);
  const int per = (T + 255) / 256;  // ids per thread, consecutive
  const int t0 = tid * per;
  int sum = 0;
  for (int i = 0; i < per; i++) {
    const int t = t0 + i;
    const int d = t < Tb ? max(dur[(size_t)b * T + t], 0) : 0;
    if (rep && t < T) rep[n + (size_t)b * T + t] = d;
    sum += d;
    if (t < T) cum[t] = sum;  // within the thread's run
  }
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // inclusive scan of the per-thread sums
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const int base = tid ? part[tid - 1] : 0;
  for (int i = 0; i < per; i++)
    if (t0 + i < T) cum[t0 + i] += base;
  __syncthreads();
  const int total = part[255];
  const int want = max(total, 1);          // an all-zero prediction still yields one frame (of id 0)
  const int Fv = min(min(want, cap), F);
  if (tid == 0) {
    lensF[b] = Fv;
    if (rep) { rep[b] = want; if (total < 1) rep[n + (size_t)b * T] = 1; }
  }
  for (int f = tid; f < F; f += 256) {
    int id = 0;
    if (f < Fv && f < total) {  // smallest t with cum[t] > f
      int lo = 0, hi = Tb - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > f) hi = mid; else lo = mid + 1;
      }
      id = lo;
    }
    frame2id[(size_t)b * F + f] = id;
  }
}

}  // namespace

// Whole utterances with PREDICTED durations and no host round trip between the predictor and the rest: the caller states an upper bound on
// the frames per item; the plan is the bucket of that bound and every kernel masks by the frame count the device decides. One stream carries
// uploads → encoder + predictor → generate_path (dp_paths_kernel) → flow + generator; the host learns the lengths with the waveform (collect).
PH_EXPORT int piper_hip_voice_prepare_batch_bounded(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int slot, int max_frames) {
  if (!v || !utts) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (n < 1 || n > 256) PH_FAIL(PIPER_HIP_ERR_SHAPE, "batch size %d outside [1,256]", n);
  if (slot < 0 || slot >= kMaxSlots) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d out of range [0,%d)", slot, kMaxSlots);
  if (max_frames < 1 || (int64_t)max_frames * std::max(v->hop, 1) > 0x3fffffff / 64) PH_FAIL(PIPER_HIP_ERR_SHAPE, "prepare_batch_bounded: max_frames %d out of range", max_frames);
  if (!v->cfg.dp_present) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "voice has no duration predictor (dp_present = 0): supply durations");
  int Tmax = 0;
  for (int b = 0; b < n; b++) {
    const piper_hip_utterance& u = utts[b];
    if (!u.phoneme_ids) PH_FAIL(PIPER_HIP_ERR_ARG, "utterance %d: null ids", b);
    if (u.t < 1 || u.t > 4096) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance %d: %d ids outside [1,4096]", b, u.t);
    if (u.durations || u.noise)
      PH_FAIL(PIPER_HIP_ERR_ARG, "utterance %d: prepare_batch_bounded predicts the durations and cannot take a host noise tensor (its shape depends "
                                 "on them): pass durations = noise = NULL (noise_mode DEVICE draws it on the device)", b);
    if (u.noise_mode != PIPER_HIP_NOISE_INJECTED && u.noise_mode != PIPER_HIP_NOISE_DEVICE)
      PH_FAIL(PIPER_HIP_ERR_ARG, "utterance %d: unknown noise_mode %d", b, u.noise_mode);
    Tmax = std::max(Tmax, u.t);
  }
  const int T = bucket_t(Tmax), F = bucket_f(max_frames), I = v->cfg.inter;
  if ((int64_t)F * v->hop * n > 0x3fffffff) PH_FAIL(PIPER_HIP_ERR_SHAPE, "batch too large");
  PH_HIP(hipSetDevice(v->ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  int rc;
  Slot* cur = v->attached[slot];
  const bool same = cur && cur->built && cur->kind == 3 && cur->T == T && cur->F == F && cur->NB == n && cur->prec == v->precision;
  if (cur && !same) detach(v, slot);
  if (!same) {
    bool built = false;
    if ((rc = acquire_plan(v, 3, T, F, n, &cur, &built))) return rc;
    cur->in_use = true;
    v->attached[slot] = cur;
  }
  Slot& s = *cur;
  s.last_use = ++v->use_clock;
  PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);  // the previous request of this slot id: its staging and its predictor plan are idle now
  Slot* dp = v->attached_dp[slot];
  if (dp && !(dp->built && dp->T == T && dp->NB == n && dp->prec == v->precision)) { dp->in_use = false; dp = v->attached_dp[slot] = nullptr; }
  if (!dp) {
    bool built = false;
    if ((rc = acquire_plan(v, 2, T, 16, n, &dp, &built))) return rc;
    dp->in_use = true;
    v->attached_dp[slot] = dp;
  }
  dp->last_use = s.last_use;
  evict_idle_plans(v);
  auto& sg = v->staging[slot];
  const size_t misc_bytes = (size_t)n * (sizeof(int) + 2 * sizeof(unsigned) + sizeof(float)) + dp_scalars_bytes(n);
  if ((rc = grow_pinned(sg.h_ids, sg.cap_t, (size_t)T * n)) || (rc = grow_pinned(sg.h_misc, sg.cap_misc, misc_bytes)) ||
      (rc = grow_pinned(sg.h_dpn, sg.cap_dpn, (size_t)n * 2 * T)) || (rc = grow_pinned(sg.h_res, sg.cap_res, (size_t)n + (size_t)n * T)))
    return rc;
  int* h_lens = (int*)sg.h_misc;
  unsigned* h_rng = (unsigned*)(h_lens + n);
  float* h_ns = (float*)(h_rng + 2 * n);
  char* h_sc = (char*)(h_ns + n);
  bool any_noise = false;
  for (int b = 0; b < n; b++) {
    const piper_hip_utterance& u = utts[b];
    memcpy(sg.h_ids + (size_t)b * T, u.phoneme_ids, (size_t)u.t * sizeof(int64_t));
    for (int t = u.t; t < T; t++) sg.h_ids[(size_t)b * T + t] = 0;
    h_lens[b] = u.t;
    const bool dev = u.noise_mode == PIPER_HIP_NOISE_DEVICE;
    h_rng[2 * b] = dev ? 1u : 0u;
    h_rng[2 * b + 1] = u.seed;
    h_ns[b] = u.noise_scale;
    for (int r = 0; r < 2; r++) {
      float* row = sg.h_dpn + ((size_t)b * 2 + r) * T;
      if (u.dp_noise) { memcpy(row, u.dp_noise + (size_t)r * u.t, (size_t)u.t * sizeof(float)); any_noise = true; }
      for (int t = u.dp_noise ? u.t : 0; t < T; t++) row[t] = 0.0f;
    }
    dp_scalars_fill(h_sc, b, u.noise_w, u.length_scale == 0.0f ? 1.0f : u.length_scale, (!u.dp_noise && dev) ? 1u : 0u, u.seed);
    sg.h_res[b] = -1;
  }
  (void)any_noise;
  const hipStream_t q = s.stream;
  PH_HIP(hipMemcpyAsync(dp->ids, sg.h_ids, (size_t)n * T * sizeof(int64_t), hipMemcpyHostToDevice, q), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(dp->lensT, h_lens, (size_t)n * sizeof(int), hipMemcpyHostToDevice, q), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(dp->dp_noise, sg.h_dpn, (size_t)n * 2 * T * sizeof(float), hipMemcpyHostToDevice, q), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(dp->dp_scalars, h_sc, dp_scalars_bytes(n), hipMemcpyHostToDevice, q), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.lensT, h_lens, (size_t)n * sizeof(int), hipMemcpyHostToDevice, q), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.rng, h_rng, 2 * (size_t)n * sizeof(unsigned), hipMemcpyHostToDevice, q), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipMemcpyAsync(s.noise_scale, h_ns, (size_t)n * sizeof(float), hipMemcpyHostToDevice, q), PIPER_HIP_ERR_LAUNCH);
  for (int b = 0; b < n; b++)  // INJECTED without a tensor = zero noise, as in prepare_batch
    if (!h_rng[2 * b]) PH_HIP(hipMemsetAsync(s.noise + (size_t)b * I * F, 0, (size_t)I * F * sizeof(float), q), PIPER_HIP_ERR_LAUNCH);
  if ((rc = launch_plan(v, *dp, q))) return rc;
  PH_HIP(hipMemcpyAsync(s.stats, dp->stats, (size_t)n * 2 * I * T * sizeof(float), hipMemcpyDeviceToDevice, q), PIPER_HIP_ERR_LAUNCH);
  int32_t* rep = nullptr;
  if (hipHostGetDevicePointer((void**)&rep, sg.h_res, 0) != hipSuccess) { rep = nullptr; (void)hipGetLastError(); }
  if (!rep) PH_FAIL(PIPER_HIP_ERR_UNAVAILABLE, "prepare_batch_bounded: page-locked memory has no device mapping on this system");
  hipLaunchKernelGGL(dp_paths_kernel, dim3(n), dim3(256), 0, q, dp->dp_dur, dp->lensT, T, F, std::min(max_frames, F), s.frame2id, s.lensF, rep, n);
  PH_HIP(hipGetLastError(), PIPER_HIP_ERR_LAUNCH);
  s.st_next = -1;
  s.h_T.assign(n, 0);
  for (int b = 0; b < n; b++) s.h_T[b] = utts[b].t;
  s.h_F.assign(n, F);  // capacity until collect has the device's answer
  s.h_dur.clear();
  s.bounded_pending = true;
  s.bounded_cap = std::min(max_frames, F);
  s.timed = false;
  return slot;
}

namespace {
// after the slot's stream has been synchronised: the device's frame counts and durations → h_F / h_dur; an item over the bound is an error
int bounded_finish(piper_hip_voice* v, int slot, Slot& s) {
  const auto& sg = v->staging[slot];
  s.bounded_pending = false;
  s.h_dur.clear();
  int over = -1;
  for (int b = 0; b < s.NB; b++) {
    const int want = sg.h_res[b];
    if (want < 1) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "bounded utterance %d: the device reported no frame count (was the slot launched?)", b);
    if (want > s.bounded_cap && over < 0) over = b;
    s.h_F[b] = std::min(want, s.bounded_cap);
    const int32_t* d = sg.h_res + s.NB + (size_t)b * s.T;
    s.h_dur.insert(s.h_dur.end(), d, d + s.h_T[b]);
  }
  if (over >= 0)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "bounded utterance %d: the predictor wants %d frames, the caller allowed %d — prepare it again with a larger bound "
                                 "(or without one)", over, sg.h_res[over], s.bounded_cap);
  return PIPER_HIP_OK;
}

// item b of a bounded slot → page-locked host memory at its bucket offset; the length comes from device memory
__global__ __launch_bounds__(256) void copy_out_len_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ lensF, int b, int hop) {
  const int64_t n = (int64_t)lensF[b] * hop;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  const int64_t n4 = n >> 2;  // hop·F·4 bytes: rows are 16-byte aligned when hop % 4 == 0 (checked by the caller)
  for (int64_t i = tid; i < n4; i += nth) ((float4*)dst)[i] = ((const float4*)src)[i];
  for (int64_t i = (n4 << 2) + tid; i < n; i += nth) dst[i] = src[i];
}
}  // namespace

namespace {
int predict_impl(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int32_t* durations_out, float* logw_out, int max_entries, Slot** plan_out);
}
PH_EXPORT int piper_hip_voice_predict_durations(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int32_t* durations_out, float* logw_out,
                                                int max_entries) {
  return predict_impl(v, utts, n, durations_out, logw_out, max_entries, nullptr);
}
namespace {
// plan_out (optional): the encoder + predictor plan that ran, left marked in_use so that its m_p / logs_p stay put until the caller
// has enqueued the copy into the plan that continues from them (the caller clears in_use)
int predict_impl(piper_hip_voice* v, const piper_hip_utterance* utts, int n, int32_t* durations_out, float* logw_out, int max_entries, Slot** plan_out) {
  if (!v || !utts || !durations_out) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (n < 1 || n > 256) PH_FAIL(PIPER_HIP_ERR_SHAPE, "batch size %d outside [1,256]", n);
  if (!v->cfg.dp_present) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "voice has no duration predictor (dp_present = 0)");
  int Tmax = 0;
  int64_t total = 0;
  for (int b = 0; b < n; b++) {
    if (!utts[b].phoneme_ids) PH_FAIL(PIPER_HIP_ERR_ARG, "utterance %d: null ids", b);
    if (utts[b].t < 1 || utts[b].t > 4096) PH_FAIL(PIPER_HIP_ERR_SHAPE, "utterance %d: %d ids outside [1,4096]", b, utts[b].t);
    if (utts[b].noise_mode != PIPER_HIP_NOISE_INJECTED && utts[b].noise_mode != PIPER_HIP_NOISE_DEVICE)
      PH_FAIL(PIPER_HIP_ERR_ARG, "utterance %d: unknown noise_mode %d", b, utts[b].noise_mode);
    Tmax = std::max(Tmax, utts[b].t);
    total += utts[b].t;
  }
  if (max_entries < total) PH_FAIL(PIPER_HIP_ERR_SHAPE, "predict_durations: output holds %d < %lld entries", max_entries, (long long)total);
  PH_HIP(hipSetDevice(v->ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  const int T = bucket_t(Tmax);
  Slot* pl = nullptr;
  bool built = false;
  int rc = acquire_plan(v, 2, T, 16, n, &pl, &built);
  if (rc) return rc;
  Slot& s = *pl;
  s.in_use = true;
  s.last_use = ++v->use_clock;
  std::vector<int64_t> ids((size_t)n * T, 0);
  std::vector<int> lens(n);
  std::vector<float> nz((size_t)n * 2 * T, 0.0f);
  std::vector<char> sc(dp_scalars_bytes(n));
  for (int b = 0; b < n; b++) {
    const piper_hip_utterance& u = utts[b];
    memcpy(ids.data() + (size_t)b * T, u.phoneme_ids, (size_t)u.t * sizeof(int64_t));
    lens[b] = u.t;
    if (u.dp_noise)
      for (int r = 0; r < 2; r++) memcpy(nz.data() + ((size_t)b * 2 + r) * T, u.dp_noise + (size_t)r * u.t, (size_t)u.t * sizeof(float));
    // device mode draws element (row, t) of the item's OWN [1, 2, T_b] tensor: the kernel indexes the bucket row, so the draw
    // index must be remapped when T_b < T — done by generating on the host side of the index: see dp_init_kernel (uses T).
    dp_scalars_fill(sc.data(), b, u.noise_w, u.length_scale == 0.0f ? 1.0f : u.length_scale,
                    (!u.dp_noise && u.noise_mode == PIPER_HIP_NOISE_DEVICE) ? 1u : 0u, u.seed);
  }
  hipError_t e = hipMemcpyAsync(s.ids, ids.data(), ids.size() * sizeof(int64_t), hipMemcpyHostToDevice, s.stream);
  if (e == hipSuccess) e = hipMemcpyAsync(s.lensT, lens.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s.stream);
  if (e == hipSuccess) e = hipMemcpyAsync(s.dp_noise, nz.data(), nz.size() * sizeof(float), hipMemcpyHostToDevice, s.stream);
  if (e == hipSuccess) e = hipMemcpyAsync(s.dp_scalars, sc.data(), sc.size(), hipMemcpyHostToDevice, s.stream);
  if (e == hipSuccess && launch_plan(v, s)) e = hipErrorUnknown;
  std::vector<int32_t> dur((size_t)n * T);
  std::vector<float> lw(logw_out ? (size_t)n * T : 0);
  if (e == hipSuccess) e = hipMemcpyAsync(dur.data(), s.dp_dur, dur.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream);
  if (e == hipSuccess && logw_out) e = hipMemcpyAsync(lw.data(), s.taps["logw"].p, lw.size() * sizeof(float), hipMemcpyDeviceToHost, s.stream);
  if (e == hipSuccess) e = stream_wait(s.stream);
  if (plan_out && e == hipSuccess) *plan_out = pl;  // stays in_use: see above
  else s.in_use = false;
  evict_idle_plans(v);
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "predict_durations: %s", hipGetErrorString(e));
  int64_t off = 0;
  for (int b = 0; b < n; b++) {
    memcpy(durations_out + off, dur.data() + (size_t)b * T, (size_t)utts[b].t * sizeof(int32_t));
    if (logw_out) memcpy(logw_out + off, lw.data() + (size_t)b * T, (size_t)utts[b].t * sizeof(float));
    off += utts[b].t;
  }
  return PIPER_HIP_OK;
}
}  // namespace

PH_EXPORT int piper_hip_voice_prepared_samples(const piper_hip_voice* v, int slot, int64_t* per_item, int max_items, int64_t* total) {
  const Slot* p = slot_plan(v, slot);
  if (!p) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d is not prepared", slot);
  int64_t sum = 0;
  for (int b = 0; b < p->NB; b++) {
    const int64_t nb = (int64_t)p->h_F[b] * v->hop;
    if (per_item && b < max_items) per_item[b] = nb;
    sum += nb;
  }
  if (total) *total = sum;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_durations(const piper_hip_voice* v, int slot, int32_t* out, int max_entries, int* n_entries) {
  const Slot* p = slot_plan(v, slot);
  if (!p) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d is not prepared", slot);
  if (n_entries) *n_entries = (int)p->h_dur.size();
  if (out) {
    if (max_entries < (int)p->h_dur.size()) PH_FAIL(PIPER_HIP_ERR_SHAPE, "durations: buffer too small");
    memcpy(out, p->h_dur.data(), p->h_dur.size() * sizeof(int32_t));
  }
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_prepare(piper_hip_voice* v, const piper_hip_utterance* u, int slot) {
  return piper_hip_voice_prepare_batch(v, u, 1, slot);
}

PH_EXPORT int piper_hip_voice_batch_size(const piper_hip_voice* v, int slot) {
  const Slot* p = slot_plan(v, slot);
  return p ? p->NB : 0;
}

PH_EXPORT int piper_hip_voice_set_plan_cache(piper_hip_voice* v, int max_plans, size_t max_bytes) {
  if (!v || max_plans < 1) PH_FAIL(PIPER_HIP_ERR_ARG, "set_plan_cache: need a voice and max_plans >= 1");
  v->plan_cache_max = (size_t)max_plans;
  v->plan_cache_bytes = max_bytes ? max_bytes : kPlanCacheBytes;
  evict_idle_plans(v);
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_last_build_breakdown(const piper_hip_voice* v, double out_ms[6]) {
  if (!v || !out_ms) PH_FAIL(PIPER_HIP_ERR_ARG, "last_build_breakdown: null argument");
  for (int i = 0; i < 6; i++) out_ms[i] = v->last_build_ms[i];
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_plan_info(const piper_hip_voice* v, int slot, int32_t* bucket_t_out, int32_t* bucket_f_out, int32_t* cached_plans,
                                        size_t* cached_bytes) {
  if (!v) PH_FAIL(PIPER_HIP_ERR_ARG, "null voice");
  const Slot* p = slot_plan(v, slot);
  if (bucket_t_out) *bucket_t_out = p ? p->T : 0;
  if (bucket_f_out) *bucket_f_out = p ? p->F : 0;
  if (cached_plans) *cached_plans = (int32_t)v->plans.size();
  if (cached_bytes) {
    size_t b = 0;
    for (auto& pl : v->plans) b += pl->arena_bytes;
    *cached_bytes = b;
  }
  return PIPER_HIP_OK;
}

namespace {
// waveform → page-locked host memory through its device mapping (collect): 16-byte stores where both sides allow, grid-stride
__global__ __launch_bounds__(256) void copy_out_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nth) ((float4*)dst)[i] = ((const float4*)src)[i];
    for (int64_t i = (n4 << 2) + tid; i < n; i += nth) dst[i] = src[i];
  } else {
    for (int64_t i = tid; i < n; i += nth) dst[i] = src[i];
  }
}
}  // namespace

PH_EXPORT int piper_hip_voice_launch(piper_hip_voice* v, int slot) {
  if (!v) PH_FAIL(PIPER_HIP_ERR_ARG, "null voice");
  Slot* p = slot_plan(v, slot);
  if (!p) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d is not prepared", slot);
  PH_HIP(hipSetDevice(v->ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  Slot& s = *p;
  PH_HIP(hipEventRecord(s.ev0, s.stream), PIPER_HIP_ERR_LAUNCH);
  {
    const int lrc = launch_plan(v, s);
    if (lrc) return lrc;
  }
  PH_HIP(hipEventRecord(s.ev1, s.stream), PIPER_HIP_ERR_LAUNCH);
  s.timed = true;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_collect(piper_hip_voice* v, int slot, float* host_audio, int64_t max_samples) {
  if (!v) PH_FAIL(PIPER_HIP_ERR_ARG, "null voice");
  Slot* p = slot_plan(v, slot);
  if (!p) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d is not prepared", slot);
  PH_HIP(hipSetDevice(v->ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  Slot& s = *p;
  if (s.bounded_pending) {
    // lengths still on the device. Short buckets: the waveform rows go to the slot id's page-locked buffer at bucket stride by a kernel that
    // reads each length from device memory — ONE synchronisation for lengths and samples; long ones: synchronise, then the usual copy.
    const int64_t row = (int64_t)s.F * v->hop;
    const size_t ub = (size_t)row * s.NB * sizeof(float);
    if (host_audio && max_samples < row * s.NB)
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "collect: a bounded slot needs room for its capacity (%lld samples: piper_hip_voice_prepared_samples before collect), got %lld",
              (long long)(row * s.NB), (long long)max_samples);
    auto& sg = v->staging[slot];
    float* dst_dev = nullptr;
    if (host_audio && ub <= ((size_t)1 << 20) && (v->hop & 3) == 0) {
      if (sg.audio_cap < ub) {
        if (sg.h_audio) (void)hipHostFree(sg.h_audio);
        sg.h_audio = nullptr; sg.audio_cap = 0;
        size_t c = 65536;
        while (c < ub) c <<= 1;
        if (hipHostMalloc((void**)&sg.h_audio, c) == hipSuccess) sg.audio_cap = c;
        else { sg.h_audio = nullptr; (void)hipGetLastError(); }
      }
      if (sg.h_audio && hipHostGetDevicePointer((void**)&dst_dev, sg.h_audio, 0) != hipSuccess) { dst_dev = nullptr; (void)hipGetLastError(); }
    }
    if (dst_dev) {
      for (int b = 0; b < s.NB; b++) {
        const int blocks = (int)std::min<int64_t>((row + 4 * 256 - 1) / (4 * 256), 1024);
        hipLaunchKernelGGL(copy_out_len_kernel, dim3(blocks), dim3(256), 0, s.stream, s.audio + (int64_t)b * s.n_samples, dst_dev + (int64_t)b * row, s.lensF, b, v->hop);
        PH_HIP(hipGetLastError(), PIPER_HIP_ERR_LAUNCH);
      }
    }
    PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
    const int frc = bounded_finish(v, slot, s);
    if (frc) return frc;
    if (dst_dev) {
      int64_t off = 0;
      for (int b = 0; b < s.NB; b++) {
        const int64_t nb = (int64_t)s.h_F[b] * v->hop;
        memcpy(host_audio + off, sg.h_audio + (int64_t)b * row, (size_t)nb * sizeof(float));
        off += nb;
      }
      return PIPER_HIP_OK;
    }
    // fall through: the lengths are known now
  }
  if (host_audio) {
    int64_t total = 0;  // batch items back to back, each at its own true length
    for (int b = 0; b < s.NB; b++) total += (int64_t)s.h_F[b] * v->hop;
    if (max_samples < total) PH_FAIL(PIPER_HIP_ERR_SHAPE, "collect: buffer holds %lld < %lld samples", (long long)max_samples, (long long)total);
    // A copy into the caller's (pageable) buffer goes through the runtime's own staging in chunks and its time varies from
    // box to box (r2z: 45 … 130 µs for 344 KB). Up to 1 MB (a factor-8 … 16 utterance) the waveform lands in a pinned buffer of
    // the plan by one DMA and is copied out by the host; beyond that the runtime's pipelined chunks beat DMA + memcpy
    // (factor 64, 2.75 MB: +0.12 ms with the pinned hop).
    constexpr size_t kPinnedMax = (size_t)1 << 20;
    // 1 … 16 MB into PAGEABLE memory: handing the caller's buffer to the runtime makes it page-lock that buffer on the spot, and for a buffer
    // it has not seen before that took 7 ms (r3, tools/probe/first_run.py: the first 1.2 MB waveform of a process, 9.4 ms in collect for
    // 1.9 ms of GPU work). Such waveforms land in the slot id's page-locked buffer in 1 MB chunks, each followed by an event, and the host
    // copies chunk k out while chunk k + 1 is on the wire. Larger ones still go to the runtime (its pipelined staging wins there).
    constexpr size_t kChunkedMax = (size_t)16 << 20, kChunk = (size_t)1 << 20;
    const size_t bytes = (size_t)total * sizeof(float);
    // a destination the caller page-locked itself (piper_hip_host_alloc) takes the DMA directly
    bool caller_pinned = false;
    {
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, host_audio) == hipSuccess) caller_pinned = at.type == hipMemoryTypeHost;
      else (void)hipGetLastError();
    }
    {
      auto& sg = v->staging[slot];
      if (!caller_pinned && bytes <= kChunkedMax && sg.audio_cap < bytes) {
        if (sg.h_audio) (void)hipHostFree(sg.h_audio);
        sg.h_audio = nullptr; sg.audio_cap = 0;
        size_t c = 65536;
        while (c < bytes) c <<= 1;
        if (hipHostMalloc((void**)&sg.h_audio, c) == hipSuccess) sg.audio_cap = c;
        else { sg.h_audio = nullptr; (void)hipGetLastError(); }
      }
      s.h_audio = sg.audio_cap >= bytes ? sg.h_audio : nullptr;
    }
    if (!caller_pinned && bytes > kPinnedMax && bytes <= kChunkedMax && s.h_audio) {
      auto& sg = v->staging[slot];
      // the items back to back in the staging buffer, cut into chunks; `cuts` = end offset (floats) of each chunk
      std::vector<int64_t> cuts;
      int64_t off = 0;
      for (int b = 0; b < s.NB; b++) {
        const int64_t nb = (int64_t)s.h_F[b] * v->hop;
        const float* src = s.audio + (int64_t)b * s.n_samples;
        for (int64_t c0 = 0; c0 < nb; c0 += (int64_t)(kChunk / sizeof(float))) {
          const int64_t cn = std::min<int64_t>((int64_t)(kChunk / sizeof(float)), nb - c0);
          PH_HIP(hipMemcpyAsync(s.h_audio + off + c0, src + c0, (size_t)cn * sizeof(float), hipMemcpyDeviceToHost, s.stream), PIPER_HIP_ERR_LAUNCH);
          const size_t k = cuts.size();
          if (sg.chunk_ev.size() <= k) {
            hipEvent_t e = nullptr;
            PH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), PIPER_HIP_ERR_LAUNCH);
            sg.chunk_ev.push_back(e);
          }
          PH_HIP(hipEventRecord(sg.chunk_ev[k], s.stream), PIPER_HIP_ERR_LAUNCH);
          cuts.push_back(off + c0 + cn);
        }
        off += nb;
      }
      int64_t done = 0;
      for (size_t k = 0; k < cuts.size(); k++) {
        for (;;) {  // poll: the chunks arrive every ≈ 20 µs
          const hipError_t e = hipEventQuery(sg.chunk_ev[k]);
          if (e == hipSuccess) break;
          if (e != hipErrorNotReady) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "collect: %s", hipGetErrorString(e));
          (void)hipGetLastError();
          for (int i = 0; i < 16; i++) __builtin_ia32_pause();
        }
        memcpy(host_audio + done, s.h_audio + done, (size_t)(cuts[k] - done) * sizeof(float));
        done = cuts[k];
      }
      return PIPER_HIP_OK;
    }
    float* dst = (!caller_pinned && bytes <= kPinnedMax && s.h_audio) ? s.h_audio : host_audio;
    // Short waveforms into page-locked memory are written by a KERNEL through the host mapping instead of the copy engine: the
    // hand-over from the last kernel of the graph to another kernel costs ≈ 2 µs, to the DMA engine ≈ 10 µs (PIPER_HIP_COLLECT_DMA=1:
    // always the copy engine).
    static const bool dma_only = getenv("PIPER_HIP_COLLECT_DMA") != nullptr;
    float* dst_dev = nullptr;
    // (a destination the caller page-locked takes the copy kernel up to 16 MB: the engine's hand-over and completion varied 0.07 … 0.3 ms
    // from process to process on a 2.75 MB waveform, r3)
    if (!dma_only && (caller_pinned ? bytes <= kChunkedMax : bytes <= kPinnedMax) && (caller_pinned || dst == s.h_audio)) {
      if (hipHostGetDevicePointer((void**)&dst_dev, dst, 0) != hipSuccess) { dst_dev = nullptr; (void)hipGetLastError(); }
    }
    int64_t off = 0;
    for (int b = 0; b < s.NB; b++) {
      const int64_t nb = (int64_t)s.h_F[b] * v->hop;
      const float* src = s.audio + (int64_t)b * s.n_samples;
      if (dst_dev && nb > 0) {
        const int blocks = (int)std::min<int64_t>((nb + 4 * 256 - 1) / (4 * 256), 1024);
        hipLaunchKernelGGL(copy_out_kernel, dim3(blocks), dim3(256), 0, s.stream, src, dst_dev + off, nb);
        PH_HIP(hipGetLastError(), PIPER_HIP_ERR_LAUNCH);
      } else {
        PH_HIP(hipMemcpyAsync(dst + off, src, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost, s.stream), PIPER_HIP_ERR_LAUNCH);
      }
      off += nb;
    }
    // a copy-ENGINE transfer completes through the runtime's own signal handling: polling the stream next to it delayed the completion of a
    // 2.75 MB waveform by ≈ 0.2 ms (factor 64, r3) — park for those, poll only behind the copy kernel
    if (dst_dev) PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
    else PH_HIP(hipStreamSynchronize(s.stream), PIPER_HIP_ERR_LAUNCH);
    if (dst != host_audio) memcpy(host_audio, dst, bytes);
    return PIPER_HIP_OK;
  }
  PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
  return PIPER_HIP_OK;
}

// ---- streaming: encoder + flow once, then the generator window by window --------------------------------------------
namespace {

// Frames of latent the generator needs on each side of an output frame (its receptive field, walked from the waveform back
// to z): conv_post ±3 samples; per stage the widest ResBlock reach, then the ConvTranspose; conv_pre ±3 frames.
int generator_halo_frames(const piper_hip_voice_config& c) {
  int64_t r = 3;
  for (int u = c.n_ups - 1; u >= 0; u--) {
    int64_t rb = 0;
    for (int j = 0; j < c.n_rb; j++) {
      int64_t reach = 0;
      for (int di = 0; di < c.rb_n_dil; di++) {
        const int64_t k = c.rb_kernels[j], d = c.rb_dilations[j][di];
        reach += (k * d - d) / 2 + (c.resblock_type == 1 ? (k - 1) / 2 : 0);
      }
      rb = std::max(rb, reach);
    }
    r += rb;
    r = (r + c.up_kernels[u]) / c.up_rates[u] + 1;
  }
  return (int)(r + 3);
}

// graph of the steps selected by `pick` (eager pass first: validates launches and sets kernel attributes)
int capture_steps(Slot& s, const std::vector<int>& pick, hipGraph_t* g, hipGraphExec_t* ge) {
  for (int i : pick) {
    int rc = s.steps[i].run(s.stream);
    if (rc) return rc;
  }
  PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal), PIPER_HIP_ERR_LAUNCH);
  int rc = PIPER_HIP_OK;
  for (int i : pick)
    if ((rc = s.steps[i].run(s.stream))) break;
  hipError_t ce = hipStreamEndCapture(s.stream, g);
  if (rc || ce != hipSuccess) {
    if (*g) { (void)hipGraphDestroy(*g); *g = nullptr; }
    if (rc) return rc;
    PH_FAIL(PIPER_HIP_ERR_LAUNCH, "stream: graph capture failed: %s", hipGetErrorString(ce));
  }
  ce = hipGraphInstantiate(ge, *g, nullptr, nullptr, 0);
  if (ce != hipSuccess) {
    (void)hipGraphDestroy(*g); *g = nullptr;
    PH_FAIL(PIPER_HIP_ERR_LAUNCH, "stream: graph instantiate failed: %s", hipGetErrorString(ce));
  }
  return PIPER_HIP_OK;
}

}  // namespace

PH_EXPORT int piper_hip_voice_receptive_field(const piper_hip_voice* v) { return v ? generator_halo_frames(v->cfg) : -1; }

PH_EXPORT int piper_hip_voice_stream_begin(piper_hip_voice* v, const piper_hip_utterance* u, int slot, int chunk_frames) {
  if (chunk_frames < 1) PH_FAIL(PIPER_HIP_ERR_ARG, "stream_begin: chunk_frames must be >= 1");
  int rc = piper_hip_voice_prepare(v, u, slot);
  if (rc < 0) return rc;
  Slot& s = *v->attached[slot];
  if (!s.front_exec) {  // encoder + flow as their own graph (everything before the generator's first launch)
    std::vector<int> front;
    for (int i = 0; i < (int)s.steps.size(); i++) {
      if (s.steps[i].name.compare(0, 4, "dec.") == 0) break;
      if (s.steps[i].kind == Step::LAUNCH) front.push_back(i);
    }
    if ((rc = capture_steps(s, front, &s.front_graph, &s.front_exec))) return rc;
  }
  PH_HIP(hipGraphLaunch(s.front_exec, s.stream), PIPER_HIP_ERR_LAUNCH);
  PH_HIP(hipEventRecord(s.ev1, s.stream), PIPER_HIP_ERR_LAUNCH);  // z is ready when ev1 fires
  s.st_chunk = chunk_frames;
  s.st_halo = generator_halo_frames(v->cfg);
  s.st_next = 0;
  return (int)ceil_div(s.h_F[0], chunk_frames);
}

PH_EXPORT int piper_hip_voice_stream_next(piper_hip_voice* v, int slot, float* host_audio, int64_t max_samples, int64_t* n_samples) {
  if (!v || !n_samples) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  Slot* sp = slot_plan(v, slot);
  if (!sp || sp->st_next < 0) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d has no stream in progress", slot);
  PH_HIP(hipSetDevice(v->ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
  Slot& s = *sp;
  const int Ftrue = s.h_F[0];
  *n_samples = 0;
  if (s.st_next >= Ftrue) return PIPER_HIP_OK;  // end of stream
  const int f0 = s.st_next, f1 = std::min(Ftrue, f0 + s.st_chunk);
  // window = chunk + receptive field, clamped to the utterance: at the utterance's own ends the convs' zero padding is
  // then the same zero padding the whole-utterance run sees, inside it the halo frames are recomputed and dropped
  const int a = std::max(0, f0 - s.st_halo), b = std::min(Ftrue, f1 + s.st_halo);
  const int Fc = b - a;
  const int64_t want = (int64_t)(f1 - f0) * v->hop;
  if (host_audio && max_samples < want) PH_FAIL(PIPER_HIP_ERR_SHAPE, "stream_next: buffer holds %lld < %lld samples", (long long)max_samples, (long long)want);
  // generator-only plan of the window's bucket (first / interior / last windows of a stream usually share one)
  Slot* gs = nullptr;
  bool built = false;
  int rc = acquire_plan(v, 1, 0, bucket_f(Fc), 1, &gs, &built);
  if (rc) return rc;
  gs->in_use = true;
  gs->last_use = ++v->use_clock;
  const int I = v->cfg.inter;
  hipError_t e = hipStreamWaitEvent(gs->stream, s.ev1, 0);
  int lens[2] = {0, Fc};
  if (e == hipSuccess) e = hipMemcpyAsync(gs->lensF, &lens[1], sizeof(int), hipMemcpyHostToDevice, gs->stream);
  if (e == hipSuccess)
    e = hipMemcpy2DAsync(gs->zin, (size_t)gs->F * sizeof(float), s.z_out + a, (size_t)s.F * sizeof(float), (size_t)Fc * sizeof(float), (size_t)I,
                         hipMemcpyDeviceToDevice, gs->stream);
  if (e == hipSuccess && launch_plan(v, *gs)) e = hipErrorUnknown;
  if (e == hipSuccess && host_audio)
    e = hipMemcpyAsync(host_audio, gs->audio + (int64_t)(f0 - a) * v->hop, (size_t)want * sizeof(float), hipMemcpyDeviceToHost, gs->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(gs->stream);
  gs->in_use = false;
  evict_idle_plans(v);
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "stream_next: %s", hipGetErrorString(e));
  *n_samples = want;
  s.st_next = f1;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_synthesize(piper_hip_voice* v, const piper_hip_utterance* u, float* host_audio,
                                         int64_t max_samples, int64_t* n_samples) {
  int rc = piper_hip_voice_prepare(v, u, 0);
  if (rc < 0) return rc;
  if ((rc = piper_hip_voice_launch(v, 0))) return rc;
  if ((rc = piper_hip_voice_collect(v, 0, host_audio, max_samples))) return rc;
  if (n_samples) *n_samples = (int64_t)v->attached[0]->h_F[0] * v->hop;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_tap(piper_hip_voice* v, int slot, const char* name, float* host, size_t max_floats,
                                  size_t* n_floats) {
  if (!v || !name) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  Slot* sp = slot_plan(v, slot);
  if (!sp) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d is not prepared", slot);
  Slot& s = *sp;
  auto it = s.taps.find(name);
  if (it == s.taps.end()) PH_FAIL(PIPER_HIP_ERR_ARG, "unknown tap '%s'", name);
  const Slot::Tap& t = it->second;
  // items back to back, each compacted to its true length: [C][len_b]
  size_t total = 0;
  for (int b = 0; b < s.NB; b++) total += (size_t)t.C * (size_t)(t.unit == 0 ? s.h_T[b] : s.h_F[b]);
  if (n_floats) *n_floats = total;
  if (host) {
    if (max_floats < total) PH_FAIL(PIPER_HIP_ERR_SHAPE, "tap buffer too small");
    PH_HIP(hipSetDevice(v->ctx->device), PIPER_HIP_ERR_UNAVAILABLE);
    PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
    size_t off = 0;
    for (int b = 0; b < s.NB; b++) {
      const size_t len = (size_t)(t.unit == 0 ? s.h_T[b] : s.h_F[b]);
      PH_HIP(hipMemcpy2D(host + off, len * sizeof(float), t.p + (size_t)b * t.batch_stride, (size_t)t.row * sizeof(float), len * sizeof(float),
                         (size_t)t.C, hipMemcpyDeviceToHost), PIPER_HIP_ERR_LAUNCH);
      off += (size_t)t.C * len;
    }
  }
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_last_gpu_ms(piper_hip_voice* v, int slot, double* ms) {
  if (!v || !ms) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  Slot* sp = slot_plan(v, slot);
  if (!sp || !sp->timed) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d has no timed launch", slot);
  Slot& s = *sp;
  PH_HIP(hipEventSynchronize(s.ev1), PIPER_HIP_ERR_LAUNCH);
  float f = 0;
  PH_HIP(hipEventElapsedTime(&f, s.ev0, s.ev1), PIPER_HIP_ERR_LAUNCH);
  *ms = f;
  return PIPER_HIP_OK;
}

PH_EXPORT piper_hip_stream piper_hip_voice_slot_stream(piper_hip_voice* v, int slot) {
  Slot* sp = slot_plan(v, slot);
  return sp ? (piper_hip_stream)sp->stream : nullptr;
}

PH_EXPORT int piper_hip_voice_profile(piper_hip_voice* v, int slot, int iters, piper_hip_kernel_stat* out, int max_entries,
                                      int* n_entries) {
  if (!v) PH_FAIL(PIPER_HIP_ERR_ARG, "null voice");
  Slot* sp = slot_plan(v, slot);
  if (!sp) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d is not prepared", slot);
  if (iters < 1) iters = 1;
  Slot& s = *sp;
  const int n = (int)s.steps.size();
  if (n_entries) *n_entries = n;
  if (!out) return PIPER_HIP_OK;
  std::vector<hipEvent_t> ev((size_t)n + 1);
  for (auto& e : ev) PH_HIP(hipEventCreate(&e), PIPER_HIP_ERR_LAUNCH);
  std::vector<std::vector<float>> samples((size_t)n);  // per launch: one delta per pass — the MEDIAN is reported (a stall of one
                                                        // pass on a fresh box must not poison a row: BENCH_r01 showed a 324 µs enc2.qkv)
  int rc = PIPER_HIP_OK;
  for (int it = 0; it < iters + 1 && !rc; it++) {  // first pass is warm-up
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s.stream, (unsigned long long)(100 * (1500 + 12 * n)));  // µs → ticks
    PH_HIP(hipEventRecord(ev[0], s.stream), PIPER_HIP_ERR_LAUNCH);
    for (int i = 0; i < n && !rc; i++) {
      if (s.steps[i].kind == Step::LAUNCH) rc = s.steps[i].run(s.stream);
      if (!rc && hipEventRecord(ev[i + 1], s.stream) != hipSuccess) rc = PIPER_HIP_ERR_LAUNCH;
    }
    if (rc) break;
    PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
    if (it == 0) continue;
    for (int i = 0; i < n; i++) {
      float ms = 0;
      PH_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]), PIPER_HIP_ERR_LAUNCH);
      samples[i].push_back(ms * 1000.0f);
    }
  }
  // event floor: the same bracket around an empty kernel (dispatch boundary + event markers, no work) — what has to be
  // subtracted from a step's event delta to compare it with rocprofv3's begin→end kernel duration
  double floor_us = 0.0;
  if (!rc) {
    const int nf = std::min(32, n);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s.stream, (unsigned long long)(100 * 800));
    if (hipEventRecord(ev[0], s.stream) != hipSuccess) rc = PIPER_HIP_ERR_LAUNCH;
    for (int i = 1; i <= nf && !rc; i++) {
      hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s.stream);
      if (hipEventRecord(ev[i], s.stream) != hipSuccess) rc = PIPER_HIP_ERR_LAUNCH;
    }
    if (!rc && nf > 0 && stream_wait(s.stream) == hipSuccess) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev[0], ev[nf]) == hipSuccess) floor_us = ms * 1000.0 / nf;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc) return rc;
  for (int i = 0; i < n && i < max_entries; i++) {
    memset(&out[i], 0, sizeof out[i]);
    snprintf(out[i].name, sizeof out[i].name, "%s", s.steps[i].name.c_str());
    std::vector<float>& sm = samples[i];
    std::sort(sm.begin(), sm.end());
    out[i].avg_us = sm.empty() ? 0.0 : (sm.size() & 1 ? sm[sm.size() / 2] : 0.5 * (sm[sm.size() / 2 - 1] + sm[sm.size() / 2]));
    out[i].flops = s.steps[i].flops;
    out[i].bytes = s.steps[i].bytes;
  }
  if (n < max_entries) {  // extra pseudo-entry carrying the calibration
    memset(&out[n], 0, sizeof out[n]);
    snprintf(out[n].name, sizeof out[n].name, "(event floor: empty kernel)");
    out[n].avg_us = floor_us;
    if (n_entries) *n_entries = n + 1;
  }
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_time_subset(piper_hip_voice* v, int slot, const char* name_filter, int iters, double* avg_launch_us,
                                          int* n_launches, double* flops, double* bytes) {
  if (!v || !name_filter) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  Slot* sp = slot_plan(v, slot);
  if (!sp) PH_FAIL(PIPER_HIP_ERR_ARG, "slot %d is not prepared", slot);
  if (iters < 1) iters = 1;
  Slot& s = *sp;
  std::vector<int> pick;
  double fl = 0, by = 0;
  // "a|b|c": the launches with exactly these names (one replayed graph for a whole kernel family); otherwise substring / tag match
  std::vector<std::string> exact;
  if (strchr(name_filter, '|')) {
    std::string f(name_filter);
    size_t b = 0;
    while (b <= f.size()) {
      const size_t e = f.find('|', b);
      const std::string part = f.substr(b, e == std::string::npos ? std::string::npos : e - b);
      if (!part.empty()) exact.push_back(part);
      if (e == std::string::npos) break;
      b = e + 1;
    }
  }
  auto wanted = [&](const Step& st) {
    if (!exact.empty()) return std::find(exact.begin(), exact.end(), st.name) != exact.end();
    return st.name.find(name_filter) != std::string::npos || (!st.tag.empty() && st.tag == name_filter);
  };
  for (int i = 0; i < (int)s.steps.size(); i++)
    if (s.steps[i].kind == Step::LAUNCH && wanted(s.steps[i])) {
      pick.push_back(i);
      fl += s.steps[i].flops;
      by += s.steps[i].bytes;
    }
  if (n_launches) *n_launches = (int)pick.size();
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  if (pick.empty()) {
    if (avg_launch_us) *avg_launch_us = 0;
    return PIPER_HIP_OK;
  }
  PH_HIP(stream_wait(s.stream), PIPER_HIP_ERR_LAUNCH);
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  PH_HIP(hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal), PIPER_HIP_ERR_LAUNCH);
  int rc = PIPER_HIP_OK;
  for (int i : pick)
    if ((rc = s.steps[i].run(s.stream))) break;
  hipError_t ce = hipStreamEndCapture(s.stream, &g);
  if (rc || ce != hipSuccess) {
    if (g) (void)hipGraphDestroy(g);
    if (rc) return rc;
    PH_FAIL(PIPER_HIP_ERR_LAUNCH, "time_subset: capture failed: %s", hipGetErrorString(ce));
  }
  ce = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  if (ce != hipSuccess) {
    (void)hipGraphDestroy(g);
    PH_FAIL(PIPER_HIP_ERR_LAUNCH, "time_subset: instantiate failed: %s", hipGetErrorString(ce));
  }
  hipError_t e = hipGraphLaunch(ge, s.stream);  // warm-up
  if (e == hipSuccess) e = hipEventRecord(s.ev0, s.stream);
  for (int it = 0; it < iters && e == hipSuccess; it++) e = hipGraphLaunch(ge, s.stream);
  if (e == hipSuccess) e = hipEventRecord(s.ev1, s.stream);
  if (e == hipSuccess) e = hipEventSynchronize(s.ev1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, s.ev0, s.ev1);
  (void)hipGraphExecDestroy(ge);
  (void)hipGraphDestroy(g);
  s.timed = false;
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "time_subset: %s", hipGetErrorString(e));
  if (avg_launch_us) *avg_launch_us = (double)ms * 1000.0 / ((double)iters * (double)pick.size());
  return PIPER_HIP_OK;
}
