// voice_blob.cpp — voice geometry presets, packed-blob layout and the synthetic weight generator (host only).
// Geometry: SURVEY.md §8a († upstream Piper VITS settings; hints at GraphExecutor.swift:1872, ONNXParsingTests.swift:32).
#include <cmath>

#include "../../include/piper_hip_voice_layout.h"
#include "common.h"

PH_EXPORT int piper_hip_voice_config_preset(int quality, piper_hip_voice_config* c) {
  if (!c) PH_FAIL(PIPER_HIP_ERR_ARG, "null config");
  memset(c, 0, sizeof *c);
  c->n_vocab = 256;
  c->hidden = 192;
  c->n_heads = 2;
  c->n_layers = 6;
  c->ffn = 768;
  c->ffn_kernel = 3;
  c->window = 4;
  c->inter = 192;
  c->n_flows = 4;
  c->wn_layers = 4;
  c->wn_kernel = 5;
  c->n_rb = 3;
  c->sample_rate = 22050;
  c->dp_present = 1; c->dp_kernel = 3; c->dp_dds_layers = 3; c->dp_n_flows = 4; c->dp_bins = 10; c->dp_tail_bound = 5.0f;
  if (quality == 0) {  // medium: ResBlock2, 256 ch, rates 8,8,4
    c->up_initial = 256;
    c->n_ups = 3;
    const int r[3] = {8, 8, 4}, k[3] = {16, 16, 8};
    for (int i = 0; i < 3; i++) { c->up_rates[i] = r[i]; c->up_kernels[i] = k[i]; }
    c->resblock_type = 2;
    const int rk[3] = {3, 5, 7};
    const int rd[3][2] = {{1, 2}, {2, 6}, {3, 12}};
    c->rb_n_dil = 2;
    for (int j = 0; j < 3; j++) {
      c->rb_kernels[j] = rk[j];
      for (int d = 0; d < 2; d++) c->rb_dilations[j][d] = rd[j][d];
    }
  } else if (quality == 1) {  // high: ResBlock1, 512 ch, rates 8,8,2,2
    c->up_initial = 512;
    c->n_ups = 4;
    const int r[4] = {8, 8, 2, 2}, k[4] = {16, 16, 4, 4};
    for (int i = 0; i < 4; i++) { c->up_rates[i] = r[i]; c->up_kernels[i] = k[i]; }
    c->resblock_type = 1;
    const int rk[3] = {3, 7, 11};
    c->rb_n_dil = 3;
    for (int j = 0; j < 3; j++) {
      c->rb_kernels[j] = rk[j];
      c->rb_dilations[j][0] = 1; c->rb_dilations[j][1] = 3; c->rb_dilations[j][2] = 5;
    }
  } else {
    PH_FAIL(PIPER_HIP_ERR_ARG, "unknown quality preset %d (0 = medium, 1 = high)", quality);
  }
  return PIPER_HIP_OK;
}

namespace ph {
int validate_config(const piper_hip_voice_config* c) {
  if (!c) PH_FAIL(PIPER_HIP_ERR_ARG, "null config");
  if (c->hidden <= 0 || c->n_heads <= 0 || c->hidden % c->n_heads) PH_FAIL(PIPER_HIP_ERR_SHAPE, "hidden %% n_heads != 0");
  if (c->inter <= 0 || c->inter % 2) PH_FAIL(PIPER_HIP_ERR_SHAPE, "inter must be even");
  if (c->n_ups < 1 || c->n_ups > PIPER_HIP_MAX_UPS) PH_FAIL(PIPER_HIP_ERR_SHAPE, "n_ups out of range");
  if (c->n_rb < 1 || c->n_rb > PIPER_HIP_MAX_RB) PH_FAIL(PIPER_HIP_ERR_SHAPE, "n_rb out of range");
  if (c->rb_n_dil < 1 || c->rb_n_dil > 3) PH_FAIL(PIPER_HIP_ERR_SHAPE, "rb_n_dil out of range");
  if (c->resblock_type != 1 && c->resblock_type != 2) PH_FAIL(PIPER_HIP_ERR_SHAPE, "resblock_type must be 1 or 2");
  if ((c->up_initial >> c->n_ups) < 1 || c->up_initial % (1 << c->n_ups)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "up_initial not divisible by 2^n_ups");
  for (int u = 0; u < c->n_ups; u++)
    if (c->up_rates[u] < 1 || c->up_kernels[u] < c->up_rates[u] || (c->up_kernels[u] - c->up_rates[u]) % 2)
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "upsampler %d: kernel/rate unsupported", u);
  if (c->n_layers < 0 || c->n_flows < 0 || c->wn_layers < 1 || c->n_vocab < 1 || c->window < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "bad counts");
  if (!(c->wn_kernel & 1) || !(c->ffn_kernel >= 1)) PH_FAIL(PIPER_HIP_ERR_SHAPE, "wn_kernel must be odd");
  // hard bounds: the geometry can come from an untrusted file (piper_hip_onnx_infer_config), and every count below sizes an
  // allocation or a loop
  if (c->hidden > 4096 || c->inter > 4096 || c->n_vocab > (1 << 20) || c->n_layers > 64 || c->n_flows > 32 || c->wn_layers > 32 ||
      c->n_heads > 64 || c->window > 1024 || c->up_initial > 8192)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "voice geometry outside the supported bounds");
  if (c->ffn < 1 || c->ffn > 16384 || c->ffn_kernel > 15 || !(c->ffn_kernel & 1) || c->wn_kernel < 1 || c->wn_kernel > 15)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "ffn %d / ffn_kernel %d / wn_kernel %d unsupported (kernels must be odd, ≤ 15)", c->ffn, c->ffn_kernel, c->wn_kernel);
  for (int u = 0; u < c->n_ups; u++)
    if (c->up_rates[u] > 64 || c->up_kernels[u] > 128) PH_FAIL(PIPER_HIP_ERR_SHAPE, "upsampler %d: rate/kernel too large", u);
  for (int j = 0; j < c->n_rb; j++) {
    if (c->rb_kernels[j] < 1 || c->rb_kernels[j] > 15 || !(c->rb_kernels[j] & 1))
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "resblock %d: kernel %d must be odd and ≤ 15", j, c->rb_kernels[j]);
    for (int d = 0; d < c->rb_n_dil; d++)
      if (c->rb_dilations[j][d] < 1 || c->rb_dilations[j][d] > 64) PH_FAIL(PIPER_HIP_ERR_SHAPE, "resblock %d: dilation %d out of [1,64]", j, c->rb_dilations[j][d]);
  }
  if (c->sample_rate < 0 || c->sample_rate > 384000) PH_FAIL(PIPER_HIP_ERR_SHAPE, "sample_rate out of range");
  if (c->dp_present) {
    if (c->dp_present != 1 || c->dp_kernel < 1 || c->dp_kernel > 7 || !(c->dp_kernel & 1) || c->dp_dds_layers < 1 || c->dp_dds_layers > 4 ||
        c->dp_n_flows < 2 || c->dp_n_flows > 8 || c->dp_bins < 2 || c->dp_bins > 32 || !(c->dp_tail_bound > 0.0f) || c->dp_tail_bound > 100.0f)
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "duration predictor geometry unsupported");
    // what the predictor's kernels cover (dp.hip: dds_layer_eligible, kMaxBins) — checked HERE so that a voice inferred from an
    // untrusted .onnx is refused at piper_hip_voice_create, not at its first predict
    if (c->dp_bins > 16) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "duration predictor: %d spline bins (the spline kernel covers at most 16)", c->dp_bins);
    if (c->hidden < 16 || c->hidden > 256)
      PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "duration predictor: hidden = %d (the DDSConv layer kernel covers 16 … 256 channels); load the voice with dp_present = 0 "
                                         "and supply durations", c->hidden);
  }
  return PIPER_HIP_OK;
}
}  // namespace ph

PH_EXPORT int piper_hip_voice_blob_floats(const piper_hip_voice_config* cfg, size_t* n_floats) {
  int rc = ph::validate_config(cfg);
  if (rc) return rc;
  if (!n_floats) PH_FAIL(PIPER_HIP_ERR_ARG, "null n_floats");
  *n_floats = piper_hip_layout_walk(cfg, nullptr, nullptr);
  return PIPER_HIP_OK;
}

namespace {
struct LayoutOut {
  piper_hip_tensor_info* out;
  int max, n;
};
void layout_visit(const piper_tensor_desc* d, void* user) {
  LayoutOut* lo = (LayoutOut*)user;
  if (lo->out && lo->n < lo->max) {
    piper_hip_tensor_info* t = &lo->out[lo->n];
    memset(t, 0, sizeof *t);
    snprintf(t->name, sizeof t->name, "%s", d->name);
    t->kind = (int32_t)d->kind;
    t->rank = d->rank;
    for (int i = 0; i < 3; i++) t->shape[i] = d->shape[i];
    t->fan_in = d->fan_in;
    t->offset = d->offset;
    t->count = d->count;
  }
  lo->n++;
}

inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct SynthOut {
  float* blob;
  uint64_t seed;
  int index;
};
void synth_visit(const piper_tensor_desc* d, void* user) {
  SynthOut* so = (SynthOut*)user;
  const uint64_t ts = so->seed ^ ((uint64_t)(so->index + 1) * 0xD1B54A32D192ED03ull);
  so->index++;
  float scale = 0.f, base = 0.f;
  switch (d->kind) {
    case PIPER_T_WEIGHT:
    case PIPER_T_EMB: scale = (float)std::sqrt(3.0 / (double)(d->fan_in > 0 ? d->fan_in : 1)); break;
    case PIPER_T_BIAS: scale = (float)(0.01 * std::sqrt(3.0)); break;
    case PIPER_T_GAMMA: scale = 0.1f; base = 1.0f; break;
    case PIPER_T_BETA: scale = 0.1f; break;
  }
  float* p = so->blob + d->offset;
  for (size_t j = 0; j < d->count; j++) {
    const uint64_t z = mix64(ts + (uint64_t)(j + 1) * 0x9E3779B97F4A7C15ull);
    const float u = (float)(z >> 40) * 5.9604644775390625e-08f;  // 2^-24
    const float v = (2.0f * u - 1.0f) * scale;
    p[j] = d->kind == PIPER_T_GAMMA ? base + v : v;
  }
}
}  // namespace

PH_EXPORT int piper_hip_voice_blob_layout(const piper_hip_voice_config* cfg, piper_hip_tensor_info* out, int max_entries,
                                          int* n_entries) {
  int rc = ph::validate_config(cfg);
  if (rc) return rc;
  LayoutOut lo{out, max_entries, 0};
  piper_hip_layout_walk(cfg, layout_visit, &lo);
  if (n_entries) *n_entries = lo.n;
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_synthetic_blob(const piper_hip_voice_config* cfg, uint64_t seed, float* host_blob,
                                             size_t n_floats) {
  int rc = ph::validate_config(cfg);
  if (rc) return rc;
  if (!host_blob) PH_FAIL(PIPER_HIP_ERR_ARG, "null blob");
  const size_t need = piper_hip_layout_walk(cfg, nullptr, nullptr);
  if (n_floats < need) PH_FAIL(PIPER_HIP_ERR_SHAPE, "blob too small: %zu < %zu floats", n_floats, need);
  SynthOut so{host_blob, seed, 0};
  piper_hip_layout_walk(cfg, synth_visit, &so);
  return PIPER_HIP_OK;
}
