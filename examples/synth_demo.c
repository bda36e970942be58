/* synth_demo.c — the C-ABI used from plain C, the way a Swift/C/C++ host binds it:
 *   create context → synthetic voice (or a Piper .onnx) → one utterance → 16-bit WAV.
 * build: gcc -std=c99 -Iinclude examples/synth_demo.c -Lpiper-swift_amd/lib -lpiper_hip -Wl,-rpath,$PWD/piper-swift_amd/lib -o synth_demo
 * run:   ./synth_demo out.wav [factor] [voice.onnx | -] [predict]
 *        `predict`: frames per id from the voice's duration predictor and both noise tensors drawn on the device (seed 1234) —
 *        PiperMetalRuntime.synthesize(phonemeIDs:noiseScale:lengthScale:noiseW:) — instead of 3 pinned frames per id and zero noise
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "piper_hip.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != PIPER_HIP_OK) {                                                           \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, piper_hip_last_error());       \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

int main(int argc, char** argv) {
  const char* out = argc > 1 ? argv[1] : "out.wav";
  const int factor = argc > 2 ? atoi(argv[2]) : 1;
  const char* onnx = (argc > 3 && strcmp(argv[3], "-") != 0) ? argv[3] : NULL;
  const int predict = argc > 4 && strcmp(argv[4], "predict") == 0;
  static const int64_t fixture[14] = {1, 20, 0, 120, 0, 61, 0, 24, 0, 59, 0, 100, 0, 2}; /* bench/fixtures/test_summary.json:8 */

  piper_hip_voice_config cfg;
  size_t n_floats = 0;
  float* blob = NULL;
  if (onnx) {
    piper_hip_onnx* m = NULL;
    CHECK(piper_hip_onnx_open(onnx, &m));
    CHECK(piper_hip_onnx_infer_config(m, &cfg));
    CHECK(piper_hip_voice_blob_floats(&cfg, &n_floats));
    blob = (float*)malloc(n_floats * sizeof(float));
    CHECK(piper_hip_onnx_build_blob(m, &cfg, blob, n_floats));
    piper_hip_onnx_close(m);
  } else {
    CHECK(piper_hip_voice_config_preset(0, &cfg));
    CHECK(piper_hip_voice_blob_floats(&cfg, &n_floats));
    blob = (float*)malloc(n_floats * sizeof(float));
    CHECK(piper_hip_voice_synthetic_blob(&cfg, 1234, blob, n_floats));
  }

  piper_hip_ctx* ctx = NULL;
  piper_hip_voice* voice = NULL;
  CHECK(piper_hip_create(0, &ctx));
  CHECK(piper_hip_voice_create(ctx, &cfg, blob, 0, &voice));
  free(blob);

  const int T = 14 * factor;
  int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * T);
  int32_t* dur = (int32_t*)malloc(sizeof(int32_t) * T);
  for (int i = 0; i < T; i++) { ids[i] = fixture[i % 14]; dur[i] = 3; }
  piper_hip_utterance u;
  memset(&u, 0, sizeof u);
  u.phoneme_ids = ids; u.t = T; u.durations = dur; u.noise = NULL; u.noise_scale = 0.667f;
  int64_t n = 0, got = 0;
  float* audio = NULL;
  if (predict) {
    u.durations = NULL; u.noise_mode = PIPER_HIP_NOISE_DEVICE; u.seed = 1234; u.length_scale = 1.0f; u.noise_w = 0.8f;
    int rcp = piper_hip_voice_prepare(voice, &u, 0); /* runs the duration predictor, uploads the inputs */
    if (rcp < 0) { fprintf(stderr, "prepare failed (%d): %s\n", rcp, piper_hip_last_error()); return 1; }
    CHECK(piper_hip_voice_prepared_samples(voice, 0, NULL, 0, &n));
    audio = (float*)malloc(sizeof(float) * (size_t)n);
    CHECK(piper_hip_voice_launch(voice, 0));
    CHECK(piper_hip_voice_collect(voice, 0, audio, n));
    got = n;
  } else {
    n = piper_hip_voice_num_samples(voice, &u);
    if (n <= 0) { fprintf(stderr, "bad utterance: %s\n", piper_hip_last_error()); return 1; }
    audio = (float*)malloc(sizeof(float) * (size_t)n);
    CHECK(piper_hip_voice_synthesize(voice, &u, audio, n, &got));
  }
  double ms = 0.0;
  CHECK(piper_hip_voice_last_gpu_ms(voice, 0, &ms));
  CHECK(piper_hip_wav_write(out, audio, (size_t)got, cfg.sample_rate));
  printf("%lld samples (%.3f s at %d Hz) in %.3f ms on the GPU -> %s\n", (long long)got, (double)got / cfg.sample_rate, cfg.sample_rate, ms, out);
  free(audio); free(ids); free(dur);
  piper_hip_voice_destroy(voice);
  piper_hip_destroy(ctx);
  return 0;
}
