/* piper_hip_cli.c — the bench / one-shot modes of the reference's command line (Sources/PiperCLI/PiperCLI.swift) over the C-ABI, in plain C.
 *
 *   --scale-bench            PiperCLI.runScaleBench (PiperCLI.swift:381-551): the fixture utterance tiled by each scale factor, truncated at
 *                            --max-phonemes, `--warmup` untimed + `--iters` timed calls of synthesize per factor, ONE JSON object on stdout with the
 *                            reference's keys (backend, mode, model_path, sample_rate, warmup, iters, max_phonemes, scale_factors,
 *                            base_test_phonemes, results[{factor, phoneme_count, ms_mean, ms_p50, ms_p95, ms_max}]); with
 *                            PIPER_BENCH_GPU_TIMING=1 also gpu_ms_mean, cpu_user_ms_mean, cpu_sys_ms_mean, gpu_busy_fraction_mean, max_rss_max
 *                            (PiperCLI.swift:288, 395, 524-537). Same flags: --warmup (1) --iters (3) --scale-factors (1,2,4,8,16) --max-phonemes (4096)
 *                            --model voice.onnx [--config voice.onnx.json].
 *   --phoneme-ids a,b,c …    one utterance → --output file.wav (16-bit, WavFileWriter.swift:20-60); the espeak-ng front end of the reference CLI is
 *                            out of scope, so the ids are the input.
 * Without --model the synthetic voice of the tests is used (--quality medium|high); its duration predictor has random weights, so frames per id are
 * pinned to --pin-frames (3, the bench's convention) unless --predict asks for the predictor (the only mode a real voice has).
 *
 * build: gcc -std=c99 -O2 -Iinclude examples/piper_hip_cli.c -Lpiper-swift_amd/lib -lpiper_hip -Wl,-rpath,$PWD/piper-swift_amd/lib -o piper_hip_cli
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <time.h>

#include "piper_hip.h"

static const int64_t kFixture[14] = {1, 20, 0, 120, 0, 61, 0, 24, 0, 59, 0, 100, 0, 2}; /* bench/fixtures/test_summary.json:8 */

#define CHECK(call)                                                                \
  do {                                                                             \
    int rc_ = (call);                                                              \
    if (rc_ < 0) {                                                                 \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, piper_hip_last_error()); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

static const char* arg_value(int argc, char** argv, const char* key) {
  for (int i = 1; i + 1 < argc; i++)
    if (strcmp(argv[i], key) == 0) return argv[i + 1];
  return NULL;
}
static int has_flag(int argc, char** argv, const char* key) {
  for (int i = 1; i < argc; i++)
    if (strcmp(argv[i], key) == 0) return 1;
  return 0;
}
static int parse_csv_i64(const char* s, int64_t* out, int cap) {
  int n = 0;
  while (*s && n < cap) {
    char* end = NULL;
    const long long v = strtoll(s, &end, 10);
    if (end == s) return -1;
    out[n++] = v;
    s = end;
    while (*s == ',' || *s == ' ') s++;
  }
  return n;
}
static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static double tv_ms(struct timeval tv) { return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3; }
static int cmp_double(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }
/* linear-interpolated percentile of the sorted sample, as PiperCLI.swift:428-436 */
static double percentile(double* sorted, int n, double p) {
  const double k = (n - 1) * (p / 100.0);
  const int f = (int)k;
  const int c = (k > f) ? f + 1 : f;
  return f == c ? sorted[f] : sorted[f] + (sorted[c] - sorted[f]) * (k - f);
}

typedef struct {
  piper_hip_voice* voice;
  float noise_scale, length_scale, noise_w;
  int predict, pin_frames;
  float* audio;
  int64_t audio_cap;
} runner;

/* PiperMetalRuntime.synthesize(phonemeIDs:noiseScale:lengthScale:noiseW:) on slot 0; *samples = waveform length */
static int run_one(runner* r, const int64_t* ids, int t, int64_t* samples) {
  piper_hip_utterance u;
  memset(&u, 0, sizeof u);
  u.phoneme_ids = ids; u.t = t; u.noise_scale = r->noise_scale; u.seed = 1234; /* the reference's fixed seed (GraphExecutor.swift:2658) */
  u.length_scale = r->length_scale; u.noise_w = r->noise_w; u.noise_mode = PIPER_HIP_NOISE_DEVICE;
  int32_t* dur = NULL;
  if (!r->predict) {
    dur = (int32_t*)malloc(sizeof(int32_t) * (size_t)t);
    for (int i = 0; i < t; i++) dur[i] = r->pin_frames;
    u.durations = dur;
  }
  int rc = piper_hip_voice_prepare(r->voice, &u, 0);
  free(dur);
  if (rc < 0) return rc;
  int64_t n = 0;
  if ((rc = piper_hip_voice_prepared_samples(r->voice, 0, NULL, 0, &n)) < 0) return rc;
  if (n > r->audio_cap) {
    free(r->audio);
    r->audio = (float*)malloc(sizeof(float) * (size_t)n);
    r->audio_cap = n;
  }
  if ((rc = piper_hip_voice_launch(r->voice, 0)) < 0) return rc;
  if ((rc = piper_hip_voice_collect(r->voice, 0, r->audio, n)) < 0) return rc;
  *samples = n;
  return 0;
}

int main(int argc, char** argv) {
  const char* model = arg_value(argc, argv, "--model");
  const char* config = arg_value(argc, argv, "--config");
  const char* quality = arg_value(argc, argv, "--quality");
  const int scale_bench = has_flag(argc, argv, "--scale-bench");
  const char* ids_arg = arg_value(argc, argv, "--phoneme-ids");
  if (!scale_bench && !ids_arg) {
    fprintf(stderr, "usage: %s --scale-bench [--warmup N] [--iters N] [--scale-factors 1,2,4,8,16] [--max-phonemes N]\n"
                    "       %s --phoneme-ids 1,20,0,… --output out.wav\n"
                    "       common: [--model voice.onnx [--config voice.onnx.json]] [--quality medium|high] [--predict] [--pin-frames N]\n", argv[0], argv[0]);
    return 2;
  }

  piper_hip_voice_config cfg;
  size_t n_floats = 0;
  float* blob = NULL;
  runner r;
  memset(&r, 0, sizeof r);
  r.noise_scale = 0.667f; r.length_scale = 1.0f; r.noise_w = 0.8f; /* PiperConfig defaults; the voice's .onnx.json overrides them */
  r.predict = has_flag(argc, argv, "--predict");
  r.pin_frames = arg_value(argc, argv, "--pin-frames") ? atoi(arg_value(argc, argv, "--pin-frames")) : 3;
  if (model) {
    piper_hip_onnx* m = NULL;
    CHECK(piper_hip_onnx_open(model, &m));
    CHECK(piper_hip_onnx_infer_config(m, &cfg));
    CHECK(piper_hip_voice_blob_floats(&cfg, &n_floats));
    blob = (float*)malloc(n_floats * sizeof(float));
    CHECK(piper_hip_onnx_build_blob(m, &cfg, blob, n_floats)); /* verifies the graph first */
    piper_hip_onnx_close(m);
    r.predict = 1; /* a real voice decides its own durations */
    if (config) { /* PiperConfig (PiperConfig.swift:3-47): sample rate and the three inference scales */
      FILE* fj = fopen(config, "rb");
      if (!fj) { fprintf(stderr, "cannot open %s\n", config); return 1; }
      fseek(fj, 0, SEEK_END);
      const long len = ftell(fj);
      fseek(fj, 0, SEEK_SET);
      char* text = (char*)malloc((size_t)len + 1);
      if (fread(text, 1, (size_t)len, fj) != (size_t)len) { fprintf(stderr, "cannot read %s\n", config); return 1; }
      text[len] = 0;
      fclose(fj);
      piper_hip_piper_json_info info;
      const int jrc = piper_hip_piper_json(text, &info);
      free(text);
      CHECK(jrc);
      if (info.sample_rate > 0) cfg.sample_rate = info.sample_rate;
      r.noise_scale = info.noise_scale; r.length_scale = info.length_scale; r.noise_w = info.noise_w;
    }
  } else {
    CHECK(piper_hip_voice_config_preset(quality && strcmp(quality, "high") == 0 ? 1 : 0, &cfg));
    CHECK(piper_hip_voice_blob_floats(&cfg, &n_floats));
    blob = (float*)malloc(n_floats * sizeof(float));
    CHECK(piper_hip_voice_synthetic_blob(&cfg, 1234, blob, n_floats));
  }
  piper_hip_ctx* ctx = NULL;
  CHECK(piper_hip_create(0, &ctx));
  CHECK(piper_hip_voice_create(ctx, &cfg, blob, 0, &r.voice));
  free(blob);

  if (!scale_bench) { /* one shot: ids → WAV */
    int64_t ids[4096];
    const int t = parse_csv_i64(ids_arg, ids, 4096);
    const char* out = arg_value(argc, argv, "--output");
    if (t < 1 || !out) { fprintf(stderr, "--phoneme-ids needs a comma-separated list and --output a path\n"); return 2; }
    int64_t n = 0;
    CHECK(run_one(&r, ids, t, &n));
    double gpu = 0.0;
    CHECK(piper_hip_voice_last_gpu_ms(r.voice, 0, &gpu));
    CHECK(piper_hip_wav_write(out, r.audio, (size_t)n, cfg.sample_rate));
    fprintf(stderr, "%lld samples (%.3f s at %d Hz), %.3f ms on the GPU -> %s\n", (long long)n, (double)n / cfg.sample_rate, cfg.sample_rate, gpu, out);
    piper_hip_voice_destroy(r.voice);
    piper_hip_destroy(ctx);
    return 0;
  }

  const int warmup = arg_value(argc, argv, "--warmup") ? atoi(arg_value(argc, argv, "--warmup")) : 1;
  const int iters = arg_value(argc, argv, "--iters") ? atoi(arg_value(argc, argv, "--iters")) : 3;
  const int max_phonemes = arg_value(argc, argv, "--max-phonemes") ? atoi(arg_value(argc, argv, "--max-phonemes")) : 4096;
  int64_t factors[64];
  const int nf = parse_csv_i64(arg_value(argc, argv, "--scale-factors") ? arg_value(argc, argv, "--scale-factors") : "1,2,4,8,16", factors, 64);
  if (nf < 1 || iters < 1 || warmup < 0 || max_phonemes < 1 || max_phonemes > 4096) { fprintf(stderr, "bad --scale-factors / --iters / --warmup / --max-phonemes\n"); return 2; }
  const char* tenv = getenv("PIPER_BENCH_GPU_TIMING");
  const int want_timings = tenv && strcmp(tenv, "1") == 0;

  printf("{\n  \"backend\": \"piper-hip\",\n  \"base_test_phonemes\": 14,\n  \"iters\": %d,\n  \"max_phonemes\": %d,\n  \"mode\": \"scale-bench\",\n  \"model_path\": \"%s\",\n  \"results\": [\n",
         iters, max_phonemes, model ? model : (quality && strcmp(quality, "high") == 0 ? "synthetic:high" : "synthetic:medium"));
  int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * 4096);
  double* wall = (double*)malloc(sizeof(double) * (size_t)iters);
  for (int fi = 0; fi < nf; fi++) {
    const int f = factors[fi] < 1 ? 1 : (int)factors[fi];
    long long want = 14LL * f;
    const int t = (int)(want > max_phonemes ? max_phonemes : want); /* tiled, then truncated at --max-phonemes (PiperCLI.swift:467-473) */
    for (int i = 0; i < t; i++) ids[i] = kFixture[i % 14];
    int64_t n = 0;
    for (int w = 0; w < warmup; w++) CHECK(run_one(&r, ids, t, &n));
    double gpu_sum = 0.0, user_sum = 0.0, sys_sum = 0.0, rss_max = 0.0;
    for (int it = 0; it < iters; it++) {
      struct rusage ru0, ru1;
      getrusage(RUSAGE_SELF, &ru0);
      const double t0 = now_ms();
      CHECK(run_one(&r, ids, t, &n));
      wall[it] = now_ms() - t0;
      getrusage(RUSAGE_SELF, &ru1);
      double g = 0.0;
      CHECK(piper_hip_voice_last_gpu_ms(r.voice, 0, &g));
      gpu_sum += g;
      user_sum += tv_ms(ru1.ru_utime) - tv_ms(ru0.ru_utime);
      sys_sum += tv_ms(ru1.ru_stime) - tv_ms(ru0.ru_stime);
      if ((double)ru1.ru_maxrss > rss_max) rss_max = (double)ru1.ru_maxrss;
    }
    double mean = 0.0, mx = 0.0;
    for (int it = 0; it < iters; it++) { mean += wall[it]; if (wall[it] > mx) mx = wall[it]; }
    mean /= iters;
    qsort(wall, (size_t)iters, sizeof(double), cmp_double);
    printf("    {\"factor\": %d, \"phoneme_count\": %d, \"ms_mean\": %.4f, \"ms_p50\": %.4f, \"ms_p95\": %.4f, \"ms_max\": %.4f", f, t, mean, percentile(wall, iters, 50.0),
           percentile(wall, iters, 95.0), mx);
    if (want_timings)
      printf(", \"gpu_ms_mean\": %.4f, \"cpu_user_ms_mean\": %.4f, \"cpu_sys_ms_mean\": %.4f, \"gpu_busy_fraction_mean\": %.4f, \"max_rss_max\": %.0f", gpu_sum / iters,
             user_sum / iters, sys_sum / iters, mean > 0.0 ? gpu_sum / iters / mean : 0.0, rss_max);
    /* beyond the reference's keys: what the audio was worth (audio-s per wall-s) */
    printf(", \"audio_sec\": %.4f, \"rtf_inv\": %.1f}%s\n", (double)n / cfg.sample_rate, mean > 0.0 ? (double)n / cfg.sample_rate / (mean * 1e-3) : 0.0, fi + 1 < nf ? "," : "");
  }
  printf("  ],\n  \"sample_rate\": %d,\n  \"scale_factors\": [", cfg.sample_rate);
  for (int fi = 0; fi < nf; fi++) printf("%s%lld", fi ? ", " : "", (long long)factors[fi]);
  printf("],\n  \"warmup\": %d\n}\n", warmup);
  free(ids); free(wall); free(r.audio);
  piper_hip_voice_destroy(r.voice);
  piper_hip_destroy(ctx);
  return 0;
}
