// audio_io.cpp — float waveform → 16-bit PCM / mono WAV file, the last hop after synthesize (host-only).
//
// Same sample conversion as the reference's CLI writer (WavFileWriter.swift:20-30): clamp to [−1, 1] in double, multiply by
// 32767.0, truncate toward zero; header layout of its writeHeaderPlaceholder / finalize (RIFF, fmt chunk 16 bytes, PCM,
// 1 channel, 16 bits, data chunk).
#include <cstdio>

#include "common.h"

using namespace ph;

PH_EXPORT int piper_hip_pcm16_from_f32(const float* samples, size_t n, int16_t* pcm) {
  if ((!samples || !pcm) && n) PH_FAIL(PIPER_HIP_ERR_ARG, "pcm16_from_f32: null argument");
  for (size_t i = 0; i < n; i++) {
    double x = (double)samples[i];
    x = x != x ? 0.0 : (x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x));  // NaN → silence rather than undefined conversion
    const long v = (long)(x * 32767.0);                            // C cast truncates toward zero like Swift's Int(_:)
    pcm[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_wav_write(const char* path, const float* samples, size_t n, int32_t sample_rate) {
  if (!path || (!samples && n)) PH_FAIL(PIPER_HIP_ERR_ARG, "wav_write: null argument");
  if (sample_rate <= 0) PH_FAIL(PIPER_HIP_ERR_ARG, "wav_write: sample_rate %d", sample_rate);
  if (n > 0x7fffffffu / 2) PH_FAIL(PIPER_HIP_ERR_SHAPE, "wav_write: %zu samples do not fit a RIFF file", n);
  FILE* f = fopen(path, "wb");
  if (!f) PH_FAIL(PIPER_HIP_ERR_ARG, "wav_write: cannot create '%s'", path);
  const uint32_t data_bytes = (uint32_t)(n * 2), riff = 36 + data_bytes, rate = (uint32_t)sample_rate, byte_rate = rate * 2;
  uint8_t h[44];
  auto u32 = [&](int at, uint32_t v) { for (int i = 0; i < 4; i++) h[at + i] = (uint8_t)(v >> (8 * i)); };
  auto u16 = [&](int at, uint32_t v) { h[at] = (uint8_t)v; h[at + 1] = (uint8_t)(v >> 8); };
  memcpy(h, "RIFF", 4); u32(4, riff); memcpy(h + 8, "WAVE", 4);
  memcpy(h + 12, "fmt ", 4); u32(16, 16); u16(20, 1); u16(22, 1); u32(24, rate); u32(28, byte_rate); u16(32, 2); u16(34, 16);
  memcpy(h + 36, "data", 4); u32(40, data_bytes);
  bool ok = fwrite(h, 1, 44, f) == 44;
  std::vector<int16_t> pcm(n < 65536 ? n : 65536);
  for (size_t at = 0; ok && at < n; at += pcm.size()) {
    const size_t m = n - at < pcm.size() ? n - at : pcm.size();
    piper_hip_pcm16_from_f32(samples + at, m, pcm.data());
    ok = fwrite(pcm.data(), 2, m, f) == m;  // little-endian host (x86-64)
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) PH_FAIL(PIPER_HIP_ERR_ARG, "wav_write: short write to '%s'", path);
  return PIPER_HIP_OK;
}
