// fuzz_onnx.cpp — sanitizer harness for the ONNX reader (host-only): every prefix of a seed model and N random byte
// mutations go through open → infer_config → verify_graph → build_blob. Built with g++ -fsanitize=address,undefined (CPU only; GPU
// sanitizers are not available on the pool). The library sources are compiled in directly; error text goes to a stub.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../piper-swift_amd/csrc/common.h"

namespace ph {
static thread_local char g_err[512];
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace ph

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {  // SplitMix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int try_model(const uint8_t* data, size_t n) {
  piper_hip_onnx* m = nullptr;
  if (piper_hip_onnx_open_memory(data, n, &m) != PIPER_HIP_OK) return 0;
  int parsed = 1;
  int64_t ir = 0, op = 0;
  int nn = 0, ni = 0;
  piper_hip_onnx_counts(m, &ir, &op, &nn, &ni);
  for (int i = 0; i < ni; i++) {
    piper_hip_onnx_tensor_info info;
    piper_hip_onnx_initializer(m, i, &info);
    if (info.count > 0 && info.count < 4096) {
      std::vector<float> tmp((size_t)info.count);
      (void)piper_hip_onnx_read_f32(m, i, tmp.data(), tmp.size());
    }
  }
  piper_hip_voice_config cfg;
  if (piper_hip_onnx_infer_config(m, &cfg) == PIPER_HIP_OK) {
    size_t nf = 0;
    if (piper_hip_voice_blob_floats(&cfg, &nf) == PIPER_HIP_OK && nf < (64u << 20)) {
      std::vector<float> blob(nf);
      (void)piper_hip_onnx_build_blob(m, &cfg, blob.data(), nf);            // graph verifier, then the initializers
      (void)piper_hip_onnx_build_blob_unchecked(m, &cfg, blob.data(), nf);  // the initializers alone
      parsed = 2;
    }
  }
  piper_hip_onnx_close(m);
  return parsed;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: fuzz_onnx seed.onnx [mutations]\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> seed((size_t)n);
  if (fread(seed.data(), 1, (size_t)n, f) != (size_t)n) return 2;
  fclose(f);
  const int mutations = argc > 2 ? atoi(argv[2]) : 2000;
  if (try_model(seed.data(), seed.size()) != 2) { fprintf(stderr, "seed does not load\n"); return 3; }
  int ok = 0, full = 0;
  const size_t step = seed.size() > 4096 ? seed.size() / 1024 : 1;
  for (size_t len = 0; len < seed.size(); len += step) {  // prefixes (copied so ASAN sees the true end of the buffer)
    std::vector<uint8_t> p(seed.begin(), seed.begin() + len);
    ok += try_model(p.data(), p.size()) > 0;
  }
  for (int it = 0; it < mutations; it++) {
    std::vector<uint8_t> m = seed;
    const int edits = 1 + (int)(rnd() % 4);
    for (int e = 0; e < edits; e++) {
      // bias the edits toward the structural bytes at the front of each message: headers, tags, lengths
      const size_t pos = (rnd() & 1) ? rnd() % m.size() : rnd() % (m.size() < 4096 ? m.size() : 4096);
      switch (rnd() % 3) {
        case 0: m[pos] = (uint8_t)rnd(); break;
        case 1: m[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
        default: m[pos] = 0xff; break;
      }
    }
    const int r = try_model(m.data(), m.size());
    ok += r > 0;
    full += r == 2;
  }
  printf("fuzz_onnx: %zu-byte seed, %d mutations: %d inputs parsed, %d built a blob, no sanitizer report\n", seed.size(), mutations, ok, full);
  return 0;
}
