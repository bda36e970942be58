// attprobe2 — phase stamps (s_memtime) of rel_attention_lds_kernel through the op-level entry point (tools/probe; not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_ATT_LDS_TRACE -x hip tools/probe/attprobe2.cpp
//        piper-swift_amd/csrc/attention.hip piper-swift_amd/csrc/context.cpp -o attprobe2      usage: attprobe2 [T]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/piper_hip.h"
void ph_att_set_trace(unsigned long long* buf);
namespace ph { int pack_conv_weights(hipStream_t, const float*, int, int, int, float*, int) { return 0; } size_t packed_conv_floats(int, int, int, int) { return 0; } }  // only piper_hip_attention_block_f32 needs it

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 112, H = 2, d = 96, w = 4;
  piper_hip_ctx* ctx = nullptr;
  if (piper_hip_create(0, &ctx)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  const size_t n = (size_t)H * d * T;
  std::vector<float> h(n);
  for (size_t i = 0; i < n; i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  float *q, *k, *v, *ek, *ev, *out = nullptr;
  hipMalloc(&q, n * 4); hipMalloc(&k, n * 4); hipMalloc(&v, n * 4); hipMalloc(&ek, 9 * d * 4); hipMalloc(&ev, 9 * d * 4);
  hipMemcpy(q, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(k, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(ek, 0, 9 * d * 4); hipMemset(ev, 0, 9 * d * 4);
  for (int i = 0; i < 3; i++)
    if (piper_hip_rel_attention_f32(ctx, q, k, v, ek, ev, 1, H, d, T, w, &out, nullptr)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  const int blocks = ((T + 15) / 16) * H, waves = 8;
  const size_t nst = (size_t)blocks * waves * 8;
  unsigned long long* tb;
  hipMalloc(&tb, nst * 8);
  hipMemset(tb, 0, nst * 8);
  ph_att_set_trace(tb);
  piper_hip_rel_attention_f32(ctx, q, k, v, ek, ev, 1, H, d, T, w, &out, nullptr);
  hipDeviceSynchronize();
  ph_att_set_trace(nullptr);
  std::vector<unsigned long long> t(nst);
  hipMemcpy(t.data(), tb, nst * 8, hipMemcpyDeviceToHost);
  const char* names[7] = {"", "K fetch + q strip + commit", "barrier", "rel-K logits (wave 7) + scores + barrier", "softmax", "V commit + barrier", "P·V + rel-V + store"};
  double sum[7] = {0};
  int cnt = 0;
  for (size_t wv = 0; wv < (size_t)blocks * waves; wv++) {
    const unsigned long long* s = &t[wv * 8];
    if (!s[0] || !s[6]) continue;
    for (int p = 1; p <= 6; p++) sum[p] += (double)(s[p] - s[p - 1]);
    cnt++;
  }
  printf("rel_attention_lds_kernel T=%d: %d waves traced\n", T, cnt);
  double tot = 0;
  for (int p = 1; p <= 6; p++) { printf("   %-44s %8.1f ticks\n", names[p], sum[p] / std::max(cnt, 1)); tot += sum[p] / std::max(cnt, 1); }
  printf("   total %.1f ticks per wave\n", tot);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 50; i++) piper_hip_rel_attention_f32(ctx, q, k, v, ek, ev, 1, H, d, T, w, &out, nullptr);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("   %.2f us per call (eager, blocking entry point)\n", ms * 20.0);
  return 0;
}
