// winprobe2 — phase trace (s_memtime) and timing of conv_win_kernel on the three same-shape ResBlock convs of one launch (tools/probe).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_WIN_TRACE -x hip tools/probe/winprobe2.cpp
//        piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o winprobe2
// usage: winprobe2 C L K0,K1,K2 d0,d1,d2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../piper-swift_amd/csrc/conv_win.h"

using namespace ph;
namespace ph { void conv_win_set_trace(unsigned long long* buf); }

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 128, L = argc > 2 ? atoi(argv[2]) : 2688;
  int K[3] = {3, 5, 7}, D[3] = {1, 1, 1};
  if (argc > 3) sscanf(argv[3], "%d,%d,%d", &K[0], &K[1], &K[2]);
  if (argc > 4) sscanf(argv[4], "%d,%d,%d", &D[0], &D[1], &D[2]);
  piper_hip_ctx* ctx = nullptr;
  if (piper_hip_create(0, &ctx)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  hipStream_t s;
  (void)hipStreamCreate(&s);
  ConvWinArgs a[3];
  double flops = 0;
  float* x;
  (void)hipMalloc(&x, (size_t)C * L * 4);
  std::vector<float> h((size_t)C * L);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int j = 0; j < 3; j++) {
    float *y, *w, *wp, *b;
    (void)hipMalloc(&y, (size_t)C * L * 4);
    (void)hipMalloc(&w, (size_t)C * C * K[j] * 4); (void)hipMalloc(&b, C * 4);
    (void)hipMalloc(&wp, packed_conv_win_floats(C, C, K[j]) * 4);
    std::vector<float> hw((size_t)C * C * K[j], 0.01f);
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(b, 0, C * 4);
    pack_conv_weights_win(s, w, C, C, K[j], wp);
    ConvWinArgs& c = a[j];
    c.x = x; c.w4 = wp; c.bias = b; c.res = x; c.y = y; c.pro_alpha = 0.1f;
    c.N = 1; c.Cin = C; c.Cout = C; c.K = K[j]; c.dil = D[j]; c.padL = (K[j] * D[j] - D[j]) / 2; c.Lin = L; c.Lout = L; c.y_len = L;
    flops += 2.0 * C * C * K[j] * (double)L;
  }
  (void)hipStreamSynchronize(s);
  for (int i = 0; i < 3; i++) launch_conv_win_multi(ctx, s, a, 3);
  (void)hipStreamSynchronize(s);
  const size_t nst = (size_t)1024 * 4 * 8;
  unsigned long long* tb;
  (void)hipMalloc(&tb, nst * 8);
  (void)hipMemset(tb, 0, nst * 8);
  conv_win_set_trace(tb);
  launch_conv_win_multi(ctx, s, a, 3);
  (void)hipStreamSynchronize(s);
  conv_win_set_trace(nullptr);
  std::vector<unsigned long long> t(nst);
  (void)hipMemcpy(t.data(), tb, nst * 8, hipMemcpyDeviceToHost);
  const char* names[7] = {"", "args + ring prologue", "staging loads + LDS stores", "barrier", "K loop", "K-split reduce", "epilogue"};
  double sum[7] = {0};
  int cnt = 0;
  unsigned long long t0 = ~0ull, t1 = 0, last_start = 0;
  for (size_t wv = 0; wv < (size_t)1024 * 4; wv++) {
    const unsigned long long* q = &t[wv * 8];
    if (!q[0] || !q[4]) continue;
    for (int k = 1; k <= 4; k++) sum[k] += (double)(q[k] - q[k - 1]);
    if (q[5]) sum[5] += (double)(q[5] - q[4]);
    if (q[6]) sum[6] += (double)(q[6] - q[5]);
    t0 = std::min(t0, q[0]); t1 = std::max(t1, q[6] ? q[6] : q[4]);
    last_start = std::max(last_start, q[0]);
    cnt++;
  }
  printf("conv x3 C=%d L=%d K=%d,%d,%d d=%d,%d,%d: %d waves traced (first 1024 blocks), span %.0f ticks, last block start at +%.0f\n", C, L, K[0], K[1], K[2], D[0], D[1], D[2], cnt,
         (double)(t1 - t0), (double)(last_start - t0));
  for (int k = 1; k < 7; k++) printf("    %-28s %9.1f ticks\n", names[k], sum[k] / std::max(cnt, 1));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 50;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; i++) launch_conv_win_multi(ctx, s, a, 3);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / reps;
  printf("  %.2f us per launch, %.1f TFLOP/s\n", us, flops / us * 1e-6);
  return 0;
}
