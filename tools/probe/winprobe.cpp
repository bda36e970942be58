// winprobe — phase trace (s_memtime) and timing of conv_win_kernel on a ConvTranspose (tools/probe; not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_WIN_TRACE -x hip tools/probe/winprobe.cpp
//        piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o winprobe
// usage: winprobe Cin Cout K stride L [avg3]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../piper-swift_amd/csrc/conv_win.h"

using namespace ph;
namespace ph { void conv_win_set_trace(unsigned long long* buf); }

int main(int argc, char** argv) {
  const int Cin = argc > 1 ? atoi(argv[1]) : 128, Cout = argc > 2 ? atoi(argv[2]) : 64, K = argc > 3 ? atoi(argv[3]) : 16, st = argc > 4 ? atoi(argv[4]) : 8;
  const int L = argc > 5 ? atoi(argv[5]) : 2688;
  const bool avg = argc > 6;
  piper_hip_ctx* ctx = nullptr;
  if (piper_hip_create(0, &ctx)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  hipStream_t s;
  (void)hipStreamCreate(&s);
  float *x, *x2, *x3, *y, *w, *wp, *b;
  (void)hipMalloc(&x, (size_t)Cin * L * 4); (void)hipMalloc(&x2, (size_t)Cin * L * 4); (void)hipMalloc(&x3, (size_t)Cin * L * 4);
  (void)hipMalloc(&y, (size_t)Cout * L * st * 4);
  (void)hipMalloc(&w, (size_t)Cin * Cout * K * 4); (void)hipMalloc(&b, Cout * 4);
  (void)hipMalloc(&wp, packed_convt_win_floats(Cin, Cout, K, st) * 4);
  std::vector<float> h((size_t)Cin * L);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(x2, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(x3, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hw((size_t)Cin * Cout * K, 0.01f);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(b, 0, Cout * 4);
  pack_convt_weights_win(s, w, Cin, Cout, K, st, (K - st) / 2, wp);
  ConvWinArgs a;
  a.x = x; if (avg) { a.x2 = x2; a.x3 = x3; }
  a.w4 = wp; a.bias = b; a.y = y; a.pro_alpha = 0.1f; a.N = 1; a.Cin = Cin; a.Cout = Cout; a.K = K; a.Lin = L; a.Lout = L; a.y_len = L * st;
  a.ct_stride = st; a.ct_pad = (K - st) / 2;
  (void)hipStreamSynchronize(s);
  for (int i = 0; i < 3; i++) launch_conv_win(ctx, s, a);
  (void)hipStreamSynchronize(s);
  const size_t nst = (size_t)1024 * 4 * 8;
  unsigned long long* tb;
  (void)hipMalloc(&tb, nst * 8);
  (void)hipMemset(tb, 0, nst * 8);
  conv_win_set_trace(tb);
  launch_conv_win(ctx, s, a);
  (void)hipStreamSynchronize(s);
  conv_win_set_trace(nullptr);
  std::vector<unsigned long long> t(nst);
  (void)hipMemcpy(t.data(), tb, nst * 8, hipMemcpyDeviceToHost);
  const char* names[7] = {"", "args + ring prologue", "staging loads + LDS stores", "barrier", "K loop", "K-split reduce", "epilogue"};
  double sum[7] = {0};
  int cnt = 0;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t wv = 0; wv < (size_t)1024 * 4; wv++) {
    const unsigned long long* q = &t[wv * 8];
    if (!q[0] || !q[4]) continue;
    for (int k = 1; k <= 4; k++) sum[k] += (double)(q[k] - q[k - 1]);
    if (q[5]) sum[5] += (double)(q[5] - q[4]);
    if (q[6]) sum[6] += (double)(q[6] - q[5]);
    t0 = std::min(t0, q[0]); t1 = std::max(t1, q[6] ? q[6] : q[4]);
    cnt++;
  }
  printf("convT Cin=%d Cout=%d K=%d s=%d L=%d%s: %d waves traced, kernel span %.0f cycles\n", Cin, Cout, K, st, L, avg ? " avg3" : "", cnt, (double)(t1 - t0));
  for (int k = 1; k < 7; k++) printf("    %-28s %9.1f cycles\n", names[k], sum[k] / std::max(cnt, 1));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 50;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; i++) launch_conv_win(ctx, s, a);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / reps, fl = 2.0 * Cin * Cout * (double)K * L;
  printf("  %.2f us per launch, %.1f TFLOP/s\n", us, fl / us * 1e-6);
  return 0;
}
