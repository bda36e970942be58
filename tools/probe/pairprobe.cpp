// pairprobe — correctness (vs a plain CPU loop) and timing of rb_pair_kernel (tools/probe; not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DPH_PAIR_VARIANT=n] -x hip tools/probe/pairprobe.cpp
//        piper-swift_amd/csrc/rb_pair.hip piper-swift_amd/csrc/conv_win.hip piper-swift_amd/csrc/context.cpp -o pairprobe
// usage: pairprobe C L K0,K1,K2 da0,da1,da2 db0,db1,db2 [check]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../piper-swift_amd/csrc/conv_win.h"

using namespace ph;
#ifdef PH_PAIR_TRACE
namespace ph { void rb_pair_set_trace(unsigned long long* buf); }
#endif

static void cpu_conv(const std::vector<float>& x, const std::vector<float>& w, const std::vector<float>& b, int C, int L, int K, int d, float alpha,
                     const float* res, std::vector<float>& y) {
  const int pad = (K - 1) * d / 2;
  for (int co = 0; co < C; co++)
    for (int t = 0; t < L; t++) {
      double acc = b[co];
      for (int k = 0; k < K; k++) {
        const int pos = t - pad + k * d;
        if (pos < 0 || pos >= L) continue;
        for (int ci = 0; ci < C; ci++) {
          float v = x[(size_t)ci * L + pos];
          v = v >= 0 ? v : v * alpha;
          acc += (double)w[((size_t)co * C + ci) * K + k] * v;
        }
      }
      y[(size_t)co * L + t] = (float)acc + (res ? res[(size_t)co * L + t] : 0.0f);
    }
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 32, L = argc > 2 ? atoi(argv[2]) : 86016;
  int K[3] = {3, 5, 7}, DA[3] = {1, 2, 3}, DB[3] = {2, 6, 12};
  if (argc > 3) sscanf(argv[3], "%d,%d,%d", &K[0], &K[1], &K[2]);
  if (argc > 4) sscanf(argv[4], "%d,%d,%d", &DA[0], &DA[1], &DA[2]);
  if (argc > 5) sscanf(argv[5], "%d,%d,%d", &DB[0], &DB[1], &DB[2]);
  const bool check = argc > 6;
  piper_hip_ctx* ctx = nullptr;
  if (piper_hip_create(0, &ctx)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  hipStream_t s;
  (void)hipStreamCreate(&s);
  RbPairArgs a[3];
  double flops = 0;
  std::vector<float> hx((size_t)C * L), hw[3][2], hb[3][2];
  for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  float* xd;
  (void)hipMalloc(&xd, hx.size() * 4);
  (void)hipMemcpy(xd, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  float* yd[3];
  for (int j = 0; j < 3; j++) {
    float *wd[2], *wp[2], *bd[2];
    for (int q = 0; q < 2; q++) {
      hw[j][q].resize((size_t)C * C * K[j]);
      hb[j][q].resize(C);
      for (size_t i = 0; i < hw[j][q].size(); i++) hw[j][q][i] = ((float)(((i + 977 * q + 131 * j) * 40503u >> 4) & 0xfff) / 4096.0f - 0.5f) * 0.1f;
      for (int i = 0; i < C; i++) hb[j][q][i] = 0.01f * (float)((i * 7 + q) % 13 - 6);
      (void)hipMalloc(&wd[q], hw[j][q].size() * 4); (void)hipMalloc(&bd[q], C * 4);
      (void)hipMalloc(&wp[q], packed_conv_win_floats(C, C, K[j]) * 4);
      (void)hipMemcpy(wd[q], hw[j][q].data(), hw[j][q].size() * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(bd[q], hb[j][q].data(), C * 4, hipMemcpyHostToDevice);
      pack_conv_weights_win(s, wd[q], C, C, K[j], wp[q]);
    }
    (void)hipMalloc(&yd[j], hx.size() * 4);
    RbPairArgs& c = a[j];
    c.x = xd; c.y = yd[j]; c.wa4 = wp[0]; c.ba = bd[0]; c.wb4 = wp[1]; c.bb = bd[1];
    c.Ka = K[j]; c.dila = DA[j]; c.Kb = K[j]; c.dilb = DB[j]; c.res_a = 1; c.res_b_x = 0; c.alpha = 0.1f; c.N = 1; c.C = C; c.L = L;
    flops += 2.0 * 2.0 * C * C * K[j] * (double)L;
  }
  (void)hipStreamSynchronize(s);
  if (launch_rb_pair_multi(ctx, s, a, 3)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  (void)hipStreamSynchronize(s);
  if (check) {
    for (int j = 0; j < 3; j++) {
      std::vector<float> x1(hx.size()), y(hx.size()), got(hx.size());
      cpu_conv(hx, hw[j][0], hb[j][0], C, L, K[j], DA[j], 0.1f, hx.data(), x1);
      cpu_conv(x1, hw[j][1], hb[j][1], C, L, K[j], DB[j], 0.1f, x1.data(), y);
      (void)hipMemcpy(got.data(), yd[j], got.size() * 4, hipMemcpyDeviceToHost);
      double err = 0;
      size_t at = 0;
      for (size_t i = 0; i < y.size(); i++) if (std::fabs(got[i] - y[i]) > err) { err = std::fabs(got[i] - y[i]); at = i; }
      printf("  pair %d (K=%d da=%d db=%d): max|err| = %.3e at row %zu col %zu\n", j, K[j], DA[j], DB[j], err, at / L, at % L);
    }
  }
#ifdef PH_PAIR_TRACE
  {
    const size_t nst = (size_t)512 * 8 * 4 * 8;
    unsigned long long* tb;
    (void)hipMalloc(&tb, nst * 8);
    (void)hipMemset(tb, 0, nst * 8);
    rb_pair_set_trace(tb);
    launch_rb_pair_multi(ctx, s, a, 3);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(nst);
    (void)hipMemcpy(h.data(), tb, nst * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"top", "ring-a wait", "conv a", "ring b + res + barrier", "epilogue a + barrier", "conv b", "ring a' + epilogue b", "barrier + commit + barrier"};
    const int waves = C / 8;  // per block
    for (int round = 0; round < 4; round++) {
      double sum[8] = {0}, tile = 0;
      int cnt = 0;
      unsigned long long kmin = ~0ull, kmax = 0;
      for (int b = 0; b < 512; b++)
        for (int w = 0; w < waves; w++) {
          const unsigned long long* t = &h[(((size_t)b * waves + w) * 4 + round) * 8];
          if (!t[0] || !t[6]) continue;
          for (int k = 1; k < 7; k++) sum[k] += (double)(t[k] - t[k - 1]);
          if (t[7]) sum[7] += (double)(t[7] - t[6]);
          tile += (double)(t[6] - t[0]);
          kmin = std::min(kmin, t[0]); kmax = std::max(kmax, t[6]);
          cnt++;
        }
      if (!cnt) continue;
      printf("  round %d (%d waves): tile %.0f ticks; span of the round %.0f ticks (s_memtime: shader cycles)\n", round, cnt, tile / cnt, (double)(kmax - kmin));
      for (int k = 1; k < 8; k++) printf("    %-28s %8.1f ticks\n", names[k], sum[k] / cnt);
    }
  }
#endif
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; i++) launch_rb_pair_multi(ctx, s, a, 3);
  (void)hipStreamSynchronize(s);
  const int reps = 50;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; i++) launch_rb_pair_multi(ctx, s, a, 3);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / reps;
#ifndef PH_PAIR_VARIANT
#define PH_PAIR_VARIANT 0
#endif
  printf("pair v%d C=%d L=%d K=%d,%d,%d da=%d,%d,%d db=%d,%d,%d: %.2f us  %.1f TFLOP/s  (%s)\n", PH_PAIR_VARIANT, C, L, K[0], K[1], K[2], DA[0], DA[1], DA[2],
         DB[0], DB[1], DB[2], us, flops / us * 1e-6, hipGetErrorString(hipGetLastError()));
  return 0;
}
