// pipeprobe — timing ablations of conv_pipe_kernel (tools/probe; not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPH_PIPE_ABL=<bits> -x hip tools/probe/pipeprobe.cpp
//        piper-swift_amd/csrc/conv_pipe.hip piper-swift_amd/csrc/context.cpp -o pipeprobe_<bits>
// usage: pipeprobe C L K0,K1,K2 d0,d1,d2 [convt_stride]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef PH_PIPE_ABL
#define PH_PIPE_ABL 0
#endif

#include "../../piper-swift_amd/csrc/conv_win.h"

using namespace ph;

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 21504;
  int K[3] = {3, 5, 7}, D[3] = {1, 2, 3};
  if (argc > 3) sscanf(argv[3], "%d,%d,%d", &K[0], &K[1], &K[2]);
  if (argc > 4) sscanf(argv[4], "%d,%d,%d", &D[0], &D[1], &D[2]);
  const int res = argc > 5 ? atoi(argv[5]) : 1;
  const bool win = argc > 6 && argv[6][0] == 'w';  // "win": time conv_win_kernel on the same problem instead
  piper_hip_ctx* ctx = nullptr;
  if (piper_hip_create(0, &ctx)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  hipStream_t s;
  hipStreamCreate(&s);
  ConvWinArgs a[3];
  double flops = 0;
  for (int j = 0; j < 3; j++) {
    float *x, *y, *w, *wp, *b;
    hipMalloc(&x, (size_t)C * L * 4); hipMalloc(&y, (size_t)C * L * 4);
    hipMalloc(&w, (size_t)C * C * K[j] * 4); hipMalloc(&b, C * 4);
    hipMalloc(&wp, (win ? packed_conv_win_floats(C, C, K[j]) : packed_conv_pipe_floats(C, C, K[j])) * 4);
    std::vector<float> h((size_t)C * L);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)C * C * K[j]);
    for (size_t i = 0; i < hw.size(); i++) hw[i] = ((float)((i * 40503u >> 4) & 0xfff) / 4096.0f - 0.5f) * 0.1f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, C * 4);
    if (win) pack_conv_weights_win(s, w, C, C, K[j], wp);
    else pack_conv_weights_pipe(s, w, C, C, K[j], wp);
    ConvWinArgs& c = a[j];
    c.x = x; c.w4 = wp; c.bias = b; c.res = res ? x : nullptr; c.y = y; c.pro_alpha = 0.1f;
    c.N = 1; c.Cin = C; c.Cout = C; c.K = K[j]; c.dil = D[j]; c.padL = (K[j] * D[j] - D[j]) / 2; c.Lin = L; c.Lout = L; c.y_len = L;
    flops += 2.0 * C * C * K[j] * (double)L;
  }
  hipStreamSynchronize(s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&]() { return win ? launch_conv_win_multi(ctx, s, a, 3) : launch_conv_pipe_multi(ctx, s, a, 3); };
  for (int i = 0; i < 5; i++) go();
  hipStreamSynchronize(s);
  const int reps = 50;
  hipEventRecord(e0, s);
  for (int i = 0; i < reps; i++) go();
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / reps;
  printf("%s ABL=%d C=%d L=%d K=%d,%d,%d d=%d,%d,%d res=%d: %.2f us  %.1f TFLOP/s  (%s)\n", win ? "win " : "pipe", PH_PIPE_ABL, C, L, K[0], K[1], K[2], D[0], D[1], D[2], res, us,
         flops / us * 1e-6, hipGetErrorString(hipGetLastError()));
  return 0;
}
