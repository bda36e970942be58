// ddsprobe — phase stamps (s_memtime) of dds_layer_kernel<3,true,12> (tools/probe; not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -w -DPH_DDS_TRACE -x hip tools/probe/ddsprobe.cpp piper-swift_amd/csrc/dp.hip
//        piper-swift_amd/csrc/context.cpp -o ddsprobe        usage: ddsprobe [T]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../piper-swift_amd/csrc/common.h"
namespace ph {
int launch_dds_layer(piper_hip_ctx* ctx, hipStream_t s, const float* x, const float* dw_w, const float* dw_b, const float* g1, const float* b1,
                     const float* pw16, const float* pw_b, const float* g2, const float* b2, float* out, int N, int H, int T, int K, int dil,
                     int pw_steps, const int* len_ptr, float eps);
void ph_dds_set_trace(unsigned long long* buf);
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 112, H = 192, K = 3;
  piper_hip_ctx* ctx = nullptr;
  if (piper_hip_create(0, &ctx)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  float *x, *y, *dw, *dwb, *g, *b, *pw, *pwb;
  hipMalloc(&x, (size_t)H * T * 4); hipMalloc(&y, (size_t)H * T * 4); hipMalloc(&dw, H * K * 4); hipMalloc(&dwb, H * 4); hipMalloc(&g, H * 4); hipMalloc(&b, H * 4);
  hipMalloc(&pw, (size_t)H * H * 4); hipMalloc(&pwb, H * 4);
  std::vector<float> h((size_t)H * std::max(T, H));
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
  hipMemcpy(x, h.data(), (size_t)H * T * 4, hipMemcpyHostToDevice); hipMemcpy(pw, h.data(), (size_t)H * H * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, h.data(), H * K * 4, hipMemcpyHostToDevice); hipMemcpy(dwb, h.data(), H * 4, hipMemcpyHostToDevice);
  std::vector<float> one(H, 1.0f);
  hipMemcpy(g, one.data(), H * 4, hipMemcpyHostToDevice); hipMemset(b, 0, H * 4); hipMemset(pwb, 0, H * 4);
  auto go = [&]() { return ph::launch_dds_layer(ctx, 0, x, dw, dwb, g, b, pw, pwb, g, b, y, 1, H, T, K, 1, H / 4, nullptr, 1e-5f); };
  for (int i = 0; i < 3; i++) if (go()) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  hipDeviceSynchronize();
  const int blocks = (T + 15) / 16, waves = 12;
  const size_t nst = (size_t)blocks * waves * 8;
  unsigned long long* tb;
  hipMalloc(&tb, nst * 8);
  hipMemset(tb, 0, nst * 8);
  ph::ph_dds_set_trace(tb);
  go();
  hipDeviceSynchronize();
  ph::ph_dds_set_trace(nullptr);
  std::vector<unsigned long long> t(nst);
  hipMemcpy(t.data(), tb, nst * 8, hipMemcpyDeviceToHost);
  const char* names[7] = {"", "loads + depthwise conv", "LayerNorm 1 + GELU -> LDS", "barrier", "pointwise conv (48 MFMAs)", "LayerNorm 2", "GELU + residual + store"};
  double sum[7] = {0};
  int cnt = 0;
  for (size_t wv = 0; wv < (size_t)blocks * waves; wv++) {
    const unsigned long long* s = &t[wv * 8];
    if (!s[0] || !s[6]) continue;
    for (int p = 1; p <= 6; p++) sum[p] += (double)(s[p] - s[p - 1]);
    cnt++;
  }
  printf("dds_layer_kernel<3,true,12> H=%d T=%d: %d waves traced\n", H, T, cnt);
  double tot = 0;
  for (int p = 1; p <= 6; p++) { printf("   %-32s %8.1f ticks\n", names[p], sum[p] / std::max(cnt, 1)); tot += sum[p] / std::max(cnt, 1); }
  printf("   total %.1f ticks per wave\n", tot);
  return 0;
}
