// streamprobe — phase trace (s_memtime) of conv_stream_kernel on the short-row convs of the encoder / flow (tools/probe).
// build (slow: all tap instantiations): see tools/probe/build_streamprobe.sh
// usage: streamprobe Cout Cin K L gate relu
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../piper-swift_amd/csrc/conv.h"

using namespace ph;

int main(int argc, char** argv) {
  const int Cout = argc > 1 ? atoi(argv[1]) : 384, Cin = argc > 2 ? atoi(argv[2]) : 192, K = argc > 3 ? atoi(argv[3]) : 5, L = argc > 4 ? atoi(argv[4]) : 336;
  const int gate = argc > 5 ? atoi(argv[5]) : 1, relu = argc > 6 ? atoi(argv[6]) : 0;
  piper_hip_ctx* ctx = nullptr;
  if (piper_hip_create(0, &ctx)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  hipStream_t s;
  (void)hipStreamCreate(&s);
  const int rows_out = gate ? Cout / 2 : Cout;
  float *x, *y, *w, *b, *p32, *p16;
  (void)hipMalloc(&x, (size_t)Cin * L * 4); (void)hipMalloc(&y, (size_t)rows_out * L * 4);
  (void)hipMalloc(&w, (size_t)Cout * Cin * K * 4); (void)hipMalloc(&b, Cout * 4);
  (void)hipMalloc(&p32, packed_conv_floats(Cout, Cin, K) * 4); (void)hipMalloc(&p16, packed_conv_floats(Cout, Cin, K, 16) * 4);
  std::vector<float> h((size_t)Cin * L, 0.25f), hw((size_t)Cout * Cin * K, 0.01f);
  (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(b, 0, Cout * 4);
  pack_conv_weights(s, w, Cout, Cin, K, p32);
  pack_conv_weights(s, w, Cout, Cin, K, p16, 16);
  float* p16g = nullptr;
  if (gate) { (void)hipMalloc(&p16g, packed_conv_floats(Cout, Cin, K, 16) * 4); pack_conv_weights_gate16(s, w, Cout, Cin, K, p16g); }
  ConvArgs a;
  a.x = x; a.y = y; a.w = p32; a.w16 = p16; a.w16g = p16g; a.bias = b; a.N = 1; a.Cin = Cin; a.Cout = Cout; a.K = K; a.dil = 1; a.padL = (K - 1) / 2; a.Lin = L; a.Lout = L;
  a.x_batch_stride = (int64_t)Cin * L; a.y_batch_stride = (int64_t)rows_out * L; a.y_len = L; a.gate = gate;
  if (relu) a.epilogue = EPI_RELU;
  (void)hipStreamSynchronize(s);
  for (int i = 0; i < 3; i++) if (launch_conv_mfma(ctx, s, a)) { fprintf(stderr, "%s\n", piper_hip_last_error()); return 1; }
  (void)hipStreamSynchronize(s);
  const size_t nst = (size_t)512 * 16 * 8;
  unsigned long long* tb;
  (void)hipMalloc(&tb, nst * 8);
  (void)hipMemset(tb, 0, nst * 8);
  ConvArgs at = a;
  at.trace = tb;
  launch_conv_mfma(ctx, s, at);
  (void)hipStreamSynchronize(s);
  std::vector<unsigned long long> t(nst);
  (void)hipMemcpy(t.data(), tb, nst * 8, hipMemcpyDeviceToHost);
  const char* names[5] = {"", "args + bias seed", "operand ring + MFMA loop", "split-K reduce (LDS, barrier)", "epilogue"};
  double sum[5] = {0};
  int cnt = 0, cnt4 = 0;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t wv = 0; wv < (size_t)512 * 16; wv++) {
    const unsigned long long* q = &t[wv * 8];
    if (!q[0] || !q[3]) continue;
    for (int k = 1; k <= 3; k++) sum[k] += (double)(q[k] - q[k - 1]);
    if (q[4]) { sum[4] += (double)(q[4] - q[3]); cnt4++; }
    t0 = std::min(t0, q[0]); t1 = std::max(t1, q[4] ? q[4] : q[3]);
    cnt++;
  }
  printf("conv Cout=%d Cin=%d K=%d L=%d gate=%d: %d waves traced (%d with an epilogue), first start → last end %.0f cycles\n", Cout, Cin, K, L, gate, cnt, cnt4,
         (double)(t1 - t0));
  for (int k = 1; k < 4; k++) printf("    %-32s %9.1f cycles\n", names[k], sum[k] / std::max(cnt, 1));
  printf("    %-32s %9.1f cycles\n", names[4], sum[4] / std::max(cnt4, 1));
  {
    double a5 = 0, a6 = 0, a2 = 0;
    int c = 0;
    for (size_t wv = 0; wv < (size_t)512 * 16; wv++) {
      const unsigned long long* q = &t[wv * 8];
      if (!q[1] || !q[5] || !q[6] || !q[2]) continue;
      a5 += (double)(q[5] - q[1]); a6 += (double)(q[6] - q[5]); a2 += (double)(q[2] - q[6]);
      c++;
    }
    if (c) printf("      of the loop phase: setup + prologue issue %.0f, until the first group is multiplied %.0f, rest %.0f cycles\n", a5 / c, a6 / c, a2 / c);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 100;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; i++) launch_conv_mfma(ctx, s, a);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("  %.2f us per launch (back to back in a stream)\n", ms * 1000.0 / reps);
  return 0;
}
