// attprobe.hip — phase stamps (100 MHz realtime counter) of rel_attention_kernel<4>, block (0,0).
#define PH_ATT_STAMPS 1
#include "../../piper-swift_amd/csrc/attention.hip"
#include <vector>
namespace ph { void set_error(const char*, ...) {} const char* get_error() { return ""; }
int ensure_out(piper_hip_ctx*, float**, size_t, int) { return 0; } void release_deferred(piper_hip_ctx*) {} }
int main() {
  const int H = 2, d = 96, w = 4, R = 4;
  for (int T : {14, 112}) {
    const int G = 256 / d, TK = T < 128 ? T : 128;
    size_t n = (size_t)H * d * T;
    float *q, *k, *v, *ek, *ev, *o; unsigned long long* st;
    hipMalloc(&q, n * 4); hipMalloc(&k, n * 4); hipMalloc(&v, n * 4); hipMalloc(&o, n * 4);
    hipMalloc(&ek, 9 * d * 4); hipMalloc(&ev, 9 * d * 4); hipMalloc(&st, 64);
    hipMemset(q, 0, n * 4); hipMemset(k, 0, n * 4); hipMemset(v, 0, n * 4); hipMemset(ek, 0, 9 * d * 4); hipMemset(ev, 0, 9 * d * 4);
    const size_t lds = (size_t)(R * d + R * 9 + G * R * d + (size_t)d * (TK + 1) + (size_t)R * T + 2 * 9 * d + 256 * R) * 4;
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(rel_attention_kernel<4>, dim3((T + R - 1) / R, H, 1), dim3(256), lds, 0, q, k, v, ek, ev, o, H, d, T, w, (int64_t)n, (int64_t)n, G, TK, st);
      hipDeviceSynchronize();
    }
    unsigned long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
    const char* names[7] = {"smem setup+emb tables", "q strip", "qe (rel-K logits)", "K tiles + scores", "softmax", "V tiles + PV", "combine + store"};
    printf("T=%d (block 0,0; 10 ns ticks):\n", T);
    for (int i = 0; i < 7; i++) printf("  %-24s %6.2f us\n", names[i], (double)(h[i + 1] - h[i]) * 0.01);
    printf("  total %.2f us\n", (double)(h[7] - h[0]) * 0.01);
  }
  return 0;
}
