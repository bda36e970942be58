/*
 * piper_oracle.c — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's algorithm for the
 * Piper VITS hot path.  Nothing in the product (piper-swift_amd/, bench.py's GPU leg)
 * may link, import or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg use it, as the checker / reported CPU baseline.
 *
 * PARITY PINNING: the reference's own tests hold no numeric vectors for this path (SURVEY.md F4:
 * Tests/PiperMetalTests/SmokeTests.swift:18-19 asserts only non-empty + finite) and its Swift/Metal
 * sources cannot be compiled here ("unbuildable": needs swiftc + Metal + Darwin).  This oracle is
 * therefore pinned against independently generated PyTorch-CPU fp32 vectors committed under
 * tests/golden/ (generator: tools/gen_golden.py), i.e. by the ONNX operator definitions the
 * reference implements — not by reference-held fixtures: "parity unpinned" w.r.t. the reference's
 * own fixtures.
 *
 * Each function follows the cited reference code: loop order, accumulation order (bias first,
 * ci-major then k), bounds tests, float vs double intermediates.  File:line are relative to the
 * reference repo root (Sources/PiperMetal/...).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/piper_hip_voice_layout.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * Ops
 * ---------------------------------------------------------------------------------------------- */

/* Execution/CPUBackend.swift:20-73 (CPUBackend.conv1d); L_out formula :44; twin kernel
 * Kernels/conv1d.metal:28-71.  Returns L_out, or -1 on a contract violation. */
ORC_API long orc_conv1d(const float* x, long N, long Cin, long Lin, const float* w, long Cout, long K, const float* b,
                        long stride, long dil, long padL, long padR, long groups, float* y) {
  long g = groups < 1 ? 1 : groups;
  if (Cin % g || Cout % g) return -1;
  const long cig = Cin / g, cog = Cout / g;
  const long Lout = (Lin + padL + padR - dil * (K - 1) - 1) / stride + 1;
  if (Lout < 0) return -1;
  for (long n = 0; n < N; n++) {
    const long inBatchBase = n * Cin * Lin;
#pragma omp parallel for schedule(static)
    for (long co = 0; co < Cout; co++) {
      const long ciBase = (co / cog) * cig;
      const long wBase = co * cig * K;
      const float bias = b ? b[co] : 0.0f;
      for (long xo = 0; xo < Lout; xo++) {
        float acc = bias;
        const long inX0 = xo * stride - padL;
        for (long ci = 0; ci < cig; ci++) {
          const long inChanBase = inBatchBase + (ciBase + ci) * Lin;
          const long wChanBase = wBase + ci * K;
          for (long k = 0; k < K; k++) {
            const long inX = inX0 + k * dil;
            if (inX >= 0 && inX < Lin) acc += x[inChanBase + inX] * w[wChanBase + k];
          }
        }
        y[(n * Cout + co) * Lout + xo] = acc;
      }
    }
  }
  return Lout;
}

/* Kernels/conv1d.metal:97-142 (convtranspose1d_f32 — the only spec; no CPU version exists);
 * L_out Execution/MetalBackend.swift:2844.  w layout [Cin, Cout/g, K].  Returns L_out or -1. */
ORC_API long orc_convtranspose1d(const float* x, long N, long Cin, long Lin, const float* w, long CoutPerGroup, long K,
                                 const float* b, long stride, long dil, long padL, long padR, long outPad, long groups,
                                 float* y) {
  long g = groups < 1 ? 1 : groups;
  if (Cin % g) return -1;
  const long cig = Cin / g, cog = CoutPerGroup, Cout = cog * g;
  const long Lout = (Lin - 1) * stride - padL - padR + dil * (K - 1) + outPad + 1;
  if (Lout <= 0) return -1;
  for (long n = 0; n < N; n++) {
    const long inBatchBase = n * Cin * Lin;
#pragma omp parallel for schedule(static)
    for (long co = 0; co < Cout; co++) {
      const long gi = co / cog, coInGroup = co - gi * cog, ciBase = gi * cig;
      for (long xo = 0; xo < Lout; xo++) {
        float acc = b ? b[co] : 0.0f;
        for (long ci = 0; ci < cig; ci++) {
          const long inChan = ciBase + ci;
          const long inChanBase = inBatchBase + inChan * Lin;
          const long wBase = (inChan * cog + coInGroup) * K;
          for (long k = 0; k < K; k++) {
            const long t = xo + padL - k * dil;
            if (t % stride != 0) continue; /* C remainder, as MSL int % */
            const long inX = t / stride;
            if (inX >= 0 && inX < Lin) acc += x[inChanBase + inX] * w[wBase + k];
          }
        }
        y[(n * Cout + co) * Lout + xo] = acc;
      }
    }
  }
  return Lout;
}

/* Kernels/matmul.metal:22-49 (matmul_f32): C[b,m,n] = Σ_k A[b,m,k]·B[b,k,n], acc starts at 0, k ascending. */
ORC_API void orc_matmul(const float* A, const float* B, float* C, long batch, long M, long N, long K) {
  for (long bi = 0; bi < batch; bi++) {
    const float* a = A + bi * M * K;
    const float* bb = B + bi * K * N;
    float* c = C + bi * M * N;
#pragma omp parallel for schedule(static)
    for (long row = 0; row < M; row++)
      for (long col = 0; col < N; col++) {
        float acc = 0.0f;
        for (long k = 0; k < K; k++) acc += a[row * K + k] * bb[k * N + col];
        c[row * N + col] = acc;
      }
  }
}

/* Kernels/softmax.metal:13-41 (softmax_lastdim_f32): max, exp+sum (float), multiply by 1/s. */
ORC_API void orc_softmax_lastdim(const float* x, float* y, long rows, long cols) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; r++) {
    const long base = r * cols;
    float m = x[base];
    for (long i = 1; i < cols; i++) m = fmaxf(m, x[base + i]);
    float s = 0.0f;
    for (long i = 0; i < cols; i++) {
      const float e = expf(x[base + i] - m);
      y[base + i] = e;
      s += e;
    }
    const float inv = 1.0f / s;
    for (long i = 0; i < cols; i++) y[base + i] *= inv;
  }
}

/* Execution/CPUBackend.swift:75-110, 272-294 (relu, leakyRelu, softplus, neg, exp, ceil, sqrt, tanh, sigmoid:
 * exp/tanh evaluated in Double then rounded); erf: Kernels/elementwise.metal:292-312 is A&S 7.1.26
 * (|err| ≤ 1.5e-7) — restated with true erf, inside the stated tolerance.
 * op codes = piper_hip_unary_op. */
ORC_API int orc_unary(int op, float alpha, const float* x, float* y, long n) {
  for (long i = 0; i < n; i++) {
    const float v = x[i];
    float r;
    switch (op) {
      case 0: r = v > 0 ? v : 0; break;
      case 1: r = v >= 0 ? v : alpha * v; break;
      case 2: r = (float)tanh((double)v); break;
      case 3:
        if (v >= 0) {
          const float z = (float)exp((double)-v);
          r = 1 / (1 + z);
        } else {
          const float z = (float)exp((double)v);
          r = z / (1 + z);
        }
        break;
      case 4: r = (float)exp((double)v); break;
      case 5: r = -v; break;
      case 6: r = sqrtf(v); break;
      case 7: r = v > 0 ? v + (float)log(1.0 + exp((double)-v)) : (float)log(1.0 + exp((double)v)); break;
      case 8: r = (float)ceil((double)v); break;
      case 9: r = (float)erf((double)v); break;
      default: return -1;
    }
    y[i] = r;
  }
  return 0;
}

/* Kernels/elementwise.metal:132-163 (xorshift32, random_normal_like_f32); called with the fixed seed 1234 for both
 * RandomNormalLike nodes (Execution/GraphExecutor.swift:2656-2659; MetalBackend.swift:3404 passes the low 32 bits).
 * draws[2i], draws[2i+1] (optional) receive the raw u0 / u1 integers of element i. uint arithmetic wraps mod 2^32 as in MSL. */
static uint32_t orc_xorshift32(uint32_t x) {
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return x;
}
ORC_API void orc_random_normal_like(uint64_t seed, long n, float* out, uint32_t* draws) {
  const uint32_t seed_lo = (uint32_t)(seed & 0xffffffffu);
  for (long gid = 0; gid < n; gid++) {
    uint32_t state = seed_lo ^ ((uint32_t)gid * 747796405u + 2891336453u);
    state = orc_xorshift32(state);
    const uint32_t u0i = state;
    state = orc_xorshift32(state);
    const uint32_t u1i = state;
    const float u0 = ((float)u0i + 1.0f) / 4294967296.0f; /* (0, 1] */
    const float u1 = ((float)u1i + 1.0f) / 4294967296.0f;
    const float r = sqrtf(-2.0f * logf(u0));
    const float theta = 6.28318530718f * u1;
    if (out) out[gid] = r * cosf(theta);
    if (draws) { draws[2 * gid] = u0i; draws[2 * gid + 1] = u1i; }
  }
}

static void orc_strides(const long* shape, int rank, long* s) {
  if (rank == 0) return;
  s[rank - 1] = 1;
  for (int i = rank - 2; i >= 0; i--) s[i] = s[i + 1] * shape[i + 1];
}

/* Broadcast rule Execution/CPUBackend.swift:1766-1818 (broadcastShape + expand) and the op kernels
 * Kernels/elementwise.metal:52-130. op codes = piper_hip_binary_op. Returns out rank or -1. */
ORC_API int orc_binary_broadcast(int op, const float* a, const long* ashape, int ra, const float* b, const long* bshape,
                                 int rb, float* out, long* oshape) {
  const int r = ra > rb ? ra : rb;
  if (r > 4) return -1;
  long pa[4], pb[4], sa[4], sb[4], so[4];
  for (int i = 0; i < r; i++) {
    pa[i] = i < r - ra ? 1 : ashape[i - (r - ra)];
    pb[i] = i < r - rb ? 1 : bshape[i - (r - rb)];
    if (pa[i] == pb[i]) oshape[i] = pa[i];
    else if (pa[i] == 1) oshape[i] = pb[i];
    else if (pb[i] == 1) oshape[i] = pa[i];
    else return -1;
  }
  orc_strides(pa, r, sa);
  orc_strides(pb, r, sb);
  orc_strides(oshape, r, so);
  long total = 1;
  for (int i = 0; i < r; i++) total *= oshape[i];
  for (long f = 0; f < total; f++) {
    long rem = f, ia = 0, ib = 0;
    for (int d = 0; d < r; d++) {
      const long idx = rem / so[d];
      rem %= so[d];
      ia += (pa[d] == 1 ? 0 : idx) * sa[d];
      ib += (pb[d] == 1 ? 0 : idx) * sb[d];
    }
    const float x = a[ia], y = b[ib];
    float v;
    switch (op) {
      case 0: v = x + y; break;
      case 1: v = x - y; break;
      case 2: v = x * y; break;
      case 3: v = x / y; break;
      case 4: v = powf(x, y); break;
      default: return -1;
    }
    out[f] = v;
  }
  return r;
}

/* Execution/CPUBackend.swift:1039-1080 (padConstant): pads = [begin..., end...]. */
ORC_API int orc_pad_constant(const float* x, const long* shape, int rank, const long* pads, float value, float* out,
                             long* oshape) {
  if (rank > 4 || rank < 1) return -1;
  long si[4], so[4];
  long inCount = 1, outCount = 1;
  for (int d = 0; d < rank; d++) {
    oshape[d] = shape[d] + pads[d] + pads[rank + d];
    inCount *= shape[d];
    outCount *= oshape[d];
  }
  orc_strides(shape, rank, si);
  orc_strides(oshape, rank, so);
  for (long i = 0; i < outCount; i++) out[i] = value;
  for (long f = 0; f < inCount; f++) {
    long rem = f, of = 0;
    for (int d = 0; d < rank; d++) {
      const long idx = rem / si[d];
      rem %= si[d];
      of += (idx + pads[d]) * so[d];
    }
    out[of] = x[f];
  }
  return 0;
}

/* Slice along one axis (Execution/CPUBackend.swift:1500-1730 generic slice; step may be negative — VITS Flip is
 * axis 1, start −1, end INT_MIN, step −1, resolved by the caller to start=C−1, end=−1). Elements:
 * start, start+step, … while (step>0 ? i<end : i>end). */
ORC_API int orc_slice(const float* x, const long* shape, int rank, int axis, long start, long end, long step, float* out,
                      long* oshape) {
  if (rank > 4 || rank < 1 || step == 0 || axis < 0 || axis >= rank) return -1;
  long cnt = 0;
  if (step > 0) { if (end > start) cnt = (end - start + step - 1) / step; }
  else { if (start > end) cnt = (start - end + (-step) - 1) / (-step); }
  long outer = 1, inner = 1;
  for (int d = 0; d < axis; d++) outer *= shape[d];
  for (int d = axis + 1; d < rank; d++) inner *= shape[d];
  for (int d = 0; d < rank; d++) oshape[d] = shape[d];
  oshape[axis] = cnt;
  for (long o = 0; o < outer; o++)
    for (long j = 0; j < cnt; j++)
      memcpy(out + (o * cnt + j) * inner, x + (o * shape[axis] + start + j * step) * inner, (size_t)inner * sizeof(float));
  return 0;
}

/* Execution/CPUBackend.swift:818-875 (transpose): out dim d = in dim perm[d]. */
ORC_API int orc_transpose(const float* x, const long* shape, int rank, const int* perm, float* out, long* oshape) {
  if (rank > 4 || rank < 1) return -1;
  long si[4], so[4];
  long total = 1;
  for (int d = 0; d < rank; d++) {
    oshape[d] = shape[perm[d]];
    total *= shape[d];
  }
  orc_strides(shape, rank, si);
  orc_strides(oshape, rank, so);
  for (long f = 0; f < total; f++) {
    long rem = f, inf = 0;
    for (int d = 0; d < rank; d++) {
      const long idx = rem / so[d];
      rem %= so[d];
      inf += idx * si[perm[d]];
    }
    out[f] = x[inf];
  }
  return 0;
}

/* Stride-0 expand, Execution/CPUBackend.swift:1783-1818 (expand) / Kernels/expand.metal. Equal ranks. */
ORC_API int orc_expand(const float* x, const long* ishape, const long* oshape, int rank, float* out) {
  if (rank > 4 || rank < 1) return -1;
  long si[4], so[4], total = 1;
  orc_strides(ishape, rank, si);
  orc_strides(oshape, rank, so);
  for (int d = 0; d < rank; d++) {
    if (ishape[d] != oshape[d] && ishape[d] != 1) return -1;
    total *= oshape[d];
  }
  for (long f = 0; f < total; f++) {
    long rem = f, inf = 0;
    for (int d = 0; d < rank; d++) {
      const long idx = rem / so[d];
      rem %= so[d];
      inf += (ishape[d] == 1 ? 0 : idx) * si[d];
    }
    out[f] = x[inf];
  }
  return 0;
}

/* Execution/CPUBackend.swift:296-330 (reduceMean, last axis): float running sum, then / Float(cols). */
ORC_API void orc_reduce_mean_lastdim(const float* x, float* y, long rows, long cols) {
  for (long r = 0; r < rows; r++) {
    float s = 0;
    for (long c = 0; c < cols; c++) s += x[r * cols + c];
    y[r] = s / (float)cols;
  }
}

/* concat2_axis1 / split2_axis1 on NCL (Kernels/tensorops.metal; GraphExecutor.swift:1107-1124, 2254-2262). */
ORC_API void orc_concat2_axis1(const float* a, long N, long Ca, const float* b, long Cb, long L, float* out) {
  for (long n = 0; n < N; n++) {
    memcpy(out + n * (Ca + Cb) * L, a + n * Ca * L, (size_t)(Ca * L) * sizeof(float));
    memcpy(out + n * (Ca + Cb) * L + Ca * L, b + n * Cb * L, (size_t)(Cb * L) * sizeof(float));
  }
}
ORC_API void orc_split2_axis1(const float* x, long N, long C, long L, long c0, float* o0, float* o1) {
  for (long n = 0; n < N; n++) {
    memcpy(o0 + n * c0 * L, x + n * C * L, (size_t)(c0 * L) * sizeof(float));
    memcpy(o1 + n * (C - c0) * L, x + n * C * L + c0 * L, (size_t)((C - c0) * L) * sizeof(float));
  }
}

/* ------------------------------------------------------------------------------------------------
 * Module compositions: the op sequence the exported Piper VITS graph runs through the reference's
 * executor (one executeNode arm per op; Execution/GraphExecutor.swift:591-2664), N = 1, full-length
 * masks (all ones, so mask multiplies are identities and are omitted).
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
  float* d;
  int rank;
  long s[4];
} T;

static long t_count(const T* t) {
  long c = 1;
  for (int i = 0; i < t->rank; i++) c *= t->s[i];
  return c;
}
static T t_new(int rank, long a, long b, long c, long d) {
  T t;
  t.rank = rank;
  t.s[0] = a; t.s[1] = b; t.s[2] = c; t.s[3] = d;
  long n = t_count(&t);
  t.d = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
  return t;
}
static void t_free(T* t) { free(t->d); t->d = NULL; }
static T t_clone(const T* a) {
  T t = *a;
  long n = t_count(a);
  t.d = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
  memcpy(t.d, a->d, (size_t)n * sizeof(float));
  return t;
}
/* Reshape is metadata only (Execution/CPUBackend.swift:1006-1037; GraphExecutor.swift:1130-1178 aliases). */
static T t_view(T t, int rank, long a, long b, long c, long d) {
  t.rank = rank;
  t.s[0] = a; t.s[1] = b; t.s[2] = c; t.s[3] = d;
  return t;
}
static T t_conv(const T* x, const float* w, long Cout, long K, const float* b, long stride, long dil, long padL,
                long padR, long groups) {
  long Lout = (x->s[2] + padL + padR - dil * (K - 1) - 1) / stride + 1;
  T y = t_new(3, x->s[0], Cout, Lout, 1);
  orc_conv1d(x->d, x->s[0], x->s[1], x->s[2], w, Cout, K, b, stride, dil, padL, padR, groups, y.d);
  return y;
}
static T t_unary(const T* x, int op, float alpha) {
  T y = *x;
  y.d = (float*)malloc((size_t)t_count(x) * sizeof(float));
  orc_unary(op, alpha, x->d, y.d, t_count(x));
  return y;
}
static T t_binary(int op, const T* a, const T* b) {
  long os[4];
  int r = a->rank > b->rank ? a->rank : b->rank;
  long n = 1;
  for (int i = 0; i < r; i++) {
    long da = i < r - a->rank ? 1 : a->s[i - (r - a->rank)];
    long db = i < r - b->rank ? 1 : b->s[i - (r - b->rank)];
    os[i] = da > db ? da : db;
    n *= os[i];
  }
  T y = t_new(1, n, 1, 1, 1);
  y.rank = orc_binary_broadcast(op, a->d, a->s, a->rank, b->d, b->s, b->rank, y.d, y.s);
  return y;
}
static T t_scalar(float v) {
  T t = t_new(1, 1, 1, 1, 1);
  t.d[0] = v;
  return t;
}
static T t_pad(const T* x, const long* pads) {
  long os[4], n = 1;
  for (int d = 0; d < x->rank; d++) { os[d] = x->s[d] + pads[d] + pads[x->rank + d]; n *= os[d]; }
  T y = t_new(1, n, 1, 1, 1);
  y.rank = x->rank;
  orc_pad_constant(x->d, x->s, x->rank, pads, 0.0f, y.d, y.s);
  return y;
}
static T t_slice(const T* x, int axis, long start, long end, long step) {
  T y = t_new(1, t_count(x) > 0 ? t_count(x) : 1, 1, 1, 1);
  y.rank = x->rank;
  orc_slice(x->d, x->s, x->rank, axis, start, end, step, y.d, y.s);
  return y;
}
static T t_transpose(const T* x, const int* perm) {
  T y = t_new(1, t_count(x), 1, 1, 1);
  y.rank = x->rank;
  orc_transpose(x->d, x->s, x->rank, perm, y.d, y.s);
  return y;
}
/* MatMul arm with the rank-4 lead-dim broadcast materialised by expand (GraphExecutor.swift:1862-1915). */
static T t_matmul(const T* a, const T* b) {
  const int r = a->rank;
  T ae = *a, be = *b;
  int freeA = 0, freeB = 0;
  if (r == 4) {
    long lead[2];
    for (int i = 0; i < 2; i++) lead[i] = a->s[i] > b->s[i] ? a->s[i] : b->s[i];
    if (a->s[0] != lead[0] || a->s[1] != lead[1]) {
      ae = t_new(4, lead[0], lead[1], a->s[2], a->s[3]);
      orc_expand(a->d, a->s, ae.s, 4, ae.d);
      freeA = 1;
    }
    if (b->s[0] != lead[0] || b->s[1] != lead[1]) {
      be = t_new(4, lead[0], lead[1], b->s[2], b->s[3]);
      orc_expand(b->d, b->s, be.s, 4, be.d);
      freeB = 1;
    }
  }
  long batch = 1;
  for (int i = 0; i < r - 2; i++) batch *= ae.s[i];
  const long M = ae.s[r - 2], K = ae.s[r - 1], N = be.s[r - 1];
  T c = ae;
  c.s[r - 1] = N;
  c.d = (float*)malloc((size_t)(batch * M * N) * sizeof(float));
  orc_matmul(ae.d, be.d, c.d, batch, M, N, K);
  if (freeA) t_free(&ae);
  if (freeB) t_free(&be);
  return c;
}
static T t_softmax(const T* x) {
  T y = *x;
  y.d = (float*)malloc((size_t)t_count(x) * sizeof(float));
  const long cols = x->s[x->rank - 1];
  orc_softmax_lastdim(x->d, y.d, t_count(x) / cols, cols);
  return y;
}

/* Relative embeddings window: VITS `_get_relative_embeddings` as exported: Pad on axis 1 of [1,2w+1,d] then Slice.
 * (Pad arm GraphExecutor.swift:1180-1210; Slice arm :1371-1426.) Result [1, 2T−1, d]. */
static T rel_embeddings(const float* emb, long window, long d, long T_) {
  T e = t_new(3, 1, 2 * window + 1, d, 1);
  memcpy(e.d, emb, (size_t)((2 * window + 1) * d) * sizeof(float));
  long pad_length = T_ - (window + 1);
  if (pad_length < 0) pad_length = 0;
  long slice_start = (window + 1) - T_;
  if (slice_start < 0) slice_start = 0;
  T padded = e;
  if (pad_length > 0) {
    long pads[6] = {0, pad_length, 0, 0, pad_length, 0};
    padded = t_pad(&e, pads);
    t_free(&e);
  }
  T out = t_slice(&padded, 1, slice_start, slice_start + 2 * T_ - 1, 1);
  t_free(&padded);
  return out;
}

/* VITS `_relative_position_to_absolute_position` as Pad/Reshape/Pad/Reshape/Slice:
 * x [1,h,T,2T−1] → [1,h,T,T]. */
static T rel_to_abs(const T* x) {
  const long h = x->s[1], L = x->s[2];
  long p1[8] = {0, 0, 0, 0, 0, 0, 0, 1};
  T a = t_pad(x, p1); /* [1,h,L,2L] */
  T flat = t_view(a, 3, 1, h, L * 2 * L, 1);
  long p2[6] = {0, 0, 0, 0, 0, L - 1};
  T b = t_pad(&flat, p2); /* [1,h,2L²+L−1] */
  t_free(&a);
  T v = t_view(b, 4, 1, h, L + 1, 2 * L - 1);
  T s1 = t_slice(&v, 2, 0, L, 1);
  T s2 = t_slice(&s1, 3, L - 1, 2 * L - 1, 1);
  t_free(&b);
  t_free(&s1);
  return s2;
}
/* VITS `_absolute_position_to_relative_position`: x [1,h,T,T] → [1,h,T,2T−1]. */
static T abs_to_rel(const T* x) {
  const long h = x->s[1], L = x->s[2];
  long p1[8] = {0, 0, 0, 0, 0, 0, 0, L - 1};
  T a = t_pad(x, p1); /* [1,h,L,2L−1] */
  T flat = t_view(a, 3, 1, h, L * (2 * L - 1), 1);
  long p2[6] = {0, 0, L, 0, 0, 0};
  T b = t_pad(&flat, p2); /* [1,h,2L²] */
  t_free(&a);
  T v = t_view(b, 4, 1, h, L, 2 * L);
  T s = t_slice(&v, 3, 1, 2 * L, 1);
  t_free(&b);
  return s;
}

/* Attention core on already-projected q,k,v [1,H·d,T] (Piper attentions.MultiHeadAttention.attention as exported;
 * MatMul/Softmax arms GraphExecutor.swift:1862-1929). Returns [1,H·d,T]. */
ORC_API int orc_rel_attention(const float* q, const float* k, const float* v, const float* emb_k, const float* emb_v,
                              long heads, long d, long T_, long window, float* out) {
  T qt = t_new(4, 1, heads, d, T_), kt = t_new(4, 1, heads, d, T_), vt = t_new(4, 1, heads, d, T_);
  memcpy(qt.d, q, (size_t)(heads * d * T_) * sizeof(float));
  memcpy(kt.d, k, (size_t)(heads * d * T_) * sizeof(float));
  memcpy(vt.d, v, (size_t)(heads * d * T_) * sizeof(float));
  const int p0132[4] = {0, 1, 3, 2};
  T query = t_transpose(&qt, p0132); /* [1,h,T,d] */
  T value = t_transpose(&vt, p0132);
  /* key.view(b,h,d,T).transpose(2,3).transpose(-2,-1) == key viewed [1,h,d,T] */
  T sc = t_scalar(sqrtf((float)d));
  T qs = t_binary(3, &query, &sc); /* query / sqrt(d) (Div) */
  T scores = t_matmul(&qs, &kt);   /* [1,h,T,T] */
  T ek = rel_embeddings(emb_k, window, d, T_); /* [1,2T−1,d] */
  T ek4 = t_view(ek, 4, 1, 1, 2 * T_ - 1, d);
  T ekT = t_transpose(&ek4, p0132);    /* [1,1,d,2T−1] */
  T rel_logits = t_matmul(&qs, &ekT);  /* [1,h,T,2T−1] */
  T local = rel_to_abs(&rel_logits);
  T scores2 = t_binary(0, &scores, &local);
  T p = t_softmax(&scores2);
  T o = t_matmul(&p, &value); /* [1,h,T,d] */
  T relw = abs_to_rel(&p);    /* [1,h,T,2T−1] */
  T ev = rel_embeddings(emb_v, window, d, T_);
  T ev4 = t_view(ev, 4, 1, 1, 2 * T_ - 1, d);
  T o2 = t_matmul(&relw, &ev4);
  T osum = t_binary(0, &o, &o2);
  T ot = t_transpose(&osum, p0132); /* [1,h,d,T] → view [1,h·d,T] */
  memcpy(out, ot.d, (size_t)(heads * d * T_) * sizeof(float));
  T* all[] = {&qt, &kt, &vt, &query, &value, &sc, &qs, &scores, &ek, &ekT, &rel_logits, &local, &scores2, &p, &o, &relw,
              &ev, &o2, &osum, &ot};
  for (size_t i = 0; i < sizeof all / sizeof *all; i++) t_free(all[i]);
  return 0;
}

/* Channel LayerNorm as exported at opset 15: Transpose, ReduceMean, Sub, Pow(2), ReduceMean, Add(eps), Sqrt, Div,
 * Mul(gamma), Add(beta), Transpose (GraphExecutor.swift:2071-2125; reduce :2104-2125). x [1,C,T], y optional residual. */
ORC_API int orc_add_layernorm(const float* x, const float* y, const float* gamma, const float* beta, long C, long T_,
                              float eps, float* out) {
  T a = t_new(3, 1, C, T_, 1);
  for (long i = 0; i < C * T_; i++) a.d[i] = y ? x[i] + y[i] : x[i];
  const int p021[3] = {0, 2, 1};
  T xt = t_transpose(&a, p021); /* [1,T,C] */
  T mean = t_new(3, 1, T_, 1, 1);
  orc_reduce_mean_lastdim(xt.d, mean.d, T_, C);
  T xc = t_binary(1, &xt, &mean);
  T two = t_scalar(2.0f);
  T sq = t_binary(4, &xc, &two);
  T var = t_new(3, 1, T_, 1, 1);
  orc_reduce_mean_lastdim(sq.d, var.d, T_, C);
  T e = t_scalar(eps);
  T ve = t_binary(0, &var, &e);
  T sd = t_unary(&ve, 6, 0);
  T nrm = t_binary(3, &xc, &sd);
  T g = t_new(1, C, 1, 1, 1), bt = t_new(1, C, 1, 1, 1);
  memcpy(g.d, gamma, (size_t)C * sizeof(float));
  memcpy(bt.d, beta, (size_t)C * sizeof(float));
  T sg = t_binary(2, &nrm, &g);
  T sb = t_binary(0, &sg, &bt);
  T o = t_transpose(&sb, p021);
  memcpy(out, o.d, (size_t)(C * T_) * sizeof(float));
  T* all[] = {&a, &xt, &mean, &xc, &two, &sq, &var, &e, &ve, &sd, &nrm, &g, &bt, &sg, &sb, &o};
  for (size_t i = 0; i < sizeof all / sizeof *all; i++) t_free(all[i]);
  return 0;
}

/* One WaveNet layer (Piper modules.WN.forward body; Conv/Tanh/Sigmoid/Mul/Add arms GraphExecutor.swift:1739-1810,
 * 2017-2045, 741-779, 861-899; Slice on axis 1 :1322-1345). x [1,C,T]; skip_in may be NULL. */
ORC_API int orc_wavenet_layer(const float* x, const float* skip_in, const float* w_in, const float* b_in,
                              const float* w_rs, const float* b_rs, long C, long T_, long K, long dil, int last,
                              float* x_out, float* skip_out) {
  T xt = t_new(3, 1, C, T_, 1);
  memcpy(xt.d, x, (size_t)(C * T_) * sizeof(float));
  const long pad = (K * dil - dil) / 2;
  T x_in = t_conv(&xt, w_in, 2 * C, K, b_in, 1, dil, pad, pad, 1);
  T a = t_slice(&x_in, 1, 0, C, 1), b = t_slice(&x_in, 1, C, 2 * C, 1);
  T ta = t_unary(&a, 2, 0), sb = t_unary(&b, 3, 0);
  T acts = t_binary(2, &ta, &sb);
  const long Crs = last ? C : 2 * C;
  T rs = t_conv(&acts, w_rs, Crs, 1, b_rs, 1, 1, 0, 0, 1);
  if (!last) {
    for (long i = 0; i < C * T_; i++) x_out[i] = x[i] + rs.d[i];
    for (long i = 0; i < C * T_; i++) skip_out[i] = (skip_in ? skip_in[i] : 0.0f) + rs.d[C * T_ + i];
  } else {
    for (long i = 0; i < C * T_; i++) skip_out[i] = (skip_in ? skip_in[i] : 0.0f) + rs.d[i];
  }
  T* all[] = {&xt, &x_in, &a, &b, &ta, &sb, &acts, &rs};
  for (size_t i = 0; i < sizeof all / sizeof *all; i++) t_free(all[i]);
  return 0;
}

/* HiFi-GAN ResBlock1 / ResBlock2 (Piper hifigan; LeakyRelu arm GraphExecutor.swift:2047-2069, Conv :1739-1810,
 * Add :741-779). weights[i]/biases[i]: type 2 → n_dil convs; type 1 → 2·n_dil (c1_0,c2_0,c1_1,…). */
ORC_API int orc_hifigan_resblock(int type, const float* x, long C, long T_, long K, const int* dils, int n_dil,
                                 const float* const* weights, const float* const* biases, float slope, float* out) {
  T cur = t_new(3, 1, C, T_, 1);
  memcpy(cur.d, x, (size_t)(C * T_) * sizeof(float));
  for (int i = 0; i < n_dil; i++) {
    const long d = dils[i], pad = (K * d - d) / 2;
    T xt = t_unary(&cur, 1, slope);
    T y;
    if (type == 1) {
      T c1 = t_conv(&xt, weights[2 * i], C, K, biases[2 * i], 1, d, pad, pad, 1);
      T l2 = t_unary(&c1, 1, slope);
      const long pad1 = (K - 1) / 2;
      y = t_conv(&l2, weights[2 * i + 1], C, K, biases[2 * i + 1], 1, 1, pad1, pad1, 1);
      t_free(&c1);
      t_free(&l2);
    } else {
      y = t_conv(&xt, weights[i], C, K, biases[i], 1, d, pad, pad, 1);
    }
    T nx = t_binary(0, &y, &cur); /* xt + x */
    t_free(&xt);
    t_free(&y);
    t_free(&cur);
    cur = nx;
  }
  memcpy(out, cur.d, (size_t)(C * T_) * sizeof(float));
  t_free(&cur);
  return 0;
}

/* ---- whole utterance ---- */
typedef struct {
  const piper_hip_voice_config* cfg;
  const float* blob;
  piper_tensor_desc* descs;
  int n, cap;
} blob_index;
static void bi_visit(const piper_tensor_desc* d, void* user) {
  blob_index* bi = (blob_index*)user;
  if (bi->n == bi->cap) {
    bi->cap = bi->cap ? bi->cap * 2 : 256;
    bi->descs = (piper_tensor_desc*)realloc(bi->descs, (size_t)bi->cap * sizeof *bi->descs);
  }
  bi->descs[bi->n++] = *d;
}
static const float* bi_get(const blob_index* bi, const char* fmt, int a, int b, const char* suffix) {
  char nm[128], full[160];
  snprintf(nm, sizeof nm, fmt, a, b);
  snprintf(full, sizeof full, "%s%s", nm, suffix);
  for (int i = 0; i < bi->n; i++)
    if (!strcmp(bi->descs[i].name, full)) return bi->blob + bi->descs[i].offset;
  fprintf(stderr, "oracle: tensor %s not found\n", full);
  abort();
}

ORC_API size_t orc_voice_blob_floats(const piper_hip_voice_config* cfg) { return piper_hip_layout_walk(cfg, NULL, NULL); }

/* HiFi-GAN generator (Piper models.Generator.forward as exported). z [1,inter,F] → audio [F·Πrates]. */
static T orc_generator(const blob_index* bi, const T* z) {
  const piper_hip_voice_config* c = bi->cfg;
  T x = t_conv(z, bi_get(bi, "dec.conv_pre", 0, 0, ".weight"), c->up_initial, 7, bi_get(bi, "dec.conv_pre", 0, 0, ".bias"), 1,
               1, 3, 3, 1);
  long ch = c->up_initial;
  for (int u = 0; u < c->n_ups; u++) {
    T l = t_unary(&x, 1, 0.1f);
    t_free(&x);
    const long k = c->up_kernels[u], s = c->up_rates[u], pad = (k - s) / 2;
    const long Lout = (l.s[2] - 1) * s - 2 * pad + (k - 1) + 1;
    T up = t_new(3, 1, ch / 2, Lout, 1);
    orc_convtranspose1d(l.d, 1, ch, l.s[2], bi_get(bi, "dec.ups.%d", u, 0, ".weight"), ch / 2, k,
                        bi_get(bi, "dec.ups.%d", u, 0, ".bias"), s, 1, pad, pad, 0, 1, up.d);
    t_free(&l);
    ch /= 2;
    T xs;
    xs.d = NULL;
    for (int j = 0; j < c->n_rb; j++) {
      const int rb = u * c->n_rb + j;
      const float* ws[6];
      const float* bs[6];
      for (int d = 0; d < c->rb_n_dil; d++) {
        if (c->resblock_type == 1) {
          ws[2 * d] = bi_get(bi, "dec.resblocks.%d.convs1.%d", rb, d, ".weight");
          bs[2 * d] = bi_get(bi, "dec.resblocks.%d.convs1.%d", rb, d, ".bias");
          ws[2 * d + 1] = bi_get(bi, "dec.resblocks.%d.convs2.%d", rb, d, ".weight");
          bs[2 * d + 1] = bi_get(bi, "dec.resblocks.%d.convs2.%d", rb, d, ".bias");
        } else {
          ws[d] = bi_get(bi, "dec.resblocks.%d.convs.%d", rb, d, ".weight");
          bs[d] = bi_get(bi, "dec.resblocks.%d.convs.%d", rb, d, ".bias");
        }
      }
      T r = t_new(3, 1, ch, Lout, 1);
      orc_hifigan_resblock(c->resblock_type, up.d, ch, Lout, c->rb_kernels[j], c->rb_dilations[j], c->rb_n_dil, ws, bs,
                           0.1f, r.d);
      if (!xs.d) xs = r;
      else {
        T s2 = t_binary(0, &xs, &r);
        t_free(&xs);
        t_free(&r);
        xs = s2;
      }
    }
    t_free(&up);
    T nk = t_scalar((float)c->n_rb);
    x = t_binary(3, &xs, &nk); /* xs / num_kernels (Div) */
    t_free(&xs);
    t_free(&nk);
  }
  T l = t_unary(&x, 1, 0.01f);
  t_free(&x);
  T post = t_conv(&l, bi_get(bi, "dec.conv_post", 0, 0, ".weight"), 1, 7, NULL, 1, 1, 3, 3, 1);
  t_free(&l);
  T o = t_unary(&post, 2, 0);
  t_free(&post);
  return o;
}

ORC_API int orc_generator_forward(const piper_hip_voice_config* cfg, const float* blob, const float* z, long F,
                                  float* audio) {
  blob_index bi = {cfg, blob, NULL, 0, 0};
  piper_hip_layout_walk(cfg, bi_visit, &bi);
  T zt = t_new(3, 1, cfg->inter, F, 1);
  memcpy(zt.d, z, (size_t)(cfg->inter * F) * sizeof(float));
  T o = orc_generator(&bi, &zt);
  memcpy(audio, o.d, (size_t)t_count(&o) * sizeof(float));
  t_free(&o);
  t_free(&zt);
  free(bi.descs);
  return 0;
}

/* Flow, reverse direction (Piper models.ResidualCouplingBlock.forward(reverse=True): for each of the reversed
 * [coupling, Flip] pairs: Flip (Slice step −1, GraphExecutor.swift:1322-1345) then coupling.reverse
 * (Split :2254-2262, Sub :821-859, Concat :1107-1124)). z_p [1,inter,F] → z. */
static T orc_flow_reverse(const blob_index* bi, const T* zp) {
  const piper_hip_voice_config* c = bi->cfg;
  const long I = c->inter, half = I / 2, H = c->hidden, F = zp->s[2];
  T x = t_clone(zp);
  for (int f = c->n_flows - 1; f >= 0; f--) {
    T fl = t_slice(&x, 1, I - 1, -1, -1);
    t_free(&x);
    T x0 = t_new(3, 1, half, F, 1), x1 = t_new(3, 1, half, F, 1);
    orc_split2_axis1(fl.d, 1, I, F, half, x0.d, x1.d);
    t_free(&fl);
    T h = t_conv(&x0, bi_get(bi, "flow.flows.%d.pre", 2 * f, 0, ".weight"), H, 1,
                 bi_get(bi, "flow.flows.%d.pre", 2 * f, 0, ".bias"), 1, 1, 0, 0, 1);
    T skip = t_new(3, 1, H, F, 1);
    int have_skip = 0;
    for (int i = 0; i < c->wn_layers; i++) {
      const int last = i + 1 == c->wn_layers;
      T xo = t_new(3, 1, H, F, 1), so = t_new(3, 1, H, F, 1);
      orc_wavenet_layer(h.d, have_skip ? skip.d : NULL, bi_get(bi, "flow.flows.%d.enc.in_layers.%d", 2 * f, i, ".weight"),
                        bi_get(bi, "flow.flows.%d.enc.in_layers.%d", 2 * f, i, ".bias"),
                        bi_get(bi, "flow.flows.%d.enc.res_skip_layers.%d", 2 * f, i, ".weight"),
                        bi_get(bi, "flow.flows.%d.enc.res_skip_layers.%d", 2 * f, i, ".bias"), H, F, c->wn_kernel, 1, last,
                        xo.d, so.d);
      if (!last) { t_free(&h); h = xo; } else t_free(&xo);
      t_free(&skip);
      skip = so;
      have_skip = 1;
    }
    t_free(&h);
    T m = t_conv(&skip, bi_get(bi, "flow.flows.%d.post", 2 * f, 0, ".weight"), half, 1,
                 bi_get(bi, "flow.flows.%d.post", 2 * f, 0, ".bias"), 1, 1, 0, 0, 1);
    t_free(&skip);
    T x1n = t_binary(1, &x1, &m);
    x = t_new(3, 1, I, F, 1);
    orc_concat2_axis1(x0.d, 1, half, x1n.d, half, F, x.d);
    t_free(&x0); t_free(&x1); t_free(&m); t_free(&x1n);
  }
  return x;
}

ORC_API int orc_flow_reverse_forward(const piper_hip_voice_config* cfg, const float* blob, const float* zp, long F,
                                     float* z) {
  blob_index bi = {cfg, blob, NULL, 0, 0};
  piper_hip_layout_walk(cfg, bi_visit, &bi);
  T t = t_new(3, 1, cfg->inter, F, 1);
  memcpy(t.d, zp, (size_t)(cfg->inter * F) * sizeof(float));
  T o = orc_flow_reverse(&bi, &t);
  memcpy(z, o.d, (size_t)t_count(&o) * sizeof(float));
  t_free(&o); t_free(&t);
  free(bi.descs);
  return 0;
}

/* Text encoder (Piper models.TextEncoder + attentions.Encoder as exported). ids [T] → x [1,H,T]; stats [1,2I,T]. */
static T orc_text_encoder(const blob_index* bi, const int64_t* ids, long T_, T* stats) {
  const piper_hip_voice_config* c = bi->cfg;
  const long H = c->hidden, d = H / c->n_heads;
  /* Gather (GraphExecutor.swift:653-666) + Mul by sqrt(H) + Transpose */
  const float* emb = bi_get(bi, "enc_p.emb", 0, 0, ".weight");
  T e = t_new(3, 1, T_, H, 1);
  for (long t = 0; t < T_; t++) {
    long id = ids[t];
    if (id < 0) id += c->n_vocab; /* gather.metal:54-57: negative ids wrap once, anything still out of range reads 0.0 */
    if (id < 0 || id >= c->n_vocab) memset(e.d + t * H, 0, (size_t)H * sizeof(float));
    else memcpy(e.d + t * H, emb + id * H, (size_t)H * sizeof(float));
  }
  T sc = t_scalar(sqrtf((float)H));
  T es = t_binary(2, &e, &sc);
  const int p021[3] = {0, 2, 1};
  T x = t_transpose(&es, p021);
  t_free(&e); t_free(&sc); t_free(&es);
  for (int l = 0; l < c->n_layers; l++) {
    const char* P = "enc_p.encoder.attn_layers.%d.%s";
    (void)P;
    char nm[4][96];
    static const char* qkvo[4] = {"conv_q", "conv_k", "conv_v", "conv_o"};
    T proj[3];
    for (int j = 0; j < 3; j++) {
      snprintf(nm[j], sizeof nm[j], "enc_p.encoder.attn_layers.%d.%s", l, qkvo[j]);
      proj[j] = t_conv(&x, bi_get(bi, nm[j], 0, 0, ".weight"), H, 1, bi_get(bi, nm[j], 0, 0, ".bias"), 1, 1, 0, 0, 1);
    }
    T att = t_new(3, 1, H, T_, 1);
    orc_rel_attention(proj[0].d, proj[1].d, proj[2].d, bi_get(bi, "enc_p.encoder.attn_layers.%d.emb_rel_k", l, 0, ""),
                      bi_get(bi, "enc_p.encoder.attn_layers.%d.emb_rel_v", l, 0, ""), c->n_heads, d, T_, c->window, att.d);
    for (int j = 0; j < 3; j++) t_free(&proj[j]);
    snprintf(nm[3], sizeof nm[3], "enc_p.encoder.attn_layers.%d.conv_o", l);
    T y = t_conv(&att, bi_get(bi, nm[3], 0, 0, ".weight"), H, 1, bi_get(bi, nm[3], 0, 0, ".bias"), 1, 1, 0, 0, 1);
    t_free(&att);
    T x1 = t_new(3, 1, H, T_, 1);
    orc_add_layernorm(x.d, y.d, bi_get(bi, "enc_p.encoder.norm_layers_1.%d.gamma", l, 0, ""),
                      bi_get(bi, "enc_p.encoder.norm_layers_1.%d.beta", l, 0, ""), H, T_, 1e-5f, x1.d);
    t_free(&x); t_free(&y);
    const long kf = c->ffn_kernel, pl = (kf - 1) / 2, pr = kf / 2;
    T f1 = t_conv(&x1, bi_get(bi, "enc_p.encoder.ffn_layers.%d.conv_1", l, 0, ".weight"), c->ffn, kf,
                  bi_get(bi, "enc_p.encoder.ffn_layers.%d.conv_1", l, 0, ".bias"), 1, 1, pl, pr, 1);
    T r = t_unary(&f1, 0, 0);
    T f2 = t_conv(&r, bi_get(bi, "enc_p.encoder.ffn_layers.%d.conv_2", l, 0, ".weight"), H, kf,
                  bi_get(bi, "enc_p.encoder.ffn_layers.%d.conv_2", l, 0, ".bias"), 1, 1, pl, pr, 1);
    t_free(&f1); t_free(&r);
    x = t_new(3, 1, H, T_, 1);
    orc_add_layernorm(x1.d, f2.d, bi_get(bi, "enc_p.encoder.norm_layers_2.%d.gamma", l, 0, ""),
                      bi_get(bi, "enc_p.encoder.norm_layers_2.%d.beta", l, 0, ""), H, T_, 1e-5f, x.d);
    t_free(&x1); t_free(&f2);
  }
  *stats = t_conv(&x, bi_get(bi, "enc_p.proj", 0, 0, ".weight"), 2 * c->inter, 1, bi_get(bi, "enc_p.proj", 0, 0, ".bias"), 1, 1,
                  0, 0, 1);
  return x;
}

ORC_API int orc_text_encoder_forward(const piper_hip_voice_config* cfg, const float* blob, const int64_t* ids, long T_,
                                     float* enc_out, float* stats_out) {
  blob_index bi = {cfg, blob, NULL, 0, 0};
  piper_hip_layout_walk(cfg, bi_visit, &bi);
  T stats;
  T x = orc_text_encoder(&bi, ids, T_, &stats);
  if (enc_out) memcpy(enc_out, x.d, (size_t)t_count(&x) * sizeof(float));
  if (stats_out) memcpy(stats_out, stats.d, (size_t)t_count(&stats) * sizeof(float));
  t_free(&x); t_free(&stats);
  free(bi.descs);
  return 0;
}

/* Whole utterance: Piper SynthesizerTrn.infer with the duration predictor's output overridden (durations) and the
 * "main" RandomNormalLike injected (noise) — the reference's `overrides` mechanism (GraphExecutor.swift:101-104,
 * 2647-2651).  taps (any may be NULL): m_p/logs_p [I,T], z_p/z [I,F].  audio [F·Πrates]. */
ORC_API long orc_synthesize(const piper_hip_voice_config* cfg, const float* blob, const int64_t* ids, long T_,
                            const int32_t* durations, const float* noise, float noise_scale, float* audio, float* tap_enc,
                            float* tap_mp, float* tap_logsp, float* tap_zp, float* tap_z) {
  blob_index bi = {cfg, blob, NULL, 0, 0};
  piper_hip_layout_walk(cfg, bi_visit, &bi);
  const long I = cfg->inter;
  T stats;
  T x = orc_text_encoder(&bi, ids, T_, &stats);
  if (tap_enc) memcpy(tap_enc, x.d, (size_t)t_count(&x) * sizeof(float));
  t_free(&x);
  T m_p = t_slice(&stats, 1, 0, I, 1), logs_p = t_slice(&stats, 1, I, 2 * I, 1); /* Split */
  t_free(&stats);
  if (tap_mp) memcpy(tap_mp, m_p.d, (size_t)(I * T_) * sizeof(float));
  if (tap_logsp) memcpy(tap_logsp, logs_p.d, (size_t)(I * T_) * sizeof(float));
  long F = 0;
  for (long t = 0; t < T_; t++) F += durations[t];
  /* generate_path → attn [1,F,T] one-hot; m_p = matmul(attn, m_pᵀ)ᵀ (MatMul arm, GraphExecutor.swift:1862-1915) */
  T attn = t_new(3, 1, F, T_, 1);
  memset(attn.d, 0, (size_t)(F * T_) * sizeof(float));
  {
    long f = 0;
    for (long t = 0; t < T_; t++)
      for (long j = 0; j < durations[t]; j++, f++) attn.d[f * T_ + t] = 1.0f;
  }
  const int p021[3] = {0, 2, 1};
  T mT = t_transpose(&m_p, p021), lT = t_transpose(&logs_p, p021);
  T mE = t_matmul(&attn, &mT), lE = t_matmul(&attn, &lT); /* [1,F,I] */
  T m_e = t_transpose(&mE, p021), l_e = t_transpose(&lE, p021);
  t_free(&attn); t_free(&mT); t_free(&lT); t_free(&mE); t_free(&lE); t_free(&m_p); t_free(&logs_p);
  /* z_p = m_p + noise * exp(logs_p) * noise_scale */
  T nz = t_new(3, 1, I, F, 1);
  if (noise) memcpy(nz.d, noise, (size_t)(I * F) * sizeof(float));
  else memset(nz.d, 0, (size_t)(I * F) * sizeof(float));
  T ex = t_unary(&l_e, 4, 0);
  T ne = t_binary(2, &nz, &ex);
  T ns = t_scalar(noise_scale);
  T nes = t_binary(2, &ne, &ns);
  T zp = t_binary(0, &m_e, &nes);
  t_free(&nz); t_free(&ex); t_free(&ne); t_free(&ns); t_free(&nes); t_free(&m_e); t_free(&l_e);
  if (tap_zp) memcpy(tap_zp, zp.d, (size_t)(I * F) * sizeof(float));
  T z = orc_flow_reverse(&bi, &zp);
  t_free(&zp);
  if (tap_z) memcpy(tap_z, z.d, (size_t)(I * F) * sizeof(float));
  T o = orc_generator(&bi, &z);
  t_free(&z);
  const long ns_out = t_count(&o);
  if (audio) memcpy(audio, o.d, (size_t)ns_out * sizeof(float));
  t_free(&o);
  free(bi.descs);
  return ns_out;
}

/* ------------------------------------------------------------------------------------------------
 * Stochastic duration predictor, inference direction (VITS models.StochasticDurationPredictor.forward(reverse=True) as
 * Piper exports it). In the reference these are ordinary graph nodes: Conv (depthwise through `group`), the LayerNorm chain
 * (GraphExecutor.swift:2071-2125), GELU as Div / Erf / Add / Mul (Erf: elementwise.metal:292-312 is A&S 7.1.26, restated
 * with erff — inside the stated tolerance), Softmax, Softplus (CPUBackend.swift:75-110 form), and the
 * rational-quadratic spline spelled out with CumSum / GreaterOrEqual / ReduceSum / GatherElements / Where / Sqrt
 * (GraphExecutor.swift:2379-2645). The reference holds no vectors for it ("parity unpinned"): pinned against
 * tests/torch_ref.py, which is itself checked against transformers' VitsStochasticDurationPredictor (max |Δ| = 0).
 * ---------------------------------------------------------------------------------------------- */
static T orc_gelu(const T* x) { /* 0.5·x·(1 + erf(x / √2)): the graph's Div(√2) → Erf → Add(1) → Mul(x) → Mul(0.5) */
  T y = *x;
  const long n = t_count(x);
  y.d = (float*)malloc((size_t)n * sizeof(float));
  for (long i = 0; i < n; i++) {
    const float v = x->d[i];
    const float e = (float)erf((double)(v / 1.4142135381698608f));
    y.d[i] = (v * (e + 1.0f)) * 0.5f;
  }
  return y;
}

static T orc_ln(const T* x, const float* gamma, const float* beta) {
  T y = t_new(3, 1, x->s[1], x->s[2], 1);
  orc_add_layernorm(x->d, NULL, gamma, beta, x->s[1], x->s[2], 1e-5f, y.d);
  return y;
}

/* modules.DDSConv: x (+ g); per layer i: y = dw_conv(x, k, dilation k^i) → LN → GELU → 1×1 conv → LN → GELU; x += y */
static T orc_dds(const blob_index* bi, const char* base, const T* x_in, const T* g) {
  const piper_hip_voice_config* c = bi->cfg;
  const long H = c->hidden, K = c->dp_kernel;
  T x = g ? t_binary(0, x_in, g) : t_clone(x_in);
  long dil = 1;
  char f[96];
  for (int i = 0; i < c->dp_dds_layers; i++) {
    snprintf(f, sizeof f, "%s.convs.convs_sep.%%d", base);
    T y = t_conv(&x, bi_get(bi, f, i, 0, ".weight"), H, K, bi_get(bi, f, i, 0, ".bias"), 1, dil, (K * dil - dil) / 2, (K * dil - dil) / 2, H);
    snprintf(f, sizeof f, "%s.convs.norms_1.%%d", base);
    T n1 = orc_ln(&y, bi_get(bi, f, i, 0, ".gamma"), bi_get(bi, f, i, 0, ".beta"));
    T g1 = orc_gelu(&n1);
    snprintf(f, sizeof f, "%s.convs.convs_1x1.%%d", base);
    T y2 = t_conv(&g1, bi_get(bi, f, i, 0, ".weight"), H, 1, bi_get(bi, f, i, 0, ".bias"), 1, 1, 0, 0, 1);
    snprintf(f, sizeof f, "%s.convs.norms_2.%%d", base);
    T n2 = orc_ln(&y2, bi_get(bi, f, i, 0, ".gamma"), bi_get(bi, f, i, 0, ".beta"));
    T g2 = orc_gelu(&n2);
    T xn = t_binary(0, &x, &g2);
    t_free(&y); t_free(&n1); t_free(&g1); t_free(&y2); t_free(&n2); t_free(&g2); t_free(&x);
    x = xn;
    dil *= K;
  }
  return x;
}

static float orc_softplus(float v) { return v > 0 ? v + (float)log(1.0 + exp((double)-v)) : (float)log(1.0 + exp((double)v)); }

/* transforms.unconstrained_rational_quadratic_spline(inverse=True, tails='linear'), one element: x = z1[t], h = the 3·bins − 1
 * channels of ConvFlow.proj at t. float arithmetic in the graph's order. */
static float orc_spline_inverse(float x, const float* h, long stride, int nb, float B, float filter_channels) {
  if (!(x >= -B && x <= B)) return x; /* outside the interval: identity */
  const float mbw = 1e-3f, mbh = 1e-3f, md = 1e-3f;
  const float inv = sqrtf(filter_channels);
  float w[32], hh[32], cw[33], ch[33], d[33];
  float mw = -INFINITY, mh = -INFINITY;
  for (int i = 0; i < nb; i++) {
    w[i] = h[i * stride] / inv;
    hh[i] = h[(nb + i) * stride] / inv;
    if (w[i] > mw) mw = w[i];
    if (hh[i] > mh) mh = hh[i];
  }
  float sw = 0, sh = 0;
  for (int i = 0; i < nb; i++) { w[i] = expf(w[i] - mw); sw += w[i]; hh[i] = expf(hh[i] - mh); sh += hh[i]; }
  cw[0] = 0; ch[0] = 0;
  for (int i = 0; i < nb; i++) {
    const float wi = mbw + (1 - mbw * nb) * (w[i] * (1.0f / sw));
    const float hi = mbh + (1 - mbh * nb) * (hh[i] * (1.0f / sh));
    cw[i + 1] = cw[i] + wi;
    ch[i + 1] = ch[i] + hi;
  }
  for (int i = 0; i <= nb; i++) { cw[i] = (2 * B) * cw[i] + -B; ch[i] = (2 * B) * ch[i] + -B; }
  cw[0] = -B; cw[nb] = B; ch[0] = -B; ch[nb] = B;
  const float constant = (float)log(exp(1.0 - (double)md) - 1.0);
  d[0] = md + orc_softplus(constant);
  d[nb] = d[0];
  for (int i = 1; i < nb; i++) d[i] = md + orc_softplus(h[(2 * nb + i - 1) * stride]);
  int idx = -1; /* Σ (x ≥ location) − 1, the last location nudged by 1e-6 */
  for (int i = 0; i <= nb; i++) idx += x >= (i == nb ? ch[i] + 1e-6f : ch[i]);
  if (idx < 0) idx = 0;
  if (idx > nb - 1) idx = nb - 1;
  const float ibw = cw[idx + 1] - cw[idx], ih = ch[idx + 1] - ch[idx];
  const float idl = ih / ibw;
  const float i1 = d[idx] + d[idx + 1] - 2 * idl;
  const float i2 = x - ch[idx];
  const float i3 = i2 * i1;
  const float a = ih * (idl - d[idx]) + i3;
  const float b = ih * d[idx] - i3;
  const float cc = -idl * i2;
  const float disc = b * b - 4 * a * cc;
  const float root = (2 * cc) / (-b - sqrtf(disc));
  return root * ibw + cw[idx];
}

/* enc_out [H,T] (text-encoder output, the predictor's `x`), dp_noise [2,T] (the `dp` RandomNormalLike tensor; NULL = zeros),
 * noise_w (scales[2]) → logw [T]. */
ORC_API int orc_duration_logw(const piper_hip_voice_config* cfg, const float* blob, const float* enc_out, long T_, const float* dp_noise,
                              float noise_w, float* logw) {
  if (!cfg->dp_present) return -1;
  blob_index bi = {cfg, blob, NULL, 0, 0};
  piper_hip_layout_walk(cfg, bi_visit, &bi);
  const long H = cfg->hidden;
  const int nb = cfg->dp_bins;
  T x0 = t_new(3, 1, H, T_, 1);
  memcpy(x0.d, enc_out, (size_t)(H * T_) * sizeof(float));
  T x1 = t_conv(&x0, bi_get(&bi, "dp.pre", 0, 0, ".weight"), H, 1, bi_get(&bi, "dp.pre", 0, 0, ".bias"), 1, 1, 0, 0, 1);
  T x2 = orc_dds(&bi, "dp", &x1, NULL);
  T x = t_conv(&x2, bi_get(&bi, "dp.proj", 0, 0, ".weight"), H, 1, bi_get(&bi, "dp.proj", 0, 0, ".bias"), 1, 1, 0, 0, 1);
  t_free(&x0); t_free(&x1); t_free(&x2);
  float* z = (float*)calloc((size_t)(2 * T_), sizeof(float)); /* [2][T] */
  if (dp_noise)
    for (long i = 0; i < 2 * T_; i++) z[i] = dp_noise[i] * noise_w;
  for (int f = 2 * cfg->dp_n_flows - 1; f > 1; f -= 2) { /* Flip, ConvFlow f (reverse) */
    for (long t = 0; t < T_; t++) { const float a = z[t]; z[t] = z[T_ + t]; z[T_ + t] = a; }
    char base[64];
    snprintf(base, sizeof base, "dp.flows.%d", f);
    T z0 = t_new(3, 1, 1, T_, 1);
    memcpy(z0.d, z, (size_t)T_ * sizeof(float));
    char nm[96];
    snprintf(nm, sizeof nm, "%s.pre", base);
    T h0 = t_conv(&z0, bi_get(&bi, nm, 0, 0, ".weight"), H, 1, bi_get(&bi, nm, 0, 0, ".bias"), 1, 1, 0, 0, 1);
    T h1 = orc_dds(&bi, base, &h0, &x);
    snprintf(nm, sizeof nm, "%s.proj", base);
    T h2 = t_conv(&h1, bi_get(&bi, nm, 0, 0, ".weight"), 3 * nb - 1, 1, bi_get(&bi, nm, 0, 0, ".bias"), 1, 1, 0, 0, 1);
    for (long t = 0; t < T_; t++) z[T_ + t] = orc_spline_inverse(z[T_ + t], h2.d + t, T_, nb, cfg->dp_tail_bound, (float)H);
    t_free(&z0); t_free(&h0); t_free(&h1); t_free(&h2);
  }
  for (long t = 0; t < T_; t++) { const float a = z[t]; z[t] = z[T_ + t]; z[T_ + t] = a; } /* Flip before the ElementwiseAffine */
  const float* m = bi_get(&bi, "dp.flows.0.m", 0, 0, "");
  const float* lg = bi_get(&bi, "dp.flows.0.logs", 0, 0, "");
  for (long t = 0; t < T_; t++) logw[t] = (z[t] - m[0]) * (float)exp((double)-lg[0]);
  t_free(&x);
  free(z);
  free(bi.descs);
  return 0;
}

/* w = exp(logw)·length_scale; durations = ceil(w) (Piper infer; Exp / Mul / Ceil arms), as int32 frames per id (≥ 0). */
ORC_API void orc_durations_from_logw(const float* logw, long T_, float length_scale, int32_t* dur) {
  for (long t = 0; t < T_; t++) {
    const float w = (float)exp((double)logw[t]) * length_scale;
    const float cw = (float)ceil((double)w);
    dur[t] = cw > 0 ? (cw < 1e6f ? (int32_t)cw : 1000000) : 0;
  }
}
