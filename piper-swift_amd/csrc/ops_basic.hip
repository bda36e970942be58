// ops_basic.hip — elementwise, layout, reduction, softmax, LayerNorm and batched matmul entry points (gfx950).
// These are HBM-bound byte movers: one pass, 16-byte accesses where the layout allows, ≥ 4 waves per CU in flight.
#include "common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t work_items, int64_t per_block, int num_cus) {
  int64_t g = ph::ceil_div(work_items, per_block);
  const int64_t cap = (int64_t)num_cus * 8;  // grid-stride above ~2048 blocks (guide §6 G11)
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- unary (elementwise.metal:165-312)
template <int OP>
__device__ __forceinline__ float unary_apply(float x, float alpha) {
  if constexpr (OP == PIPER_HIP_RELU) return x > 0.0f ? x : 0.0f;
  if constexpr (OP == PIPER_HIP_LEAKYRELU) return x >= 0.0f ? x : alpha * x;
  if constexpr (OP == PIPER_HIP_TANH) return tanhf(x);
  if constexpr (OP == PIPER_HIP_SIGMOID) {
    if (x >= 0.0f) {
      const float z = expf(-x);
      return 1.0f / (1.0f + z);
    }
    const float z = expf(x);
    return z / (1.0f + z);
  }
  if constexpr (OP == PIPER_HIP_EXP) return expf(x);
  if constexpr (OP == PIPER_HIP_NEG) return -x;
  if constexpr (OP == PIPER_HIP_SQRT) return sqrtf(x);
  if constexpr (OP == PIPER_HIP_SOFTPLUS) return x > 0.0f ? x + logf(1.0f + expf(-x)) : logf(1.0f + expf(x));
  if constexpr (OP == PIPER_HIP_CEIL) return ceilf(x);
  if constexpr (OP == PIPER_HIP_ERF) return erff(x);
  return x;
}

template <int OP>
__global__ __launch_bounds__(kBlock) void unary_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                       float alpha, int vec_ok) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * kBlock;
  if (vec_ok) {
    const size_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (size_t i = tid; i < n4; i += stride) {
      float4 v = x4[i];
      v.x = unary_apply<OP>(v.x, alpha);
      v.y = unary_apply<OP>(v.y, alpha);
      v.z = unary_apply<OP>(v.z, alpha);
      v.w = unary_apply<OP>(v.w, alpha);
      y4[i] = v;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += stride) y[i] = unary_apply<OP>(x[i], alpha);
  } else {
    for (size_t i = tid; i < n; i += stride) y[i] = unary_apply<OP>(x[i], alpha);
  }
}

// ---------------------------------------------------------------- rank-≤4 index helpers
struct Idx4 {
  int64_t out_shape[4];
  int64_t a_stride[4];
  int64_t b_stride[4];
  int64_t aux[4];
  int rank;
};

template <int OP>
__device__ __forceinline__ float binary_apply(float a, float b) {
  if constexpr (OP == PIPER_HIP_ADD) return a + b;
  if constexpr (OP == PIPER_HIP_SUB) return a - b;
  if constexpr (OP == PIPER_HIP_MUL) return a * b;
  if constexpr (OP == PIPER_HIP_DIV) return a / b;
  if constexpr (OP == PIPER_HIP_POW) return powf(a, b);
  return a;
}

// generic broadcast (elementwise.metal:24-130): stride 0 on broadcast dims
template <int OP>
__global__ __launch_bounds__(kBlock) void binary_bcast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ out, size_t n, Idx4 p) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    size_t rem = i;
    int64_t ia = 0, ib = 0;
#pragma unroll
    for (int d = 3; d >= 0; d--) {
      if (d < p.rank) {
        const int64_t c = (int64_t)(rem % (size_t)p.out_shape[d]);
        rem /= (size_t)p.out_shape[d];
        ia += c * p.a_stride[d];
        ib += c * p.b_stride[d];
      }
    }
    out[i] = binary_apply<OP>(a[ia], b[ib]);
  }
}

// same-shape or scalar-b fast path
template <int OP>
__global__ __launch_bounds__(kBlock) void binary_flat_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ out, size_t n, int b_scalar, int vec_ok) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * kBlock;
  const float bs = b_scalar ? b[0] : 0.0f;
  if (vec_ok) {
    const size_t n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (size_t i = tid; i < n4; i += stride) {
      const float4 va = a4[i];
      float4 vb;
      if (b_scalar) vb = make_float4(bs, bs, bs, bs);
      else vb = b4[i];
      float4 r;
      r.x = binary_apply<OP>(va.x, vb.x);
      r.y = binary_apply<OP>(va.y, vb.y);
      r.z = binary_apply<OP>(va.z, vb.z);
      r.w = binary_apply<OP>(va.w, vb.w);
      o4[i] = r;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = binary_apply<OP>(a[i], b_scalar ? bs : b[i]);
  } else {
    for (size_t i = tid; i < n; i += stride) out[i] = binary_apply<OP>(a[i], b_scalar ? bs : b[i]);
  }
}

// ---------------------------------------------------------------- layout movers
// gather-style: every output element computes its source offset, or "fill" when outside (pad)
struct Gather4 {
  int64_t out_shape[4];
  int64_t in_stride[4];  // element stride applied to (coord*mul + add)
  int64_t add[4];        // pad: -begin ; slice: start on the axis
  int64_t mul[4];        // slice: step on the axis; 1 otherwise
  int64_t in_shape[4];   // bounds for pad
  int rank;
  int check_bounds;
};

__global__ __launch_bounds__(kBlock) void gather4_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n,
                                                         Gather4 p, float fill) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    size_t rem = i;
    int64_t off = 0;
    bool inside = true;
#pragma unroll
    for (int d = 3; d >= 0; d--) {
      if (d < p.rank) {
        const int64_t c = (int64_t)(rem % (size_t)p.out_shape[d]);
        rem /= (size_t)p.out_shape[d];
        const int64_t s = c * p.mul[d] + p.add[d];
        if (p.check_bounds && (s < 0 || s >= p.in_shape[d])) inside = false;
        off += s * p.in_stride[d];
      }
    }
    out[i] = inside ? x[off] : fill;
  }
}

// ---------------------------------------------------------------- reductions over the last dim
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// one wave per row (reduce.metal:12-25 is one THREAD per row)
__global__ __launch_bounds__(kBlock) void reduce_mean_lastdim_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                     int64_t rows, int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float* row = x + r * cols;
    float s = 0.0f;
    for (int64_t c = lane; c < cols; c += 64) s += row[c];
    s = wave_sum(s);
    if (lane == 0) y[r] = s / (float)cols;
  }
}

// softmax over the last dim (softmax.metal:13-41): one wave per row when cols ≤ 2048 (row kept in registers),
// one 256-thread block per row otherwise. Max-subtracted, exp, sum, multiply by 1/sum — the reference's three steps.
constexpr int kSoftmaxRegs = 32;  // 64 lanes × 32 = 2048 columns in registers
__global__ __launch_bounds__(kBlock) void softmax_wave_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int64_t rows, int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float* row = x + r * cols;
    float v[kSoftmaxRegs];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < kSoftmaxRegs; j++) {
      const int64_t c = lane + 64 * j;
      v[j] = c < cols ? row[c] : -INFINITY;
      m = fmaxf(m, v[j]);
    }
    m = wave_max(m);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < kSoftmaxRegs; j++) {
      const int64_t c = lane + 64 * j;
      v[j] = c < cols ? expf(v[j] - m) : 0.0f;
      s += v[j];
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int j = 0; j < kSoftmaxRegs; j++) {
      const int64_t c = lane + 64 * j;
      if (c < cols) y[r * cols + c] = v[j] * inv;
    }
  }
}

__global__ __launch_bounds__(kBlock) void softmax_block_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int64_t rows, int64_t cols) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* row = x + r * cols;
    float m = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) m = fmaxf(m, row[c]);
    m = wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      const float e = expf(row[c] - m);
      y[r * cols + c] = e;
      s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float inv = 1.0f / s;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) y[r * cols + c] *= inv;
  }
}

// ---------------------------------------------------------------- channel LayerNorm with fused residual
// x,y [N,C,T]; normalise over C for each (n,t). Block = 16 time steps × 16 channel lanes.  Every thread first issues ALL
// of its loads (≤ 2·kLnMaxV independent requests in flight), keeps x+y in registers, and the two reductions go through
// LDS — one memory round trip instead of three dependent passes (at T≈100 the op is pure latency, not bandwidth).
constexpr int kLnT = 16, kLnG = 16;
// kLnMaxV = register slots per thread (C ≤ 16·kLnMaxV).  Instantiated at 12 (C ≤ 192, every Piper voice) and 32: the
// unrolled body is straight-line code fetched cold on every launch, so the small instance is ~2.5× less I-cache traffic.
template <int kLnMaxV>
__global__ __launch_bounds__(kBlock) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ out, int C, int T, float eps) {
  __shared__ float red[kLnG][kLnT + 1];
  const int tt = threadIdx.x % kLnT, g = threadIdx.x / kLnT;
  const int64_t n = blockIdx.y;
  const int t = blockIdx.x * kLnT + tt;
  const bool ok = t < T;
  const float* xb = x + n * (int64_t)C * T;
  const float* yb = y ? y + n * (int64_t)C * T : nullptr;
  float v[kLnMaxV], w[kLnMaxV], ga[kLnMaxV], be[kLnMaxV];
#pragma unroll
  for (int i = 0; i < kLnMaxV; i++) {
    v[i] = 0.0f; w[i] = 0.0f; ga[i] = 0.0f; be[i] = 0.0f;
    if (kLnG * i < C) {  // block-uniform: rows of 16 channels that do not exist issue no loads at all
      const int c = g + kLnG * i;
      const bool in = ok && c < C;
      const int64_t idx = in ? (int64_t)c * T + t : 0;
      v[i] = xb[idx];
      if (yb) w[i] = yb[idx];
      ga[i] = gamma[c < C ? c : 0];  // fetched with x/y so the normalise step needs no second memory round trip
      be[i] = beta[c < C ? c : 0];
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < kLnMaxV; i++) {
    const int c = g + kLnG * i;
    v[i] = (ok && c < C) ? v[i] + w[i] : 0.0f;
    s += v[i];
  }
  red[g][tt] = s;
  __syncthreads();
  float mean = 0.0f;
#pragma unroll
  for (int i = 0; i < kLnG; i++) mean += red[i][tt];
  mean = mean / (float)C;
  __syncthreads();
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < kLnMaxV; i++) {
    const int c = g + kLnG * i;
    const float d = (c < C) ? v[i] - mean : 0.0f;
    v[i] = d;
    q += d * d;
  }
  red[g][tt] = q;
  __syncthreads();
  float var = 0.0f;
#pragma unroll
  for (int i = 0; i < kLnG; i++) var += red[i][tt];
  var = var / (float)C;
  const float sd = sqrtf(var + eps);
  if (ok) {
#pragma unroll
    for (int i = 0; i < kLnMaxV; i++) {
      const int c = g + kLnG * i;
      if (c < C) out[n * (int64_t)C * T + (int64_t)c * T + t] = (v[i] / sd) * ga[i] + be[i];
    }
  }
}

// any C: three strided passes (only used when C > 512)
__global__ __launch_bounds__(kBlock) void add_layernorm_big_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ out, int C, int T, float eps) {
  __shared__ float red[kLnG][kLnT + 1];
  const int tt = threadIdx.x % kLnT, g = threadIdx.x / kLnT;
  const int64_t n = blockIdx.y;
  const int t = blockIdx.x * kLnT + tt;
  const bool ok = t < T;
  const float* xb = x + n * (int64_t)C * T;
  const float* yb = y ? y + n * (int64_t)C * T : nullptr;
  float s = 0.0f;
  if (ok)
    for (int c = g; c < C; c += kLnG) s += xb[(int64_t)c * T + t] + (yb ? yb[(int64_t)c * T + t] : 0.0f);
  red[g][tt] = s;
  __syncthreads();
  float mean = 0.0f;
  for (int i = 0; i < kLnG; i++) mean += red[i][tt];
  mean = mean / (float)C;
  __syncthreads();
  float q = 0.0f;
  if (ok)
    for (int c = g; c < C; c += kLnG) {
      const float d = (xb[(int64_t)c * T + t] + (yb ? yb[(int64_t)c * T + t] : 0.0f)) - mean;
      q += d * d;
    }
  red[g][tt] = q;
  __syncthreads();
  float var = 0.0f;
  for (int i = 0; i < kLnG; i++) var += red[i][tt];
  var = var / (float)C;
  const float sd = sqrtf(var + eps);
  if (ok)
    for (int c = g; c < C; c += kLnG) {
      const float d = (xb[(int64_t)c * T + t] + (yb ? yb[(int64_t)c * T + t] : 0.0f)) - mean;
      out[n * (int64_t)C * T + (int64_t)c * T + t] = (d / sd) * gamma[c] + beta[c];
    }
}

// ---------------------------------------------------------------- batched matmul, fp32, stride-0 lead broadcast
// C[b] = A[b or 0] · B[b or 0]; 64×64 output tile per 256-thread block, K staged 16 at a time through LDS,
// 4×4 outputs per thread. (matmul.metal:22-49 is one thread per output with a K-long global-memory loop.)
constexpr int kMmT = 64, kMmK = 16;
__global__ __launch_bounds__(kBlock) void matmul_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        float* __restrict__ Cm, int M, int N, int K, int64_t a_bs,
                                                        int64_t b_bs) {
  __shared__ float As[kMmK][kMmT + 4];
  __shared__ float Bs[kMmK][kMmT + 4];
  const int64_t bi = blockIdx.z;
  const float* a = A + bi * a_bs;
  const float* b = B + bi * b_bs;
  float* c = Cm + bi * (int64_t)M * N;
  const int m0 = blockIdx.y * kMmT, n0 = blockIdx.x * kMmT;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += kMmK) {
    // A tile: 64 rows × 16 k  (thread → row = tid/4, 4 consecutive k)
    {
      const int r = threadIdx.x >> 2, kk = (threadIdx.x & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int gm = m0 + r, gk = k0 + kk + j;
        As[kk + j][r] = (gm < M && gk < K) ? a[(int64_t)gm * K + gk] : 0.0f;
      }
    }
    // B tile: 16 k × 64 cols (thread → k = tid/16, 4 consecutive cols)
    {
      const int kk = threadIdx.x >> 4, cc = (threadIdx.x & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int gk = k0 + kk, gn = n0 + cc + j;
        Bs[kk][cc + j] = (gk < K && gn < N) ? b[(int64_t)gk * N + gn] : 0.0f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kMmK; kk++) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) av[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) bv[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int gm = m0 + ty * 4 + i, gn = n0 + tx * 4 + j;
      if (gm < M && gn < N) c[(int64_t)gm * N + gn] = acc[i][j];
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline size_t shape_count(const int64_t* s, int rank) {
  size_t n = 1;
  for (int i = 0; i < rank; i++) n *= (size_t)s[i];
  return n;
}

inline void row_major_strides(const int64_t* shape, int rank, int64_t* st) {
  if (rank == 0) return;
  st[rank - 1] = 1;
  for (int i = rank - 2; i >= 0; i--) st[i] = st[i + 1] * shape[i + 1];
}

int validate_shape(const int64_t* s, int rank, const char* what) {
  if (!s) PH_FAIL(PIPER_HIP_ERR_ARG, "%s: null shape", what);
  for (int i = 0; i < rank; i++)
    if (s[i] < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "%s: negative dimension", what);
  return 0;
}

}  // namespace

// ======================================================================== entry points

PH_EXPORT int piper_hip_unary_f32(piper_hip_ctx* ctx, piper_hip_unary_op op, const float* x, size_t count, float alpha,
                                  float** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if ((int)op < 0 || (int)op > PIPER_HIP_ERF) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "unary op %d", (int)op);
  if (!x && count) PH_FAIL(PIPER_HIP_ERR_ARG, "unary: null input");
  int rc = ph::ensure_out(ctx, out, count, 0);
  if (rc) return rc;
  if (count == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  const int vec = aligned16(x) && aligned16(*out);
  const int grid = grid_for((int64_t)count, kBlock * 4, ctx->num_cus);
#define PH_UNARY(OPC) \
  case OPC: hipLaunchKernelGGL(unary_kernel<OPC>, dim3(grid), dim3(kBlock), 0, ss.s, x, *out, count, alpha, vec); break;
  switch (op) {
    PH_UNARY(PIPER_HIP_RELU)
    PH_UNARY(PIPER_HIP_LEAKYRELU)
    PH_UNARY(PIPER_HIP_TANH)
    PH_UNARY(PIPER_HIP_SIGMOID)
    PH_UNARY(PIPER_HIP_EXP)
    PH_UNARY(PIPER_HIP_NEG)
    PH_UNARY(PIPER_HIP_SQRT)
    PH_UNARY(PIPER_HIP_SOFTPLUS)
    PH_UNARY(PIPER_HIP_CEIL)
    PH_UNARY(PIPER_HIP_ERF)
  }
#undef PH_UNARY
  return ss.finish("unary_f32");
}

PH_EXPORT int piper_hip_binary_broadcast_f32(piper_hip_ctx* ctx, piper_hip_binary_op op, const float* a,
                                             const int64_t* a_shape, int a_rank, const float* b, const int64_t* b_shape,
                                             int b_rank, float** out, int64_t* out_shape, int* out_rank,
                                             piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if ((int)op < 0 || (int)op > PIPER_HIP_POW) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "binary op %d", (int)op);
  if (a_rank < 0 || b_rank < 0 || a_rank > 4 || b_rank > 4)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "Broadcast rank %d not supported", a_rank > b_rank ? a_rank : b_rank);
  if (validate_shape(a_shape, a_rank, "binary a") || validate_shape(b_shape, b_rank, "binary b")) return PIPER_HIP_ERR_SHAPE;
  const int r = a_rank > b_rank ? a_rank : b_rank;
  Idx4 p{};
  p.rank = r;
  int64_t pa[4], pb[4], sa[4], sb[4];
  for (int i = 0; i < r; i++) {
    pa[i] = i < r - a_rank ? 1 : a_shape[i - (r - a_rank)];
    pb[i] = i < r - b_rank ? 1 : b_shape[i - (r - b_rank)];
    if (pa[i] == pb[i]) p.out_shape[i] = pa[i];
    else if (pa[i] == 1) p.out_shape[i] = pb[i];
    else if (pb[i] == 1) p.out_shape[i] = pa[i];
    else PH_FAIL(PIPER_HIP_ERR_SHAPE, "Cannot broadcast dim %d: %lld vs %lld", i, (long long)pa[i], (long long)pb[i]);
  }
  row_major_strides(pa, r, sa);
  row_major_strides(pb, r, sb);
  bool same = true;
  for (int i = 0; i < r; i++) {
    p.a_stride[i] = pa[i] == 1 ? 0 : sa[i];
    p.b_stride[i] = pb[i] == 1 ? 0 : sb[i];
    if (pa[i] != p.out_shape[i] || pb[i] != p.out_shape[i]) same = false;
    if (out_shape) out_shape[i] = p.out_shape[i];
  }
  if (out_rank) *out_rank = r;
  const size_t n = shape_count(p.out_shape, r);
  const size_t nb = shape_count(pb, r), na = shape_count(pa, r);
  if ((!a || !b) && n) PH_FAIL(PIPER_HIP_ERR_ARG, "binary: null input");
  int rc = ph::ensure_out(ctx, out, n, 0);
  if (rc) return rc;
  if (n == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  const bool b_scalar = (nb == 1 && na == n);
  const int grid = grid_for((int64_t)n, kBlock * 4, ctx->num_cus);
#define PH_BIN(OPC)                                                                                                   \
  case OPC:                                                                                                           \
    if (same || b_scalar) {                                                                                           \
      const int vec = aligned16(a) && aligned16(*out) && (b_scalar || aligned16(b));                                  \
      hipLaunchKernelGGL(binary_flat_kernel<OPC>, dim3(grid), dim3(kBlock), 0, ss.s, a, b, *out, n, (int)b_scalar, vec); \
    } else {                                                                                                          \
      hipLaunchKernelGGL(binary_bcast_kernel<OPC>, dim3(grid), dim3(kBlock), 0, ss.s, a, b, *out, n, p);              \
    }                                                                                                                 \
    break;
  switch (op) {
    PH_BIN(PIPER_HIP_ADD)
    PH_BIN(PIPER_HIP_SUB)
    PH_BIN(PIPER_HIP_MUL)
    PH_BIN(PIPER_HIP_DIV)
    PH_BIN(PIPER_HIP_POW)
  }
#undef PH_BIN
  return ss.finish("binary_broadcast_f32");
}

static int launch_gather(piper_hip_ctx* ctx, const float* x, float** out, const Gather4& g, float fill,
                         piper_hip_stream stream, const char* what) {
  const size_t n = shape_count(g.out_shape, g.rank);
  int rc = ph::ensure_out(ctx, out, n, 0);
  if (rc) return rc;
  if (n == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  const int grid = grid_for((int64_t)n, kBlock * 4, ctx->num_cus);
  hipLaunchKernelGGL(gather4_kernel, dim3(grid), dim3(kBlock), 0, ss.s, x, *out, n, g, fill);
  return ss.finish(what);
}

PH_EXPORT int piper_hip_pad_constant_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank,
                                         const int64_t* pads, float value, float** out, int64_t* out_shape,
                                         piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (rank < 1 || rank > 4) PH_FAIL(PIPER_HIP_ERR_SHAPE, "pad: rank %d not supported (1..4)", rank);
  if (validate_shape(shape, rank, "pad") || !pads) return PIPER_HIP_ERR_SHAPE;
  Gather4 g{};
  g.rank = rank;
  g.check_bounds = 1;
  int64_t st[4];
  row_major_strides(shape, rank, st);
  for (int d = 0; d < rank; d++) {
    if (pads[d] < 0 || pads[rank + d] < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "pad: negative pads not supported");
    g.out_shape[d] = shape[d] + pads[d] + pads[rank + d];
    g.in_stride[d] = st[d];
    g.add[d] = -pads[d];
    g.mul[d] = 1;
    g.in_shape[d] = shape[d];
    if (out_shape) out_shape[d] = g.out_shape[d];
  }
  return launch_gather(ctx, x, out, g, value, stream, "pad_constant_f32");
}

PH_EXPORT int piper_hip_slice_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank, int axis,
                                  int64_t start, int64_t end, int64_t step, float** out, int64_t* out_shape,
                                  piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (rank < 1 || rank > 4) PH_FAIL(PIPER_HIP_ERR_SHAPE, "slice: rank %d not supported (1..4)", rank);
  if (validate_shape(shape, rank, "slice")) return PIPER_HIP_ERR_SHAPE;
  if (axis < 0 || axis >= rank) PH_FAIL(PIPER_HIP_ERR_SHAPE, "slice: axis %d out of range", axis);
  if (step == 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "slice: step must be non-zero");
  int64_t cnt = 0;
  if (step > 0) { if (end > start) cnt = (end - start + step - 1) / step; }
  else { if (start > end) cnt = (start - end + (-step) - 1) / (-step); }
  if (cnt > 0) {
    const int64_t last = start + (cnt - 1) * step;
    if (start < 0 || start >= shape[axis] || last < 0 || last >= shape[axis])
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "slice: range [%lld,%lld) step %lld outside dim %lld", (long long)start, (long long)end,
              (long long)step, (long long)shape[axis]);
  }
  Gather4 g{};
  g.rank = rank;
  int64_t st[4];
  row_major_strides(shape, rank, st);
  for (int d = 0; d < rank; d++) {
    g.out_shape[d] = d == axis ? cnt : shape[d];
    g.in_stride[d] = st[d];
    g.add[d] = d == axis ? start : 0;
    g.mul[d] = d == axis ? step : 1;
    g.in_shape[d] = shape[d];
    if (out_shape) out_shape[d] = g.out_shape[d];
  }
  return launch_gather(ctx, x, out, g, 0.0f, stream, "slice_f32");
}

PH_EXPORT int piper_hip_transpose_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank,
                                      const int32_t* perm, float** out, int64_t* out_shape, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (rank < 1 || rank > 4) PH_FAIL(PIPER_HIP_ERR_SHAPE, "transposeF32 supports rank 1..4 (got %d)", rank);
  if (validate_shape(shape, rank, "transpose") || !perm) return PIPER_HIP_ERR_SHAPE;
  int seen = 0;
  for (int d = 0; d < rank; d++) {
    if (perm[d] < 0 || perm[d] >= rank || (seen >> perm[d]) & 1) PH_FAIL(PIPER_HIP_ERR_SHAPE, "transpose: invalid perm");
    seen |= 1 << perm[d];
  }
  Gather4 g{};
  g.rank = rank;
  int64_t st[4];
  row_major_strides(shape, rank, st);
  for (int d = 0; d < rank; d++) {
    g.out_shape[d] = shape[perm[d]];
    g.in_stride[d] = st[perm[d]];
    g.add[d] = 0;
    g.mul[d] = 1;
    g.in_shape[d] = shape[perm[d]];
    if (out_shape) out_shape[d] = g.out_shape[d];
  }
  return launch_gather(ctx, x, out, g, 0.0f, stream, "transpose_f32");
}

PH_EXPORT int piper_hip_expand_f32(piper_hip_ctx* ctx, const float* x, const int64_t* in_shape, const int64_t* out_shape,
                                   int rank, float** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (rank < 1 || rank > 4) PH_FAIL(PIPER_HIP_ERR_SHAPE, "expand: rank %d not supported (1..4)", rank);
  if (validate_shape(in_shape, rank, "expand in") || validate_shape(out_shape, rank, "expand out")) return PIPER_HIP_ERR_SHAPE;
  Gather4 g{};
  g.rank = rank;
  int64_t st[4];
  row_major_strides(in_shape, rank, st);
  for (int d = 0; d < rank; d++) {
    if (in_shape[d] != out_shape[d] && in_shape[d] != 1) PH_FAIL(PIPER_HIP_ERR_SHAPE, "expand: dim %d %lld→%lld", d,
                                                                 (long long)in_shape[d], (long long)out_shape[d]);
    g.out_shape[d] = out_shape[d];
    g.in_stride[d] = in_shape[d] == 1 ? 0 : st[d];
    g.add[d] = 0;
    g.mul[d] = 1;
    g.in_shape[d] = out_shape[d];
  }
  return launch_gather(ctx, x, out, g, 0.0f, stream, "expand_f32");
}

PH_EXPORT int piper_hip_concat2_axis1_f32(piper_hip_ctx* ctx, const float* a, const int64_t a_shape[3], const float* b,
                                          const int64_t b_shape[3], float** out, int64_t out_shape[3],
                                          piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (validate_shape(a_shape, 3, "concat a") || validate_shape(b_shape, 3, "concat b")) return PIPER_HIP_ERR_SHAPE;
  if (a_shape[0] != b_shape[0] || a_shape[2] != b_shape[2]) PH_FAIL(PIPER_HIP_ERR_SHAPE, "concat2_axis1: N/L mismatch");
  const int64_t N = a_shape[0], Ca = a_shape[1], Cb = b_shape[1], L = a_shape[2];
  if (out_shape) { out_shape[0] = N; out_shape[1] = Ca + Cb; out_shape[2] = L; }
  const size_t n = (size_t)(N * (Ca + Cb) * L);
  int rc = ph::ensure_out(ctx, out, n, 0);
  if (rc) return rc;
  if (n == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  // strided device-to-device copies: [N] rows of Ca*L (then Cb*L) floats into pitch (Ca+Cb)*L
  const size_t pitch = (size_t)(Ca + Cb) * L * sizeof(float);
  if (Ca * L)
    PH_HIP(hipMemcpy2DAsync(*out, pitch, a, (size_t)Ca * L * sizeof(float), (size_t)Ca * L * sizeof(float), (size_t)N,
                            hipMemcpyDeviceToDevice, ss.s), PIPER_HIP_ERR_LAUNCH);
  if (Cb * L)
    PH_HIP(hipMemcpy2DAsync(*out + Ca * L, pitch, b, (size_t)Cb * L * sizeof(float), (size_t)Cb * L * sizeof(float), (size_t)N,
                            hipMemcpyDeviceToDevice, ss.s), PIPER_HIP_ERR_LAUNCH);
  return ss.finish("concat2_axis1_f32");
}

PH_EXPORT int piper_hip_split2_axis1_f32(piper_hip_ctx* ctx, const float* x, const int64_t x_shape[3], int64_t c0,
                                         float** out0, float** out1, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (validate_shape(x_shape, 3, "split")) return PIPER_HIP_ERR_SHAPE;
  const int64_t N = x_shape[0], C = x_shape[1], L = x_shape[2];
  if (c0 < 0 || c0 > C) PH_FAIL(PIPER_HIP_ERR_SHAPE, "split2_axis1: c0=%lld outside [0,%lld]", (long long)c0, (long long)C);
  const int64_t c1 = C - c0;
  int rc = ph::ensure_out(ctx, out0, (size_t)(N * c0 * L), 0);
  if (rc) return rc;
  rc = ph::ensure_out(ctx, out1, (size_t)(N * c1 * L), 0);
  if (rc) return rc;
  if (N * C * L == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  const size_t pitch = (size_t)C * L * sizeof(float);
  if (c0 * L)
    PH_HIP(hipMemcpy2DAsync(*out0, (size_t)c0 * L * sizeof(float), x, pitch, (size_t)c0 * L * sizeof(float), (size_t)N,
                            hipMemcpyDeviceToDevice, ss.s), PIPER_HIP_ERR_LAUNCH);
  if (c1 * L)
    PH_HIP(hipMemcpy2DAsync(*out1, (size_t)c1 * L * sizeof(float), x + c0 * L, pitch, (size_t)c1 * L * sizeof(float), (size_t)N,
                            hipMemcpyDeviceToDevice, ss.s), PIPER_HIP_ERR_LAUNCH);
  return ss.finish("split2_axis1_f32");
}

PH_EXPORT int piper_hip_reduce_mean_lastdim_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank,
                                                float** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (rank < 1 || validate_shape(shape, rank, "reduce_mean")) PH_FAIL(PIPER_HIP_ERR_SHAPE, "reduce_mean: bad shape");
  const int64_t cols = shape[rank - 1];
  if (cols <= 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "reduceMeanLastDimF32 requires non-empty last dim");
  const int64_t rows = (int64_t)shape_count(shape, rank - 1);
  int rc = ph::ensure_out(ctx, out, (size_t)rows, 0);
  if (rc) return rc;
  if (rows == 0) return PIPER_HIP_OK;
  ph::StreamScope ss(ctx, stream);
  const int grid = grid_for(rows, kBlock / 64, ctx->num_cus);
  hipLaunchKernelGGL(reduce_mean_lastdim_kernel, dim3(grid), dim3(kBlock), 0, ss.s, x, *out, rows, cols);
  return ss.finish("reduce_mean_lastdim_f32");
}

PH_EXPORT int piper_hip_softmax_lastdim_f32(piper_hip_ctx* ctx, const float* x, const int64_t* shape, int rank,
                                            float** out, piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (rank < 1 || validate_shape(shape, rank, "softmax")) PH_FAIL(PIPER_HIP_ERR_SHAPE, "softmax: bad shape");
  const int64_t cols = shape[rank - 1];
  if (cols <= 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "softmaxLastDimF32 requires non-empty last dim");
  const int64_t rows = (int64_t)shape_count(shape, rank - 1);
  int rc = ph::ensure_out(ctx, out, (size_t)(rows * cols), 0);
  if (rc) return rc;
  if (rows == 0) return PIPER_HIP_OK;
  if (!x) PH_FAIL(PIPER_HIP_ERR_ARG, "softmax: null input");
  ph::StreamScope ss(ctx, stream);
  if (cols <= 64 * kSoftmaxRegs) {
    const int grid = grid_for(rows, kBlock / 64, ctx->num_cus);
    hipLaunchKernelGGL(softmax_wave_kernel, dim3(grid), dim3(kBlock), 0, ss.s, x, *out, rows, cols);
  } else {
    const int grid = grid_for(rows, 1, ctx->num_cus);
    hipLaunchKernelGGL(softmax_block_kernel, dim3(grid), dim3(kBlock), 0, ss.s, x, *out, rows, cols);
  }
  return ss.finish("softmax_lastdim_f32");
}

PH_EXPORT int piper_hip_add_layernorm_f32(piper_hip_ctx* ctx, const float* x, const float* y, const float* gamma,
                                          const float* beta, int64_t n, int64_t c, int64_t t, float eps, float** out,
                                          piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (n < 0 || c <= 0 || t < 0) PH_FAIL(PIPER_HIP_ERR_SHAPE, "add_layernorm: bad shape");
  if (n > 65535) PH_FAIL(PIPER_HIP_ERR_SHAPE, "add_layernorm: batch too large");
  int rc = ph::ensure_out(ctx, out, (size_t)(n * c * t), 0);
  if (rc) return rc;
  if (n * t == 0) return PIPER_HIP_OK;
  if (!x || !gamma || !beta) PH_FAIL(PIPER_HIP_ERR_ARG, "add_layernorm: null input");
  ph::StreamScope ss(ctx, stream);
  if (c * t > 0x7fffffff) PH_FAIL(PIPER_HIP_ERR_SHAPE, "add_layernorm: tensor too large");
  const dim3 grid((unsigned)ph::ceil_div(t, kLnT), (unsigned)n);
  if (c <= kLnG * 12)
    hipLaunchKernelGGL(add_layernorm_kernel<12>, grid, dim3(kBlock), 0, ss.s, x, y, gamma, beta, *out, (int)c, (int)t, eps);
  else if (c <= kLnG * 32)
    hipLaunchKernelGGL(add_layernorm_kernel<32>, grid, dim3(kBlock), 0, ss.s, x, y, gamma, beta, *out, (int)c, (int)t, eps);
  else
    hipLaunchKernelGGL(add_layernorm_big_kernel, grid, dim3(kBlock), 0, ss.s, x, y, gamma, beta, *out, (int)c, (int)t, eps);
  return ss.finish("add_layernorm_f32");
}

PH_EXPORT int piper_hip_matmul_f32(piper_hip_ctx* ctx, const float* a, const int64_t* a_shape, const float* b,
                                   const int64_t* b_shape, int rank, float** out, int64_t* out_shape,
                                   piper_hip_stream stream) {
  PH_CHECK_CTX(ctx);
  if (rank < 2) PH_FAIL(PIPER_HIP_ERR_SHAPE, "matmulF32 requires rank>=2 (got %d)", rank);
  if (rank > 4) PH_FAIL(PIPER_HIP_ERR_SHAPE, "matmulF32 rank %d not supported (2..4)", rank);
  if (validate_shape(a_shape, rank, "matmul a") || validate_shape(b_shape, rank, "matmul b")) return PIPER_HIP_ERR_SHAPE;
  const int64_t M = a_shape[rank - 2], K = a_shape[rank - 1], N = b_shape[rank - 1];
  if (b_shape[rank - 2] != K)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "matmulF32 inner dim mismatch: a[..,%lld] vs b[%lld,..]", (long long)K, (long long)b_shape[rank - 2]);
  // lead dims: equal, or 1 on either side (GraphExecutor.swift:1876-1887)
  int64_t lead[2] = {1, 1}, a_ls[2] = {0, 0}, b_ls[2] = {0, 0};
  const int nl = rank - 2;
  int64_t batch = 1;
  {
    int64_t ast = M * K, bst = K * N;
    for (int i = nl - 1; i >= 0; i--) {
      const int64_t da = a_shape[i], db = b_shape[i];
      if (da != db && da != 1 && db != 1)
        PH_FAIL(PIPER_HIP_ERR_SHAPE, "matmulF32 lead dims broadcast not supported (dim %d: %lld vs %lld)", i, (long long)da, (long long)db);
      lead[i] = da > db ? da : db;
      if (da == 0 || db == 0) lead[i] = 0;
      a_ls[i] = da == 1 ? 0 : ast;
      b_ls[i] = db == 1 ? 0 : bst;
      ast *= da;
      bst *= db;
      batch *= lead[i];
      if (out_shape) out_shape[i] = lead[i];
    }
  }
  if (out_shape) { out_shape[rank - 2] = M; out_shape[rank - 1] = N; }
  const size_t n = (size_t)(batch * M * N);
  int rc = ph::ensure_out(ctx, out, n, 0);
  if (rc) return rc;
  if (n == 0) return PIPER_HIP_OK;
  if (!a || !b) PH_FAIL(PIPER_HIP_ERR_ARG, "matmul: null input");
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) PH_FAIL(PIPER_HIP_ERR_SHAPE, "matmul: dims exceed int32");
  ph::StreamScope ss(ctx, stream);
  // Up to two lead dims: launch per outer lead index so that each launch has a single (possibly zero) batch stride.
  const int64_t l0 = nl >= 2 ? lead[0] : 1, l1 = nl >= 1 ? lead[nl - 1] : 1;
  const int64_t a_s1 = nl >= 1 ? a_ls[nl - 1] : 0, b_s1 = nl >= 1 ? b_ls[nl - 1] : 0;
  const int64_t a_s0 = nl >= 2 ? a_ls[0] : 0, b_s0 = nl >= 2 ? b_ls[0] : 0;
  if (l1 > 65535) PH_FAIL(PIPER_HIP_ERR_SHAPE, "matmul: batch dim too large");
  for (int64_t i0 = 0; i0 < l0; i0++) {
    dim3 grid((unsigned)ph::ceil_div(N, kMmT), (unsigned)ph::ceil_div(M, kMmT), (unsigned)l1);
    hipLaunchKernelGGL(matmul_kernel, grid, dim3(kBlock), 0, ss.s, a + i0 * a_s0, b + i0 * b_s0,
                       *out + i0 * l1 * M * N, (int)M, (int)N, (int)K, a_s1, b_s1);
  }
  return ss.finish("matmul_f32");
}
namespace { PH_WARM(ops_basic, gather4_kernel); }
