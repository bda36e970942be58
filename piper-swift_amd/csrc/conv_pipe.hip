// conv_pipe.hip — fp32 Conv1d / ConvTranspose1d for long rows as a PERSISTENT, software-pipelined implicit GEMM
// (exact fp32, v_mfma_f32_32x32x2_f32).
//
// Role: the ResBlock convs and ConvTranspose upsamplers of the HiFi-GAN generator (conv1d.metal:28-71, 97-142 in the
// reference). Round 1's conv_win_kernel gave every block ONE tile: stage the whole window → barrier → K loop → epilogue.
// With ~4 blocks per CU all resident at once, the whole grid ran those three phases in lockstep — every CU loading, then
// every CU on the matrix pipe, then every CU storing — so the pipe sat at 26 % busy with 56 % of wave time waiting
// (profiles/r1c_mfma_busy.md). Here
//   * a block is persistent: it walks tiles t = blockIdx.x, + gridDim.x, … (tile = conv of the launch × batch item × row
//     group × column block, convs interleaved so that every block gets a mix of kernel sizes);
//   * the contraction is cut into chunks of 32 input channels: the window of chunk c+1 (of this tile or of the block's next
//     tile) is in flight global → registers while chunk c feeds the matrix pipe from LDS, and is written to the other LDS
//     buffer behind the MFMAs — one barrier per chunk, loads never exposed except for the very first chunk of a block;
//   * the weight-fragment ring (float4 = 4 steps per lane, 8 deep) runs across tile boundaries (its last groups already
//     fetch the next tile's first fragments), and the residual of the epilogue is fetched before the tile's last chunk;
//   * LDS per block is 2 × 32 channels × window ≤ 64 KiB whatever Cin is (the one-shot window of 256/512 channels did not
//     fit), so two blocks share a CU and one block's epilogue hides under the other's MFMAs.
// Geometry (wave arrangement WM × WN, NTW column tiles per wave) is chosen by the host from the channel count.
#include <algorithm>
#include <type_traits>

#include "conv_win.h"

// PH_PIPE_ABL (tools/probe/pipeprobe): ablation bits for timing experiments — results are WRONG when set. 1: no weight-ring
// loads in the loop; 2: no window staging after the prologue; 4: no epilogue stores; 8: no residual loads; 16: no LDS reads.
#ifndef PH_PIPE_ABL
#define PH_PIPE_ABL 0
#endif

namespace ph {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBT = 256;
constexpr int kRA = 8;        // float4 weight groups in flight (32 steps)
constexpr int kCP = 16;       // channel pairs per chunk (32 channels)
constexpr int kCh = 2 * kCP;  // channels per chunk
constexpr int kSt = 8;        // staging float4 slots per thread: 32 rows × Wp/4 ≤ 256·kSt  ⇒  Wp ≤ 256
constexpr int kMaxWp = 256;
constexpr int kTailFloats = 4096;  // readable floats behind the image: the ring runs up to kRA groups past the last tile

__device__ __forceinline__ float lrelu1(float v, float alpha) { return v >= 0.0f ? v : v * alpha; }

// step s of row tile mt: chunk = s / (taps·16), tap = (s / 16) % taps, cp = s % 16; lane l holds
//   conv : w[mt·32 + (l&31)][2·(chunk·16 + cp) + (l>>5)][tap]
//   convT: w[ci = 2·(chunk·16 + cp) + (l>>5)][co][(ρ+pad) mod s + s·tap],  row R = mt·32 + (l&31) = ρ·Cout + co
// stored as float4 groups of 4 consecutive steps per lane: [mt][s/4][lane][4].
__global__ __launch_bounds__(kBT) void pack_pipe_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int ct_stride, int ct_pad,
                                                       int taps, int S, float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBT) {
    const int e = (int)(i & 3), l = (int)((i >> 2) & 63);
    const int64_t g = i >> 8;  // (mt, s4)
    const int s4 = (int)(g % (S >> 2)), mt = (int)(g / (S >> 2));
    const int step = 4 * s4 + e;
    const int cp = step & 15, tap = (step >> 4) % taps, chunk = (step >> 4) / taps;
    const int ci = 2 * (chunk * kCP + cp) + (l >> 5);
    const int R = mt * 32 + (l & 31);
    float v = 0.0f;
    if (ci < Cin) {
      if (ct_stride > 0) {
        const int rho = R / Cout, co = R - rho * Cout;
        if (rho < ct_stride) v = w[((int64_t)ci * Cout + co) * K + (rho + ct_pad) % ct_stride + ct_stride * tap];
      } else if (R < Cout) {
        v = w[((int64_t)R * Cin + ci) * K + tap];
      }
    }
    out[i] = v;
  }
}

struct PipeMulti {
  ConvWinArgs c[kWinMulti];
};

struct TileInfo {  // wave-uniform description of one tile (all scalars)
  int j, n, rg, cb;
};

template <int WM, int WN, int NTW, bool AVG>
__global__ __launch_bounds__(kBT, 2) void conv_pipe_kernel(const PipeMulti multi, const int count, const int batch, const int ntiles,
                                                       const int col_blocks, const int row_groups, const int Wp, const int order) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int NBC = WN * NTW * 32;  // columns per tile
  constexpr int NX = AVG ? 3 : 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 × [32][Wp]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 31, h = lane >> 5;
  const int W4 = Wp >> 2;
  const int buf_floats = kCh * Wp;

  // ---- staging slots of this thread: element e = tid + 256·i of the chunk's [32][W4] float4 grid. (row, c4) of slot 0 and
  // the per-slot step (256 / W4 rows, 256 % W4 columns) are kept; the slots are walked incrementally (no divisions, 2 VGPRs).
  const int st_row0 = tid / W4, st_c40 = tid - st_row0 * W4;
  const int st_dr = kBT / W4, st_dc = kBT - st_dr * W4;

  // Tile order. The launch's convs differ in taps (a ResBlock stage: kernels 3 / 7 / 11), i.e. in cost per tile. Tiles are ranked
  // heaviest conv first; a block's sequence number e = blockIdx.x + m·gridDim.x is cut into HALF-rounds of gridDim.x / 2: even
  // half-rounds take from the heavy end of the ranking, odd ones from the light end. Blocks b and b + gridDim.x / 2 are the two
  // a CU holds, so each CU gets a heavy and a light tile instead of two of a kind (r3: the 256-channel stage of the high voice
  // at factor 8 is 504 tiles for 512 block slots — one tile per block, 121 µs with k = 7 and k = 11 sharing CUs).
  const int tpc = col_blocks * row_groups * batch;  // tiles per conv
  const int half_g = max((int)gridDim.x >> 1, 1);
  auto decode = [&](int e) {
    const int hr = e / half_g, b = e - hr * half_g;
    int i = (hr & 1) ? ntiles - 1 - (hr >> 1) * half_g - b : (hr >> 1) * half_g + b;  // rank, heaviest first
    i = __builtin_amdgcn_readfirstlane(min(max(i, 0), ntiles - 1));
    const int z = i / tpc;  // position in the cost order
    int rest = i - z * tpc;
    TileInfo ti;
    ti.j = (order >> (4 * z)) & 15;
    ti.cb = rest % col_blocks;
    rest /= col_blocks;
    ti.rg = rest % row_groups;
    ti.n = rest / row_groups;
    return ti;
  };
  // first input position of a tile's window (before alignment): conv nb0 − padL, convT nb0 − (taps−1)
  auto win_start = [&](const ConvWinArgs& p, int cb) {
    const int taps = p.ct_stride > 0 ? p.K / p.ct_stride : p.K;
    return cb * NBC + (p.ct_stride > 0 ? -(taps - 1) : -p.padL);
  };

  float4 stg[NX][kSt];
  // global → registers for chunk `ch` of tile `ti` (unconditional loads from clamped addresses; masked when written to LDS)
  auto issue = [&](const TileInfo& ti, int ch) {
    const ConvWinArgs& p = multi.c[ti.j];
    const int ga = win_start(p, ti.cb) & ~3;
    const int64_t base = ((int64_t)ti.n * p.Cin + ch * kCh) * p.Lin;  // wave-uniform
    const float* xb = p.x + base;
    const float* xb2 = AVG ? p.x2 + base : nullptr;
    const float* xb3 = AVG ? p.x3 + base : nullptr;
    int srow = st_row0, sc4 = st_c40;
#pragma unroll
    for (int i = 0; i < kSt; i++) {
      const int row = min(srow, kCh - 1);
      const int pos = min(max(ga + 4 * sc4, 0), p.Lin - 4);
      const int off = row * p.Lin + pos;  // 32 rows of one batch item: fits 32 bits (host checks 32·Lin < 2^31)
      stg[0][i] = *(const float4*)(xb + off);
      if constexpr (AVG) {
        stg[1][i] = *(const float4*)(xb2 + off);
        stg[2][i] = *(const float4*)(xb3 + off);
      }
      sc4 += st_dc; srow += st_dr;
      const bool wrapc = sc4 >= W4;
      sc4 -= wrapc ? W4 : 0; srow += wrapc ? 1 : 0;
    }
  };
  // registers → LDS buffer `buf`: MRF mean, LeakyReLU and the zero padding (positions outside [0, Lv)) applied on the way in
  auto commit = [&](const TileInfo& ti, float* buf) {
    const ConvWinArgs& p = multi.c[ti.j];
    const int ga = win_start(p, ti.cb) & ~3;
    const int Lv = p.len_ptr ? min(p.len_ptr[ti.n] * p.len_mul, p.Lin) : p.Lin;  // true input length of this batch item
    int srow = st_row0, sc4 = st_c40;
#pragma unroll
    for (int i = 0; i < kSt; i++) {
      float4 v = stg[0][i];
      if constexpr (AVG) {  // ((x + x2) + x3) / 3: the association of the graph's Add, Add, Div
        v.x = ((v.x + stg[1][i].x) + stg[2][i].x) / 3.0f; v.y = ((v.y + stg[1][i].y) + stg[2][i].y) / 3.0f;
        v.z = ((v.z + stg[1][i].z) + stg[2][i].z) / 3.0f; v.w = ((v.w + stg[1][i].w) + stg[2][i].w) / 3.0f;
      }
      v.x = lrelu1(v.x, p.pro_alpha); v.y = lrelu1(v.y, p.pro_alpha); v.z = lrelu1(v.z, p.pro_alpha); v.w = lrelu1(v.w, p.pro_alpha);
      const int pos = ga + 4 * sc4;  // multiple of 4
      const int nv = pos < 0 ? 0 : Lv - pos;  // valid leading components (≥ 4: all)
      v.x = nv > 0 ? v.x : 0.0f; v.y = nv > 1 ? v.y : 0.0f; v.z = nv > 2 ? v.z : 0.0f; v.w = nv > 3 ? v.w : 0.0f;
      if (srow < kCh) *(float4*)(buf + srow * Wp + 4 * sc4) = v;
      sc4 += st_dc; srow += st_dr;
      const bool wrapc = sc4 >= W4;
      sc4 -= wrapc ? W4 : 0; srow += wrapc ? 1 : 0;
    }
  };

  // Bucketed / ragged batches: a tile whose columns all lie at or past its item's TRUE length computes nothing anyone reads;
  // the block steps over such tiles (its tile sequence is t, t + G, …: next_live keeps that stride), so the pipeline below only
  // ever sees live tiles.
  auto next_live = [&](int tt) {
    while (tt < ntiles) {
      const TileInfo ti = decode(tt);
      const ConvWinArgs& p = multi.c[ti.j];
      if (!p.len_ptr || ti.cb * NBC < min(p.len_ptr[ti.n] * p.len_mul, p.Lin)) break;
      tt += gridDim.x;
    }
    return tt;
  };
  int t = next_live(blockIdx.x);
  if (t >= ntiles) return;
  TileInfo cur = decode(t);

  // ---- weight ring: uniform base + 32-bit lane offset; the groups past the end of a tile's stream come from the next tile
  const unsigned lane16 = (unsigned)lane * 16u;
  float4 a[kRA];
  auto tile_wbase = [&](const TileInfo& ti) {
    const ConvWinArgs& p = multi.c[ti.j];
    const int taps = p.ct_stride > 0 ? p.K / p.ct_stride : p.K;
    const int S = (p.Cin / kCh) * taps * kCP;
    const int mtc = ((p.ct_stride > 0 ? p.Cout * p.ct_stride : p.Cout) + 31) >> 5;  // row tiles of the image
    return (const char*)p.w4 + (int64_t)min(ti.rg * WM + wm, mtc - 1) * S * 256;  // a row group's surplus waves re-read the last tile
  };
  auto tile_groups = [&](const TileInfo& ti) {
    const ConvWinArgs& p = multi.c[ti.j];
    const int taps = p.ct_stride > 0 ? p.K / p.ct_stride : p.K;
    return (p.Cin / kCh) * taps * kCP / 4;
  };
  const char* wa_cur = tile_wbase(cur);
  const char* wa_next = wa_cur;
  int G = tile_groups(cur);  // float4 groups of the current tile
  int gpos = 0;              // groups of the current tile consumed so far
  int phase = 0;             // ring slot of the next group = groups consumed since kernel start mod 8 (0 or 4: chunks hold 4·taps)
  auto load_a = [&](int slot, int ahead) {
#if PH_PIPE_ABL & 1
    if (gpos > 0) return;
#endif
    const int g = gpos + ahead;
    const char* src = g < G ? wa_cur + (int64_t)g * 1024 : wa_next + (int64_t)(g - G) * 1024;
    a[slot] = *(const float4*)(src + lane16);
  };

  // ---- prologue: first chunk of the first tile
  issue(cur, 0);
  {
    const int tn = next_live(t + gridDim.x);
    if (tn < ntiles) wa_next = tile_wbase(decode(tn));
  }
#pragma unroll
  for (int d = 0; d < kRA - 1; d++) load_a(d, d);
  commit(cur, lds);
  __syncthreads();

  // accumulators start at the bias (bias first, like CPUBackend.conv1d); a tile's bias is loaded while the previous tile's
  // stores drain
  f32x16 acc[NTW];
  auto seed_acc = [&](const TileInfo& ti) {
    const ConvWinArgs& p = multi.c[ti.j];
    const bool ct = p.ct_stride > 0;
    const int mt = ti.rg * WM + wm;
    const int rho = ct ? min((mt * 32) / p.Cout, p.ct_stride - 1) : 0;
    const int co0 = ct ? mt * 32 - rho * p.Cout : mt * 32;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int coc = max(min(co0 + (q & 3) + 8 * (q >> 2) + 4 * h, p.Cout - 1), 0);
      const float bv = p.bias ? p.bias[coc] : 0.0f;
#pragma unroll
      for (int j = 0; j < NTW; j++) acc[j][q] = bv;
    }
  };
  seed_acc(cur);

  int bufsel = 0;
  while (true) {
    const ConvWinArgs& p = multi.c[cur.j];
    const bool ct = p.ct_stride > 0;
    const int taps = ct ? p.K / p.ct_stride : p.K;
    const int nch = p.Cin / kCh;
    const int nb0 = cur.cb * NBC;
    const int mt = cur.rg * WM + wm;
    const int off_min = ct ? -(taps - 1) : -p.padL;
    const int shift = (nb0 + off_min) - ((nb0 + off_min) & ~3);
    const int rho = ct ? min((mt * 32) / p.Cout, p.ct_stride - 1) : 0;
    const int off0 = ct ? (rho + p.ct_pad) / p.ct_stride : -p.padL;  // window position of tap t: off0 + t·dstep
    const int dstep = ct ? -1 : p.dil;
    const int lbase = h * Wp + shift + wn * NTW * 32 + r - off_min + off0;
    const int tn = next_live(t + gridDim.x);
    const bool has_next = tn < ntiles;
    const TileInfo nxt = has_next ? decode(tn) : cur;

    // epilogue addresses of this tile (needed early: the residual is fetched before the last chunk)
    const int rows_total = ct ? p.Cout * p.ct_stride : p.Cout;
    const int row0 = mt * 32;
    const int co0 = ct ? row0 - rho * p.Cout : row0;
    float resv[NTW][16];

    for (int ch = 0; ch < nch; ch++) {
      const bool last = ch + 1 == nch;
      // (1) next stage's window: global → registers, in flight during this chunk's MFMAs
#if !(PH_PIPE_ABL & 2)
      if (!last) issue(cur, ch + 1);
      else if (has_next) issue(nxt, 0);
#endif
      // (2) the residual of the epilogue, issued ahead of the last chunk
      if (last && p.res && !(PH_PIPE_ABL & 8)) {
#pragma unroll
        for (int j = 0; j < NTW; j++) {
          const int col = min(nb0 + (wn * NTW + j) * 32 + r, p.Lout - 1);
          const int pos = ct ? col * p.ct_stride + rho : col;
          const float* rb = p.res + (int64_t)cur.n * p.Cout * p.y_len;
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const int coc = min(co0 + (q & 3) + 8 * (q >> 2) + 4 * h, p.Cout - 1);
            resv[j][q] = rb[coc * p.y_len + pos];
          }
        }
      }
      // (3) this chunk: taps·16 steps per column tile out of LDS buffer `bufsel`
      {
        const float* win = lds + bufsel * buf_floats;
        int sidx = 0, c_n = 0, left = taps * kCP - 1;  // the index stops at the chunk's last step (the ring reads one group ahead)
        const int wrap_delta = dstep - 2 * Wp * (kCP - 1);
        float b[2][NTW][4];
        auto read_b4 = [&](int slot) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int j = 0; j < NTW; j++) {
#if PH_PIPE_ABL & 16
              b[slot][j][e] = (float)(sidx + e);
#else
              b[slot][j][e] = win[lbase + sidx + 32 * j];
#endif
            }
            c_n++;
            const bool wrap = c_n == kCP;
            c_n = wrap ? 0 : c_n;
            const int delta = wrap ? wrap_delta : 2 * Wp;
            sidx += left > 0 ? delta : 0;
            left--;
          }
        };
        auto group = [&](int u) {  // u = static ring slot
          load_a((u + kRA - 1) % kRA, kRA - 1);
          gpos++;
          // keep the load HERE: left alone, the scheduler clusters the ring's loads next to their first use
          __builtin_amdgcn_sched_barrier(0);
          read_b4((u + 1) & 1);
#pragma unroll
          for (int j = 0; j < NTW; j++) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u & 1][j][0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u & 1][j][1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u & 1][j][2], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u & 1][j][3], acc[j], 0, 0, 0);
          }
        };
        read_b4(0);
        const int ngr = taps * (kCP / 4);  // groups in this chunk: a multiple of 4; the ring has 8 slots
        // ring slot of the chunk's first group: 0 or 4 (a chunk holds 4·taps groups, so odd tap counts alternate)
        if (phase == 0) {
          for (int g = 0; g + kRA <= ngr; g += kRA) {
#pragma unroll
            for (int u = 0; u < kRA; u++) group(u);
          }
          if (ngr & 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) group(u);
          }
        } else {
#pragma unroll
          for (int u = 4; u < 8; u++) group(u);
          for (int g = 4; g + kRA <= ngr; g += kRA) {
#pragma unroll
            for (int u = 0; u < kRA; u++) group(u);
          }
          if (!(ngr & 4)) {
#pragma unroll
            for (int u = 0; u < 4; u++) group(u);
          }
        }
        phase = (phase + ngr) & (kRA - 1);
      }
      // (4) next stage's window: registers → the other LDS buffer (its last readers passed the previous barrier)
#if !(PH_PIPE_ABL & 2)
      if (!last) commit(cur, lds + (bufsel ^ 1) * buf_floats);
      else if (has_next) commit(nxt, lds + (bufsel ^ 1) * buf_floats);
#endif
      __syncthreads();
      bufsel ^= 1;
    }

    // ---- epilogue. register q of lane (r,h): row (q&3) + 8·(q>>2) + 4·h, column r. Stores masked, loads were clamped.
    if (row0 < rows_total) {
#pragma unroll
      for (int j = 0; j < NTW; j++) {
        const int col = nb0 + (wn * NTW + j) * 32 + r;
        const bool okc = col < p.Lout;
        const int colc = min(col, p.Lout - 1);
        const int pos = ct ? colc * p.ct_stride + rho : colc;
        const int64_t ybase = (int64_t)cur.n * p.Cout * p.y_len;  // one batch item's tensor fits 32-bit offsets (host check)
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = acc[j][q];
        if (p.res && !(PH_PIPE_ABL & 8)) {
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] += resv[j][q];
        }
        if (p.mrf_a) {
          float ta[16], tb[16];
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const int coc = min(co0 + (q & 3) + 8 * (q >> 2) + 4 * h, p.Cout - 1);
            ta[q] = (p.mrf_a + ybase)[coc * p.y_len + pos];
            tb[q] = (p.mrf_b + ybase)[coc * p.y_len + pos];
          }
#pragma unroll
          for (int q = 0; q < 16; q++) v[q] = ((ta[q] + tb[q]) + v[q]) / 3.0f;
        }
        float* yb = p.y + ybase;
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int co = co0 + (q & 3) + 8 * (q >> 2) + 4 * h;
#if PH_PIPE_ABL & 4
          if (okc && co < p.Cout && v[q] == 1.2345e33f) yb[co * p.y_len + pos] = v[q];
#else
          if (okc && co < p.Cout) yb[co * p.y_len + pos] = lrelu1(v[q], p.out_alpha);
#endif
        }
      }
    }
    if (has_next) seed_acc(nxt);

    if (!has_next) break;
    // ---- advance to the block's next tile: its weight stream is the one the ring has been prefetching from
    t = tn;
    cur = nxt;
    gpos -= G;
    wa_cur = wa_next;
    G = tile_groups(cur);
    {
      const int tnn = next_live(t + gridDim.x);
      wa_next = tnn < ntiles ? tile_wbase(decode(tnn)) : wa_cur;
    }
  }
}

struct PipeGeom {
  int WM, WN, NTW;
};

}  // namespace

size_t packed_conv_pipe_floats(int Cout, int Cin, int K) { return (size_t)((Cout + 31) / 32) * (Cin / kCh) * K * kCP * 64 + kTailFloats; }
size_t packed_convt_pipe_floats(int Cin, int Cout, int K, int stride) {
  return (size_t)((Cout * stride + 31) / 32) * (Cin / kCh) * (K / stride) * kCP * 64 + kTailFloats;
}

int pack_conv_weights_pipe(hipStream_t s, const float* w, int Cout, int Cin, int K, float* packed) {
  const int64_t total = (int64_t)packed_conv_pipe_floats(Cout, Cin, K);
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBT), 4096);
  hipLaunchKernelGGL(pack_pipe_kernel, dim3(grid), dim3(kBT), 0, s, w, Cout, Cin, K, 0, 0, K, (Cin / kCh) * K * kCP, packed, total);
  return PIPER_HIP_OK;
}
int pack_convt_weights_pipe(hipStream_t s, const float* w, int Cin, int Cout, int K, int stride, int pad, float* packed) {
  const int64_t total = (int64_t)packed_convt_pipe_floats(Cin, Cout, K, stride);
  const int grid = (int)std::min<int64_t>(ceil_div(total, kBT), 4096);
  const int J = K / stride;
  hipLaunchKernelGGL(pack_pipe_kernel, dim3(grid), dim3(kBT), 0, s, w, Cout, Cin, K, stride, pad, J, (Cin / kCh) * J * kCP, packed, total);
  return PIPER_HIP_OK;
}

// window reach of one conv: columns the window extends beyond the tile's own
static int pipe_reach(const ConvWinArgs& a) {
  if (a.ct_stride > 0) return (a.ct_stride - 1 + a.ct_pad) / a.ct_stride + a.K / a.ct_stride - 1;
  return (a.K - 1) * a.dil;
}

bool conv_pipe_eligible(int Cout, int Cin, int K, int dil, int padL, int Lin, int Lout) {
  if (Cout < 1 || Cin < kCh || (Cin % kCh) || K < 1 || dil < 1 || padL < 0 || Lin < 4 || (Lin & 3) || Lout < 1) return false;
  if (K * (Cin / kCh) < 2) return false;  // the weight ring looks at most one tile ahead: a tile needs ≥ 8 float4 groups
  return 32 + (K - 1) * dil + 6 <= kMaxWp;
}
bool convt_pipe_eligible(int Cin, int Cout, int K, int stride, int pad, int Lin) {
  if (Cin < kCh || (Cin % kCh) || Cout < 32 || (Cout & 31) || stride < 1 || Lin < 4 || (Lin & 3)) return false;
  if ((K / stride) * (Cin / kCh) < 2) return false;
  return K % stride == 0 && K - stride == 2 * pad;
}

int launch_conv_pipe_multi(piper_hip_ctx* ctx, hipStream_t s, const ConvWinArgs* convs, int count) {
  if (count < 1 || count > kWinMulti) PH_FAIL(PIPER_HIP_ERR_ARG, "conv_pipe: %d convs in one launch (1..%d)", count, kWinMulti);
  const ConvWinArgs& a = convs[0];
  if (a.N <= 0 || a.Lout <= 0) return PIPER_HIP_OK;
  const bool ct = a.ct_stride > 0;
  const bool avg = a.x2 != nullptr;
  int reach = 0;
  for (int i = 0; i < count; i++) {
    const ConvWinArgs& b = convs[i];
    if (b.N != a.N || b.Cin != a.Cin || b.Cout != a.Cout || b.Lin != a.Lin || b.Lout != a.Lout || b.ct_stride != a.ct_stride ||
        (b.x2 != nullptr) != avg)
      PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv_pipe: convs of one launch must share N, Cin, Cout, Lin, Lout and kind");
    if (b.Cin % kCh || (b.Lin & 3) || (ct && (b.K % b.ct_stride)) || (ct ? b.K / b.ct_stride : b.K) * (b.Cin / kCh) < 2 || b.Lin < 4)
      PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_pipe: geometry not covered");
    reach = std::max(reach, pipe_reach(b));
  }
  const int rows = ct ? a.Cout * a.ct_stride : a.Cout;
  const int MT = (rows + 31) / 32;
  const int per_phase = ct ? a.Cout / 32 : MT;  // a wave's row tile must lie inside one ConvTranspose phase: always true (32 | Cout)
  (void)per_phase;
  // wave arrangement: as many waves along the rows as there are row tiles (they share the staged window), the rest along the
  // columns; two column tiles per wave when that keeps ≥ 4 tiles per CU (halves the weight stream per MFMA)
  PipeGeom g;
  g.WM = MT >= 4 ? 4 : (MT >= 2 ? 2 : 1);
  g.WN = 4 / g.WM;
  g.NTW = 1;
  auto tiles_for = [&](int ntw) { return (int64_t)ceil_div(a.Lout, g.WN * ntw * 32) * ceil_div(MT, g.WM) * a.N * count; };
  // r2d probe: 128 channels × 21 504 columns 196 → 166 µs with two column tiles per wave (half the window halo per column,
  // half the weight stream per MFMA); it stops paying when a block no longer gets ≥ 2 tiles
  // (256 channels = two row groups: 115 → 120 µs, left at one)
  if (!avg && MT <= 4 && tiles_for(2) >= 2 * (int64_t)ctx->num_cus && g.WN * 2 * 32 + reach + 6 <= kMaxWp) g.NTW = 2;
  if ((int64_t)kCh * a.Lin > 0x7fffffff || (int64_t)a.Cout * a.y_len > 0x7fffffff)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv_pipe: a batch item's tensor exceeds 32-bit offsets");
  static const char* force = getenv("PIPER_HIP_PIPE_NTW");  // tuning hook
  if (force && !avg && atoi(force) >= 1 && atoi(force) <= 2 && g.WN * atoi(force) * 32 + reach + 6 <= kMaxWp) g.NTW = atoi(force);
  const int NBC = g.WN * g.NTW * 32;
  const int Wp = (NBC + reach + 3 + 3) & ~3;  // holds shift (≤ 3) + NBC + reach
  if (Wp > kMaxWp) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_pipe: window of %d columns exceeds %d", Wp, kMaxWp);
  const int col_blocks = (int)ceil_div(a.Lout, NBC), row_groups = (int)ceil_div(MT, g.WM);
  const int64_t ntiles64 = (int64_t)col_blocks * row_groups * a.N * count;
  if (ntiles64 > 0x7fffffff) PH_FAIL(PIPER_HIP_ERR_SHAPE, "conv_pipe: too many tiles");
  const int ntiles = (int)ntiles64;
  const size_t lds = (size_t)2 * kCh * Wp * sizeof(float);
  const int per_cu = 2;  // blocks per CU: ≤ 256 VGPRs per lane (launch bounds) and ≤ 64 KiB of LDS each
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)ctx->num_cus * per_cu);
  PipeMulti multi;
  for (int i = 0; i < kWinMulti; i++) multi.c[i] = convs[i < count ? i : 0];
  int ord[kWinMulti] = {0, 1, 2}, order = 0;  // convs by taps, heaviest first (tile cost ∝ taps)
  std::sort(ord, ord + count, [&](int l, int r2) {
    const int tl = convs[l].ct_stride > 0 ? convs[l].K / convs[l].ct_stride : convs[l].K, tr = convs[r2].ct_stride > 0 ? convs[r2].K / convs[r2].ct_stride : convs[r2].K;
    return tl > tr;
  });
  for (int i = 0; i < count; i++) order |= ord[i] << (4 * i);
#define PH_PIPE_CASE(M, N_, T_)                                                                                                        \
  if (g.WM == M && g.WN == N_ && g.NTW == T_) {                                                                                        \
    if (avg) {                                                                                                                         \
      if constexpr (T_ == 1) {                                                                                                         \
        static bool raised[kMaxDevices] = {};                                                                                          \
        if (lds > 64 * 1024 && lds_optin_needed(raised))                                                                               \
          (void)hipFuncSetAttribute((const void*)conv_pipe_kernel<M, N_, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((conv_pipe_kernel<M, N_, 1, true>), dim3(grid), dim3(kBT), lds, s, multi, count, a.N, ntiles, col_blocks, row_groups, Wp, order); \
      }                                                                                                                                \
    } else {                                                                                                                           \
      static bool raised[kMaxDevices] = {};                                                                                            \
      if (lds > 64 * 1024 && lds_optin_needed(raised))                                                                                 \
        (void)hipFuncSetAttribute((const void*)conv_pipe_kernel<M, N_, T_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      hipLaunchKernelGGL((conv_pipe_kernel<M, N_, T_, false>), dim3(grid), dim3(kBT), lds, s, multi, count, a.N, ntiles, col_blocks, row_groups, Wp, order); \
    }                                                                                                                                  \
  } else
  PH_PIPE_CASE(4, 1, 1) PH_PIPE_CASE(4, 1, 2) PH_PIPE_CASE(2, 2, 1) PH_PIPE_CASE(2, 2, 2) PH_PIPE_CASE(1, 4, 1) PH_PIPE_CASE(1, 4, 2)
  PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "conv_pipe: no instance for WM=%d WN=%d NTW=%d", g.WM, g.WN, g.NTW);
#undef PH_PIPE_CASE
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) PH_FAIL(PIPER_HIP_ERR_LAUNCH, "conv_pipe launch failed: %s", hipGetErrorString(e));
  return PIPER_HIP_OK;
}

}  // namespace ph
namespace ph { namespace { PH_WARM(conv_pipe, pack_pipe_kernel); } }
