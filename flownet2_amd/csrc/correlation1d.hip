// Correlation1D (horizontal cost volume) for gfx950.
//
// Replaces Correlation1DLayer::Forward_gpu / Backward_gpu (reference: src/caffe/layers/correlation_layer1d.cu:429-616)
// behind fn2_correlation1d_{forward,backward}.  As for the 2-D layer there is no padded NHWC scratch copy: the kernels read
// the NCHW blobs and treat the horizontal padding as zeros.  The reference addresses its scratch blob [N, H, W+2p, C] with a
// flat index and no bounds check; with single_direction = -1 the first displacement lies one grid step beyond the radius
// (x_shift = -grid_width, correlation_layer1d.cu:466-471), so columns left of a padded row are read: that is the end of the
// previous row of the flat blob.  padded_flat() reproduces exactly that (memory in front of the blob reads as 0).
//
// Generic kernels: one thread per output element, x fastest (coalesced along rows); gather formulation of the backward passes, no
// atomics.  The forward of the configuration the networks use (kernel_size 1, stride_1 1, MULTIPLY) has an LDS-tiled kernel.
#include "fn2_common.hpp"

#include <cmath>

namespace fn2 {

struct Corr1dGeom {
  int N, C, H, W;
  int pad, K, md, s1, s2, kr, pW;
  int topC, topH, topW, ngr, ngw, xshift;
  int type;
};

// Correlation1DLayer::LayerSetUp + Reshape, correlation_layer1d.cpp:12-92.
static int corr1d_geometry(const fn2_corr_params* p, int N, int C, int H, int W, Corr1dGeom* g) {
  if (!p) return fail(FN2_ERR_INVALID_ARG, "correlation1d: params == NULL");
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "correlation1d: bad bottom shape [%d,%d,%d,%d]", N, C, H, W);
  if (p->kernel_size < 1 || p->kernel_size % 2 == 0)
    return fail(FN2_ERR_INVALID_ARG, "correlation1d: Odd kernel size required (got %d)", p->kernel_size);
  if (p->stride1 < 1 || p->stride2 < 1) return fail(FN2_ERR_INVALID_ARG, "correlation1d: strides must be >= 1");
  if (p->max_displacement < 0 || p->pad < 0) return fail(FN2_ERR_INVALID_ARG, "correlation1d: negative pad / max_displacement");
  if (p->single_direction < -1 || p->single_direction > 1)
    return fail(FN2_ERR_INVALID_ARG, "correlation1d: single_direction must be -1 (left), 0 (off), or 1 (right)");
  if (p->corr_type != FN2_CORR_MULTIPLY && p->corr_type != FN2_CORR_SUBTRACT)
    return fail(FN2_ERR_INVALID_ARG, "correlation1d: unknown correlation_type %d", p->corr_type);
  g->N = N; g->C = C; g->H = H; g->W = W;
  g->pad = p->pad; g->K = p->kernel_size; g->md = p->max_displacement; g->s1 = p->stride1; g->s2 = p->stride2;
  g->type = p->corr_type;
  g->kr = (g->K - 1) / 2;
  g->pW = W + 2 * g->pad;
  const int border = g->md + g->kr;
  g->topW = (int)std::ceil((float)(g->pW - border * 2) / (float)g->s1);
  g->topH = (int)std::ceil((float)(H - g->kr * 2) / (float)g->s1);
  if (g->topW < 1 || g->topH < 1)
    return fail(FN2_ERR_INVALID_ARG, "Correlation cannot be done with current settings. Neighborhood and kernel don't fit in blob");
  g->ngr = g->md / g->s2;
  g->ngw = p->single_direction != 0 ? g->ngr + 1 : 2 * g->ngr + 1;
  g->topC = g->ngw;
  g->xshift = p->single_direction == -1 ? -g->ngw : (p->single_direction == 1 ? 0 : -g->ngr);   // correlation_layer1d.cu:466-471
  return FN2_OK;
}

// Element (n, m, q, c) of the reference's flat zero-padded scratch blob [N, H, pW, C], q possibly outside [0, pW).
__device__ __forceinline__ float padded_flat(const float* __restrict__ b, int n, int c, int m, int q, const Corr1dGeom& g) {
  if (q < 0 || q >= g.pW) {
    const long long f = ((long long)n * g.H + m) * g.pW + q;
    if (f < 0 || f >= (long long)g.N * g.H * g.pW) return 0.f;   // outside the blob: undefined in the reference
    const long long row = f / g.pW;
    q = (int)(f - row * g.pW);
    n = (int)(row / g.H);
    m = (int)(row - (long long)n * g.H);
  }
  const int x = q - g.pad;
  return (x >= 0 && x < g.W) ? b[(((size_t)n * g.C + c) * g.H + m) * g.W + x] : 0.f;
}

template <bool SUB>
__global__ void __launch_bounds__(256) corr1d_fwd(const float* __restrict__ b0, const float* __restrict__ b1,
                                                  float* __restrict__ top, Corr1dGeom g) {
  const long long total = (long long)g.N * g.topC * g.topH * g.topW;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % g.topW);
    const int y = (int)((idx / g.topW) % g.topH);
    const int tc = (int)((idx / g.topW / g.topH) % g.topC);
    const int n = (int)(idx / g.topW / g.topH / g.topC);
    const int s2o = (tc + g.xshift) * g.s2;
    float sum = 0.f;
    for (int j = 0; j < g.K; ++j) {
      const int m = y * g.s1 + j;
      for (int i = 0; i < g.K; ++i) {
        const int qa = x * g.s1 + g.md + i, qb = qa + s2o;
        for (int c = 0; c < g.C; ++c) {
          const float av = padded_flat(b0, n, c, m, qa, g);
          const float bv = padded_flat(b1, n, c, m, qb, g);
          sum = SUB ? sum + fabsf(av - bv) : fmaf(av, bv, sum);
        }
      }
    }
    top[idx] = sum / (float)(g.K * g.K * g.C);
  }
}


// ---------------------------------------------------------------------------------------------------------
// Tiled forward for the layer as the networks use it: kernel_size 1, stride_1 1, MULTIPLY, and a padding that covers the overshoot
// of the left mode (then everything outside a row reads zero and the flat indexing never reaches data).
// Workgroup = one image row x 32 output columns; 8 groups of 32 lanes, group g owns the displacements g, g+8, g+16, ...; channels are
// staged through LDS 16 at a time (the 32 values of map 0 and the 32 + span values of map 1 the tile needs), so every value is
// read from memory once per tile instead of once per output, and the map-0 value is reused across a lane's displacements.
// ---------------------------------------------------------------------------------------------------------
constexpr int kC1Tile = 32, kC1Groups = 256 / kC1Tile, kC1Chunk = 16;

template <int NO>
__global__ void __launch_bounds__(256) corr1d_fwd_tiled(const float* __restrict__ b0, const float* __restrict__ b1, float* __restrict__ top, Corr1dGeom g,
                                                        int bw /* columns of map 1 staged per tile */) {
  extern __shared__ float lds[];
  float* As = lds;                              // [kC1Chunk][kC1Tile]
  float* Bs = lds + kC1Chunk * kC1Tile;         // [kC1Chunk][bw]
  const int lane = threadIdx.x % kC1Tile, grp = threadIdx.x / kC1Tile;
  const int x0 = blockIdx.x * kC1Tile;
  const int y = blockIdx.y % g.H, n = blockIdx.y / g.H;
  const int xa0 = x0 + g.md - g.pad;            // first map-0 column of the tile (unpadded coordinates)
  const int xb0 = xa0 + g.xshift * g.s2;        // first map-1 column any displacement of the tile reads
  const size_t plane = (size_t)g.H * g.W;
  const float* a_row = b0 + (size_t)n * g.C * plane + (size_t)y * g.W;
  const float* b_row = b1 + (size_t)n * g.C * plane + (size_t)y * g.W;
  float acc[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) acc[j] = 0.f;
  for (int c0 = 0; c0 < g.C; c0 += kC1Chunk) {
    __syncthreads();
    for (int i = threadIdx.x; i < kC1Chunk * kC1Tile; i += 256) {
      const int c = i / kC1Tile, x = xa0 + i % kC1Tile;
      As[i] = (c0 + c < g.C && x >= 0 && x < g.W) ? a_row[(size_t)(c0 + c) * plane + x] : 0.f;
    }
    for (int i = threadIdx.x; i < kC1Chunk * bw; i += 256) {
      const int c = i / bw, x = xb0 + i % bw;
      Bs[i] = (c0 + c < g.C && x >= 0 && x < g.W) ? b_row[(size_t)(c0 + c) * plane + x] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kC1Chunk; ++c) {
      const float a = As[c * kC1Tile + lane];
      const float* brow = Bs + c * bw + lane + grp * g.s2;
#pragma unroll
      for (int j = 0; j < NO; ++j) acc[j] = fmaf(a, brow[kC1Groups * j * g.s2], acc[j]);
    }
  }
  const int x = x0 + lane;
  if (x < g.topW) {
#pragma unroll
    for (int j = 0; j < NO; ++j) {
      const int o = grp + kC1Groups * j;
      if (o < g.topC) top[(((size_t)n * g.topC + o) * g.topH + y) * g.topW + x] = acc[j] / (float)g.C;
    }
  }
}

static bool corr1d_tiled_supported(const Corr1dGeom& g, int* no, int* bw) {
  if (g.K != 1 || g.s1 != 1 || g.type != FN2_CORR_MULTIPLY) return false;
  const int overshoot = -(g.md + g.xshift * g.s2);               // > 0 only in the left mode
  if (overshoot > g.pad) return false;                           // the flat index would reach data of the previous row
  const int need = (g.topC + kC1Groups - 1) / kC1Groups;
  *no = need <= 2 ? 2 : need <= 4 ? 4 : need <= 6 ? 6 : need <= 8 ? 8 : need <= 12 ? 12 : need <= 16 ? 16 : 0;
  if (!*no) return false;
  *bw = kC1Tile + kC1Groups * (*no) * g.s2;                      // covers lane + (grp + 8 j) * s2 for every j < NO
  return (size_t)kC1Chunk * (kC1Tile + *bw) * sizeof(float) <= 60 * 1024 && (long long)g.N * g.H <= 65535;
}

template <int NO>
static void corr1d_tiled_launch(const Corr1dGeom& g, const float* b0, const float* b1, float* top, int bw, hipStream_t st) {
  const size_t lds = (size_t)kC1Chunk * (kC1Tile + bw) * sizeof(float);
  hipLaunchKernelGGL(corr1d_fwd_tiled<NO>, dim3((g.topW + kC1Tile - 1) / kC1Tile, g.N * g.H), dim3(256), lds, st, b0, b1, top, g, bw);
}

// ---------------------------------------------------------------------------------------------------------
// The same layer on the matrix cores (round 6): per image row the cost volume is a BANDED product between the columns of map 0 (M) and of
// map 1 (N), contracted over the channels -- top[o][x] = 1/C sum_c A[c][xa] B[c][xa + o + xshift], stride_2 = 1.  A wave owns one M tile
// of 16 output columns and the NT N tiles the band of that tile touches (16 + ngw - 1 columns of map 1): NT accumulators of
// v_mfma_f32_16x16x4_f32, whose k-ordered fma chain is the generic kernel's `fmaf` loop over the channels bit for bit.  Workgroup = one
// image row x 32 output columns (two waves); the channels arrive 32 at a time through LDS (rows padded to a stride of 16 mod 32 floats:
// the operand reads -- lane (k, m) reads [4 q + k][m] -- are conflict-free); the accumulators go back through LDS, where displacement o of
// column m sits on the diagonal [m][o + m], and leave as 64-byte runs of one displacement.  The tiled kernel above made one LDS read per
// fma (59 us for [4,256,48,96] with 41 displacements); this one reads 5 operands per 4 MFMAs = 4,096 fmas.
// ---------------------------------------------------------------------------------------------------------
using c1_f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kC1MW = 2;            // waves (M tiles of 16 output columns) per workgroup
constexpr int kC1MChunk = 32;       // channels per LDS chunk

constexpr int c1_stride(int row) { return row + ((16 - row % 32) + 32) % 32; }     // >= row, == 16 (mod 32)

template <int NT>
__global__ void __launch_bounds__(64 * kC1MW) corr1d_fwd_mfma(const float* __restrict__ b0, const float* __restrict__ b1, float* __restrict__ top,
                                                               Corr1dGeom g) {
  constexpr int AW = 16 * kC1MW, BW = 16 * (kC1MW - 1 + NT), SA = c1_stride(AW), SB = c1_stride(BW), ST = 16 * NT + 1;
  constexpr int STAGE = kC1MChunk * (SA + SB), TRANS = kC1MW * 16 * ST;
  __shared__ float lds[STAGE > TRANS ? STAGE : TRANS];
  float* As = lds;                               // [kC1MChunk][SA]
  float* Bs = lds + kC1MChunk * SA;              // [kC1MChunk][SB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * AW;
  const int y = blockIdx.y % g.H, n = blockIdx.y / g.H;
  const int xa0 = x0 + g.md - g.pad;             // first map-0 column of the workgroup (unpadded coordinates)
  const int xb0 = xa0 + g.xshift;                // first map-1 column the band of the workgroup reads (stride_2 = 1)
  const size_t plane = (size_t)g.H * g.W;
  const float* a_row = b0 + (size_t)n * g.C * plane + (size_t)y * g.W;
  const float* b_row = b1 + (size_t)n * g.C * plane + (size_t)y * g.W;
  c1_f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = c1_f32x4{0.f, 0.f, 0.f, 0.f};
  const int k = lane >> 4, m = lane & 15;
  // staging: thread (cs = tid / 16, xq = tid % 16) copies column 16 j + xq of channels cs, cs + CS, ... -- every index a shift.  Every load
  // of a chunk is issued before the first one is used: addresses clamped into the blob, out-of-image / out-of-blob values replaced by 0
  // afterwards (a load inside a conditional in front of its LDS store is a round trip of its own: 28 serial round trips per chunk made the
  // first version slower than the kernel it replaces); the loads of chunk c + 1 are in flight while chunk c is multiplied.
  constexpr int CS = 64 * kC1MW / 16;            // channel rows a pass covers
  constexpr int NA = AW / 16, NB = BW / 16, CP = kC1MChunk / CS;
  const int cs = tid >> 4, xq = tid & 15;
  float va[NA][CP], vb[NB][CP];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int x = min(max(xa0 + 16 * j + xq, 0), g.W - 1);
#pragma unroll
      for (int i = 0; i < CP; ++i) va[j][i] = a_row[(size_t)min(c0 + cs + i * CS, g.C - 1) * plane + x];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int x = min(max(xb0 + 16 * j + xq, 0), g.W - 1);
#pragma unroll
      for (int i = 0; i < CP; ++i) vb[j][i] = b_row[(size_t)min(c0 + cs + i * CS, g.C - 1) * plane + x];
    }
  };
  load_chunk(0);
  for (int c0 = 0; c0 < g.C; c0 += kC1MChunk) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int x = xa0 + 16 * j + xq;
      const bool okx = x >= 0 && x < g.W;
#pragma unroll
      for (int i = 0; i < CP; ++i) As[(cs + i * CS) * SA + 16 * j + xq] = (okx && c0 + cs + i * CS < g.C) ? va[j][i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int x = xb0 + 16 * j + xq;
      const bool okx = x >= 0 && x < g.W;
#pragma unroll
      for (int i = 0; i < CP; ++i) Bs[(cs + i * CS) * SB + 16 * j + xq] = (okx && c0 + cs + i * CS < g.C) ? vb[j][i] : 0.f;
    }
    __syncthreads();
    if (c0 + kC1MChunk < g.C) load_chunk(c0 + kC1MChunk);
    const float* ap = As + k * SA + 16 * wave + m;
    const float* bp = Bs + k * SB + 16 * wave + m;
#pragma unroll
    for (int q = 0; q < kC1MChunk / 4; ++q) {
      const float av = ap[4 * q * SA];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[4 * q * SB + 16 * t], acc[t], 0, 0, 0);
    }
  }
  __syncthreads();                               // the staging buffers become the transposition image
  float* T = lds + wave * 16 * ST;               // [16 columns m][16 NT band positions]: D[row = m][col] of tile t at [m][16 t + col]
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) T[(4 * k + j) * ST + 16 * t + m] = acc[t][j];
  __syncthreads();
  const int x = x0 + 16 * wave + m;
  if (x < g.topW) {
    for (int o = k; o < g.topC; o += 4)          // lanes (k, m): displacement o = k, k + 4, ..., 16 consecutive columns each
      top[(((size_t)n * g.topC + o) * g.topH + y) * g.topW + x] = T[m * ST + o + m] / (float)g.C;
  }
}

static int corr1d_mfma_tiles(const Corr1dGeom& g) {      // N tiles per M tile, or 0 when the kernel does not apply
  if (g.K != 1 || g.s1 != 1 || g.s2 != 1 || g.type != FN2_CORR_MULTIPLY) return 0;
  const int overshoot = -(g.md + g.xshift * g.s2);
  if (overshoot > g.pad) return 0;                       // (as for the tiled kernel: the flat index would reach data of the previous row)
  const int nt = (15 + g.ngw + 15) / 16;
  return (nt <= 8 && (long long)g.N * g.H <= 65535) ? nt : 0;
}

__device__ __forceinline__ int ceil_div1(int a, int s) { return (a >= 0) ? (a + s - 1) / s : -((-a) / s); }
__device__ __forceinline__ int floor_div1(int a, int s) { return (a >= 0) ? a / s : -((-a + s - 1) / s); }

// WHICH = 0: bottom0 diff (CorrelateDataBackward0[Subtract], correlation_layer1d.cu:117-181, :295-359);
// WHICH = 1: bottom1 diff (CorrelateDataBackward1[Subtract], :183-249, :361-423).
template <bool SUB, int WHICH>
__global__ void __launch_bounds__(256) corr1d_bwd(const float* __restrict__ b0, const float* __restrict__ b1,
                                                  const float* __restrict__ top_diff, float* __restrict__ bdiff, Corr1dGeom g) {
  const long long total = (long long)g.N * g.C * g.H * g.W;
  const size_t tplane = (size_t)g.topH * g.topW;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % g.W);
    const int m = (int)((idx / g.W) % g.H);
    const int c = (int)((idx / g.W / g.H) % g.C);
    const int n = (int)(idx / g.W / g.H / g.C);
    const int l = x + g.pad;
    const float* td = top_diff + (size_t)n * g.topC * tplane;
    int ymin = ceil_div1(m - 2 * g.kr, g.s1);
    int ymax = floor_div1(m, g.s1);
    float sum = 0.f;
    for (int o = g.xshift; o < g.xshift + g.ngw; ++o) {
      const int s2o = g.s2 * o;
      const int sx = (WHICH == 0) ? 0 : s2o;
      int xmin = ceil_div1(l - 2 * g.kr - g.md - sx, g.s1);
      int xmax = floor_div1(l - g.md - sx, g.s1);
      if (!(xmax >= 0 && ymax >= 0 && xmin <= g.topW - 1 && ymin <= g.topH - 1)) continue;
      xmin = max(0, xmin); xmax = min(g.topW - 1, xmax);
      const int y0 = max(0, ymin), y1 = min(g.topH - 1, ymax);
      const int q = (WHICH == 0) ? l + s2o : l - s2o;
      float coef;
      if (!SUB) {
        coef = padded_flat((WHICH == 0) ? b1 : b0, n, c, m, q, g);
      } else {
        const float v0 = padded_flat(b0, n, c, m, q, g), v1 = padded_flat(b1, n, c, m, q, g);
        coef = (WHICH == 0) ? ((v0 >= v1) ? 1.f : -1.f) : ((v0 >= v1) ? -1.f : 1.f);
      }
      const float* t = td + (size_t)(o - g.xshift) * tplane;
      for (int yy = y0; yy <= y1; ++yy)
        for (int xx = xmin; xx <= xmax; ++xx) sum = fmaf(t[(size_t)yy * g.topW + xx], coef, sum);
    }
    bdiff[idx] = sum / (float)((g.kr * 2 + 1) * (g.kr * 2 + 1) * g.C);
  }
}

}  // namespace fn2

using namespace fn2;

// test hook (fn2_debug_set_correlation_impl(1) also forces the generic 1-D kernels)
namespace fn2 { int g_corr1d_force_generic = 0; int g_corr1d_no_mfma = 0; }      // (impl 2: the LDS-tiled VALU kernel instead of the MFMA one)

FN2_API int fn2_correlation1d_out_shape(const fn2_corr_params* p, int C, int H, int W, int* topC, int* topH, int* topW) {
  Corr1dGeom g;
  int rc = corr1d_geometry(p, 1, C, H, W, &g);
  if (rc) return rc;
  if (topC) *topC = g.topC;
  if (topH) *topH = g.topH;
  if (topW) *topW = g.topW;
  return FN2_OK;
}

FN2_API int fn2_correlation1d_forward(const fn2_corr_params* p, const float* bottom0, const float* bottom1, float* top,
                                      int N, int C, int H, int W, void* stream) {
  Corr1dGeom g;
  int rc = corr1d_geometry(p, N, C, H, W, &g);
  if (rc) return rc;
  if (N == 0) return FN2_OK;
  if (!bottom0 || !bottom1 || !top) return fail(FN2_ERR_INVALID_ARG, "correlation1d_forward: NULL blob pointer");
  int no = 0, bw = 0;
  const int nt = (g_corr1d_force_generic || g_corr1d_no_mfma) ? 0 : corr1d_mfma_tiles(g);
  if (nt) {
    const dim3 grid((unsigned)((g.topW + 16 * kC1MW - 1) / (16 * kC1MW)), (unsigned)(g.N * g.H));
    hipStream_t st = as_stream(stream);
#define FN2_C1(NT_) case NT_: hipLaunchKernelGGL(corr1d_fwd_mfma<NT_>, grid, dim3(64 * kC1MW), 0, st, bottom0, bottom1, top, g); break
    switch (nt) { FN2_C1(1); FN2_C1(2); FN2_C1(3); FN2_C1(4); FN2_C1(5); FN2_C1(6); FN2_C1(7); default: hipLaunchKernelGGL(corr1d_fwd_mfma<8>, grid, dim3(64 * kC1MW), 0, st, bottom0, bottom1, top, g); break; }
#undef FN2_C1
    return check_launch("correlation1d_forward");
  }
  if (!g_corr1d_force_generic && corr1d_tiled_supported(g, &no, &bw)) {
    hipStream_t st = as_stream(stream);
    switch (no) {
      case 2: corr1d_tiled_launch<2>(g, bottom0, bottom1, top, bw, st); break;
      case 4: corr1d_tiled_launch<4>(g, bottom0, bottom1, top, bw, st); break;
      case 6: corr1d_tiled_launch<6>(g, bottom0, bottom1, top, bw, st); break;
      case 8: corr1d_tiled_launch<8>(g, bottom0, bottom1, top, bw, st); break;
      case 12: corr1d_tiled_launch<12>(g, bottom0, bottom1, top, bw, st); break;
      default: corr1d_tiled_launch<16>(g, bottom0, bottom1, top, bw, st); break;
    }
    return check_launch("correlation1d_forward");
  }
  const unsigned blocks = blocks_for((long long)N * g.topC * g.topH * g.topW, 256);
  if (g.type == FN2_CORR_MULTIPLY)
    hipLaunchKernelGGL(corr1d_fwd<false>, dim3(blocks), dim3(256), 0, as_stream(stream), bottom0, bottom1, top, g);
  else
    hipLaunchKernelGGL(corr1d_fwd<true>, dim3(blocks), dim3(256), 0, as_stream(stream), bottom0, bottom1, top, g);
  return check_launch("correlation1d_forward");
}

FN2_API int fn2_correlation1d_backward(const fn2_corr_params* p, const float* bottom0, const float* bottom1, const float* top_diff,
                                       float* bottom0_diff, float* bottom1_diff, int N, int C, int H, int W, void* stream) {
  Corr1dGeom g;
  int rc = corr1d_geometry(p, N, C, H, W, &g);
  if (rc) return rc;
  if (N == 0) return FN2_OK;
  if (!bottom0 || !bottom1 || !top_diff) return fail(FN2_ERR_INVALID_ARG, "correlation1d_backward: NULL blob pointer");
  hipStream_t st = as_stream(stream);
  const unsigned blocks = blocks_for((long long)N * C * H * W, 256);
  const bool sub = (g.type == FN2_CORR_SUBTRACT);
  if (bottom0_diff) {
    if (!sub) hipLaunchKernelGGL((corr1d_bwd<false, 0>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom0_diff, g);
    else hipLaunchKernelGGL((corr1d_bwd<true, 0>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom0_diff, g);
  }
  if (bottom1_diff) {
    if (!sub) hipLaunchKernelGGL((corr1d_bwd<false, 1>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom1_diff, g);
    else hipLaunchKernelGGL((corr1d_bwd<true, 1>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom1_diff, g);
  }
  return check_launch("correlation1d_backward");
}
