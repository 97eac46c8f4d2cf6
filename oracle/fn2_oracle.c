/*
 * fn2_oracle.c -- CPU restatement of the FlowNet2 hot-path layers of lmb-freiburg/flownet2.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under flownet2_amd/ may import, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS: the reference ships NO CPU implementation for Correlation / Resample / L1Loss /
 * Downsample (correlation_layer.cpp:87-96, resample_layer.cpp:58-62, l1loss_layer.cpp:93-102,
 * downsample_layer.cpp:60-64 are NOT_IMPLEMENTED / LOG(FATAL)), has no test or golden vector for
 * any FlowNet2 layer (src/caffe/test/ is stock BVLC), and its build (boost/glog/protobuf/BLAS/
 * CUDA) cannot run in this image.  This file therefore restates the reference's CUDA kernels line
 * by line in plain C.  Two pins exist (see oracle/README.md):
 *   (1) oracle/_ref: the reference's OWN .cu/.cpp sources for Correlation, Correlation1D, FlowWarp, Resample,
 *       ChannelNorm, Downsample and L1Loss compiled in place with hipcc against stand-in caffe
 *       headers and run on an MI355X; tests/golden/ref_golden.npz holds the outputs they produced
 *       and tests/test_golden.py checks every reference-layer function below against them
 *       (PINNED for all seven layers; FlowAugmentation and DataAugmentation for given coefficients, added later, likewise).  L1LossLayer instantiates the stock Eltwise / Power /
 *       Convolution layers (l1loss_layer.cpp:19-62): those sources (+ base_conv_layer.cpp,
 *       im2col.{cpp,cu}) are compiled in place as well; the only replaced part is the cuBLAS /
 *       CBLAS calls underneath them (oracle/ref_compat/caffe/util/math_functions.hpp: plain
 *       dot / gemm / axpby with the textbook meaning).
 *   (2) an independent fp64 re-derivation + autograd gradient checks (tests/test_oracle.py).
 * The stock-layer fast paths at the end of this file (flow heads, bias + ReLU, stem convolution,
 * im2col / col2im) restate stock Caffe layers; oracle/_ref also builds the reference's Convolution,
 * Deconvolution and ReLU layers (same SGEMM stand-in) and the golden file holds their outputs
 * (stock_* arrays): PINNED as well, in addition to torch's fp64 convolutions (tests/test_oracle.py).
 *
 * The CustomData sample format at the end of this file is PINNED too: decode against the reference's custom_data_layer.cpp compiled
 * in place over an in-memory LMDB stand-in (cdata* golden arrays), the Datum wire format against the protobuf runtime; the writer's
 * packing (the tool needs OpenCV) is restated and checked through the pinned reader.
 *
 * All file:line citations are relative to the reference tree.
 * Arithmetic notes: nvcc contracts `sum += a*b` into an FMA by default, so the restatement uses
 * fmaf() where the reference has that pattern; summation ORDER follows the reference kernels.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <float.h>

#include "../include/flownet2_hip.h"

#if defined(_OPENMP)
#include <omp.h>
#endif

#define FN2_API __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

FN2_API int fn2_oracle_num_threads(void) {
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * Correlation shapes: CorrelationLayer::LayerSetUp / Reshape, correlation_layer.cpp:13-84
 * ---------------------------------------------------------------------------------------------- */
typedef struct corr_geom {
  int kr, border, pH, pW, topH, topW, ngr, ngw, topC;
} corr_geom;

static int corr_geometry(const fn2_corr_params* p, int C, int H, int W, corr_geom* g) {
  if (!p || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  if (p->kernel_size < 1 || p->kernel_size % 2 == 0) return FN2_ERR_INVALID_ARG;   /* cpp:22 */
  if (p->stride1 < 1 || p->stride2 < 1 || p->max_displacement < 0 || p->pad < 0) return FN2_ERR_INVALID_ARG;
  g->kr = (p->kernel_size - 1) / 2;                      /* cpp:56 */
  g->border = p->max_displacement + g->kr;               /* cpp:57 */
  g->pH = H + 2 * p->pad;                                /* cpp:52 */
  g->pW = W + 2 * p->pad;                                /* cpp:53 */
  g->topW = (int)ceilf((float)(g->pW - g->border * 2) / (float)p->stride1);   /* cpp:59 */
  g->topH = (int)ceilf((float)(g->pH - g->border * 2) / (float)p->stride1);   /* cpp:60 */
  if (g->topW < 1 || g->topH < 1) return FN2_ERR_INVALID_ARG;                 /* cpp:62-63 */
  g->ngr = p->max_displacement / p->stride2;             /* cpp:66 */
  g->ngw = g->ngr * 2 + 1;                               /* cpp:67 */
  g->topC = g->ngw * g->ngw;                             /* cpp:70 */
  /* The kernels index the padded blob at y1+j+s2p with no bounds check (correlation_layer.cu:95);
   * pad < max_displacement reads outside it.  Refuse instead of reproducing undefined behaviour. */
  if (p->pad < p->max_displacement) return FN2_ERR_INVALID_ARG;
  if (p->corr_type != FN2_CORR_MULTIPLY && p->corr_type != FN2_CORR_SUBTRACT) return FN2_ERR_INVALID_ARG;
  return FN2_OK;
}

FN2_API int fn2_correlation_out_shape_cpu(const fn2_corr_params* p, int C, int H, int W,
                                          int* topC, int* topH, int* topW) {
  corr_geom g;
  int rc = corr_geometry(p, C, H, W, &g);
  if (rc) return rc;
  if (topC) *topC = g.topC;
  if (topH) *topH = g.topH;
  if (topW) *topW = g.topW;
  return FN2_OK;
}

/* blob_rearrange_kernel2, correlation_layer.cu:23-42 + the cudaMemset at :447-448:
 * NCHW -> zero-padded N (H+2p) (W+2p) C */
static float* rearrange_padded(const float* in, int N, int C, int H, int W, int pad) {
  const int pH = H + 2 * pad, pW = W + 2 * pad;
  float* out = (float*)calloc((size_t)N * pH * pW * C, sizeof(float));
  if (!out) return NULL;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x)
        for (int ch = 0; ch < C; ++ch)
          out[(((size_t)n * pH + (y + pad)) * pW + (x + pad)) * C + ch] =
              in[(((size_t)n * C + ch) * H + y) * W + x];
  return out;
}

/* CorrelateData, correlation_layer.cu:45-114 (MULTIPLY) and CorrelateDataSubtract :252-293. */
FN2_API int fn2_correlation_forward_cpu(const fn2_corr_params* p, const float* bottom0,
                                        const float* bottom1, float* top, int N, int C, int H, int W) {
  corr_geom g;
  int rc = corr_geometry(p, C, H, W, &g);
  if (rc) return rc;
  float* r0 = rearrange_padded(bottom0, N, C, H, W, p->pad);
  float* r1 = rearrange_padded(bottom1, N, C, H, W, p->pad);
  if (!r0 || !r1) { free(r0); free(r1); return FN2_ERR_WORKSPACE; }
  const int K = p->kernel_size, md = p->max_displacement, s1 = p->stride1, s2 = p->stride2;
  const int pH = g.pH, pW = g.pW;
  const size_t topcount = (size_t)g.topC * g.topH * g.topW;
  const int sumelems = K * K * C;                                          /* :110 / :288 */

  if (p->corr_type == FN2_CORR_MULTIPLY) {
#pragma omp parallel for collapse(3) schedule(static)
    for (int item = 0; item < N; ++item)
      for (int by = 0; by < g.topH; ++by)          /* blockIdx.y */
        for (int bx = 0; bx < g.topW; ++bx) {      /* blockIdx.x */
          const int x1 = bx * s1 + md;             /* :56 */
          const int y1 = by * s1 + md;             /* :57 */
          for (int tc = 0; tc < g.topC; ++tc) {
            const int s2o = (tc % g.ngw - g.ngr) * s2;   /* :81 */
            const int s2p = (tc / g.ngw - g.ngr) * s2;   /* :82 */
            /* 32 lanes, lane t accumulates channels ch = t, t+32, ... over (j,i)  (:84-97) */
            float lane[32];
            for (int t = 0; t < 32; ++t) lane[t] = 0.f;
            for (int j = 0; j < K; ++j)
              for (int i = 0; i < K; ++i) {
                const float* a = r0 + (((size_t)item * pH + y1 + j) * pW + x1 + i) * C;
                const float* b = r1 + (((size_t)item * pH + y1 + s2p + j) * pW + x1 + s2o + i) * C;
                for (int ch = 0; ch < C; ++ch) lane[ch & 31] = fmaf(a[ch], b[ch], lane[ch & 31]);
              }
            float total = 0.f;                                         /* :103-106 lane-order sum */
            for (int t = 0; t < 32; ++t) total += lane[t];
            top[(size_t)item * topcount + ((size_t)tc * g.topH + by) * g.topW + bx] =
                total / (float)sumelems;                               /* :109 */
          }
        }
  } else {
    const int kr = g.kr;
#pragma omp parallel for collapse(3) schedule(static)
    for (int item = 0; item < N; ++item)
      for (int c = 0; c < g.topC; ++c)
        for (int y = 0; y < g.topH; ++y)
          for (int x = 0; x < g.topW; ++x) {
            const int s2o = (c % g.ngw - g.ngr) * s2;                  /* :264 */
            const int s2p = (c / g.ngw - g.ngr) * s2;                  /* :265 */
            const int x1 = x * s1 + kr + md;                           /* :268 */
            const int y1 = y * s1 + kr + md;                           /* :269 */
            float sum = 0.f;
            for (int j = -kr; j <= kr; ++j)
              for (int i = -kr; i <= kr; ++i) {
                const float* a = r0 + (((size_t)item * pH + y1 + j) * pW + x1 + i) * C;
                const float* b = r1 + (((size_t)item * pH + y1 + s2p + j) * pW + x1 + s2o + i) * C;
                for (int l = 0; l < C; ++l) sum += fabsf(a[l] - b[l]);  /* :285 */
              }
            top[(size_t)item * topcount + ((size_t)c * g.topH + y) * g.topW + x] = sum / (float)sumelems;
          }
  }
  free(r0);
  free(r1);
  return FN2_OK;
}

/* Integer ceil/floor division the way the kernels do it (ROUND_OFF trick, correlation_layer.cu:131-140). */
static inline int ceil_div_ro(int a, int s) { const int ro = 50000, ros = s * ro; return (a + ros - 1) / s + 1 - ro; }
static inline int floor_div_ro(int a, int s) { const int ro = 50000, ros = s * ro; return (a + ros) / s - ro; }

/* CorrelateDataBackward0 / 1, correlation_layer.cu:117-249; ...Subtract :297-427. */
FN2_API int fn2_correlation_backward_cpu(const fn2_corr_params* p, const float* bottom0,
                                         const float* bottom1, const float* top_diff,
                                         float* bottom0_diff, float* bottom1_diff,
                                         int N, int C, int H, int W) {
  corr_geom g;
  int rc = corr_geometry(p, C, H, W, &g);
  if (rc) return rc;
  float* r0 = rearrange_padded(bottom0, N, C, H, W, p->pad);
  float* r1 = rearrange_padded(bottom1, N, C, H, W, p->pad);
  if (!r0 || !r1) { free(r0); free(r1); return FN2_ERR_WORKSPACE; }
  const int md = p->max_displacement, s1 = p->stride1, s2 = p->stride2, pad = p->pad;
  const int kr = g.kr, pH = g.pH, pW = g.pW, ngr = g.ngr, ngw = g.ngw;
  const int topH = g.topH, topW = g.topW, topC = g.topC;
  const int sumelems = (kr * 2 + 1) * (kr * 2 + 1) * C;
  const size_t bottomcount = (size_t)C * H * W;
  const int sub = (p->corr_type == FN2_CORR_SUBTRACT);

  if (bottom0_diff) {
#pragma omp parallel for collapse(3) schedule(static)
    for (int item = 0; item < N; ++item)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          const int l = xx + pad, m = yy + pad;                                   /* :125-126 */
          int xmin = ceil_div_ro(l - 2 * kr - md, s1);                             /* :135 */
          int ymin = ceil_div_ro(m - 2 * kr - md, s1);                             /* :136 */
          int xmax = floor_div_ro(l - md, s1);                                     /* :139 */
          int ymax = floor_div_ro(m - md, s1);                                     /* :140 */
          const int live = (xmax >= 0 && ymax >= 0 && xmin <= topW - 1 && ymin <= topH - 1);   /* :144 */
          if (live) { xmin = imax(0, xmin); xmax = imin(topW - 1, xmax); ymin = imax(0, ymin); ymax = imin(topH - 1, ymax); }
          for (int n = 0; n < C; ++n) {
            float sum = 0.f;
            if (live)
              for (int pp = -ngr; pp <= ngr; ++pp)
                for (int o = -ngr; o <= ngr; ++o) {
                  const int s2o = s2 * o, s2p = s2 * pp;
                  const size_t idx = (((size_t)item * pH + (m + s2p)) * pW + (l + s2o)) * C + n;   /* :158 */
                  float coef;
                  if (!sub) coef = r1[idx];                                         /* :159 */
                  else coef = (r0[idx] >= r1[idx]) ? 1.f : -1.f;                    /* :341 */
                  const int op = (pp + ngr) * ngw + (o + ngr);                      /* :162 */
                  const size_t off = ((size_t)item * topC + op);
                  for (int y = ymin; y <= ymax; ++y)
                    for (int x = xmin; x <= xmax; ++x)
                      sum = fmaf(top_diff[(off * topH + y) * topW + x], coef, sum);   /* :168 */
                }
            bottom0_diff[(size_t)item * bottomcount + ((size_t)n * H + yy) * W + xx] = sum / (float)sumelems;   /* :175-176 */
          }
        }
  }
  if (bottom1_diff) {
#pragma omp parallel for collapse(3) schedule(static)
    for (int item = 0; item < N; ++item)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          const int l = xx + pad, m = yy + pad;
          for (int n = 0; n < C; ++n) {
            float sum = 0.f;
            for (int pp = -ngr; pp <= ngr; ++pp)
              for (int o = -ngr; o <= ngr; ++o) {
                const int s2o = s2 * o, s2p = s2 * pp;
                int xmin = ceil_div_ro(l - 2 * kr - md - s2o, s1);                 /* :212 */
                int ymin = ceil_div_ro(m - 2 * kr - md - s2p, s1);                 /* :213 */
                int xmax = floor_div_ro(l - md - s2o, s1);                         /* :216 */
                int ymax = floor_div_ro(m - md - s2p, s1);                         /* :217 */
                if (xmax >= 0 && ymax >= 0 && xmin <= topW - 1 && ymin <= topH - 1) {   /* :219 */
                  xmin = imax(0, xmin); xmax = imin(topW - 1, xmax);
                  ymin = imax(0, ymin); ymax = imin(topH - 1, ymax);
                  const size_t idx = (((size_t)item * pH + (m - s2p)) * pW + (l - s2o)) * C + n;   /* :228 */
                  float coef;
                  if (!sub) coef = r0[idx];                                        /* :229 */
                  else coef = (r0[idx] >= r1[idx]) ? -1.f : 1.f;                   /* :408 */
                  const int op = (pp + ngr) * ngw + (o + ngr);
                  const size_t off = ((size_t)item * topC + op);
                  for (int y = ymin; y <= ymax; ++y)
                    for (int x = xmin; x <= xmax; ++x)
                      sum = fmaf(top_diff[(off * topH + y) * topW + x], coef, sum);   /* :238 */
                }
              }
            bottom1_diff[(size_t)item * bottomcount + ((size_t)n * H + yy) * W + xx] = sum / (float)sumelems;   /* :245-246 */
          }
        }
  }
  free(r0);
  free(r1);
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Correlation1D: correlation_layer1d.cpp:12-92 (LayerSetUp, Reshape), correlation_layer1d.cu:23-616.
 * Horizontal displacements only; the scratch blob is [N, H, W+2p, C] (padding in x only, :26-44).
 * The kernels address it with a flat index and no bounds check.  With single_direction = -1 the first
 * displacement is x_shift = -grid_width (:466-471), one step beyond the grid radius, so columns left of a
 * row start are read: in the flat blob that is the tail of the previous row.  flat_at() keeps the flat
 * indexing; an index outside the blob (undefined in the reference) reads as 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct corr1d_geom {
  int kr, pW, topH, topW, ngr, ngw, topC, xshift;
} corr1d_geom;

static int corr1d_geometry(const fn2_corr_params* p, int C, int H, int W, corr1d_geom* g) {
  if (!p || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  if (p->kernel_size < 1 || p->kernel_size % 2 == 0) return FN2_ERR_INVALID_ARG;            /* cpp:22 */
  if (p->stride1 < 1 || p->stride2 < 1 || p->max_displacement < 0 || p->pad < 0) return FN2_ERR_INVALID_ARG;
  if (p->single_direction < -1 || p->single_direction > 1) return FN2_ERR_INVALID_ARG;       /* cpp:29 */
  if (p->corr_type != FN2_CORR_MULTIPLY && p->corr_type != FN2_CORR_SUBTRACT) return FN2_ERR_INVALID_ARG;
  g->kr = (p->kernel_size - 1) / 2;                                                          /* cpp:58 */
  const int border = p->max_displacement + g->kr;                                            /* cpp:59 */
  g->pW = W + 2 * p->pad;                                                                    /* cpp:55 */
  g->topW = (int)ceilf((float)(g->pW - border * 2) / (float)p->stride1);                     /* cpp:61 */
  g->topH = (int)ceilf((float)(H - g->kr * 2) / (float)p->stride1);                          /* cpp:62 */
  if (g->topW < 1 || g->topH < 1) return FN2_ERR_INVALID_ARG;                                /* cpp:64-65 */
  g->ngr = p->max_displacement / p->stride2;                                                 /* cpp:68 */
  g->ngw = p->single_direction != 0 ? g->ngr + 1 : g->ngr * 2 + 1;                           /* cpp:70-74 */
  g->topC = g->ngw;                                                                          /* cpp:78 */
  g->xshift = -g->ngr;                                                                       /* cu:466 */
  if (p->single_direction == -1) g->xshift = -g->ngw;                                        /* cu:467-468 */
  else if (p->single_direction == 1) g->xshift = 0;                                          /* cu:469-470 */
  return FN2_OK;
}

FN2_API int fn2_correlation1d_out_shape_cpu(const fn2_corr_params* p, int C, int H, int W,
                                            int* topC, int* topH, int* topW) {
  corr1d_geom g;
  int rc = corr1d_geometry(p, C, H, W, &g);
  if (rc) return rc;
  if (topC) *topC = g.topC;
  if (topH) *topH = g.topH;
  if (topW) *topW = g.topW;
  return FN2_OK;
}

/* corr1d::blob_rearrange_kernel2, correlation_layer1d.cu:25-44 + the cudaMemsets :447-448: NCHW -> N H (W+2p) C */
static float* rearrange_padded_x(const float* in, int N, int C, int H, int W, int pad) {
  const int pW = W + 2 * pad;
  float* out = (float*)calloc((size_t)N * H * pW * C + 1, sizeof(float));
  if (!out) return NULL;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x)
        for (int ch = 0; ch < C; ++ch)
          out[(((size_t)n * H + y) * pW + (x + pad)) * C + ch] = in[(((size_t)n * C + ch) * H + y) * W + x];
  return out;
}

static inline float flat_at(const float* r, long long idx, long long count) { return (idx < 0 || idx >= count) ? 0.f : r[idx]; }

FN2_API int fn2_correlation1d_forward_cpu(const fn2_corr_params* p, const float* bottom0, const float* bottom1,
                                          float* top, int N, int C, int H, int W) {
  corr1d_geom g;
  int rc = corr1d_geometry(p, C, H, W, &g);
  if (rc) return rc;
  float* r0 = rearrange_padded_x(bottom0, N, C, H, W, p->pad);
  float* r1 = rearrange_padded_x(bottom1, N, C, H, W, p->pad);
  if (!r0 || !r1) { free(r0); free(r1); return FN2_ERR_WORKSPACE; }
  const int K = p->kernel_size, md = p->max_displacement, s1 = p->stride1, s2 = p->stride2, kr = g.kr;
  const long long pW = g.pW, count = (long long)N * H * pW * C;
  const size_t topcount = (size_t)g.topC * g.topH * g.topW;
  const int sumelems = K * K * C;                                                            /* :107 / :288 */
  const int sub = (p->corr_type == FN2_CORR_SUBTRACT);
#pragma omp parallel for collapse(3) schedule(static)
  for (int item = 0; item < N; ++item)
    for (int by = 0; by < g.topH; ++by)
      for (int bx = 0; bx < g.topW; ++bx) {
        const int x1 = bx * s1 + md;                                                         /* :56 (MULTIPLY); :267 = x*s1+kr+md with i from -kr */
        const int y1 = by * s1;                                                              /* :57 / :268 */
        for (int tc = 0; tc < g.topC; ++tc) {
          const int s2o = (tc % g.ngw + g.xshift) * s2;                                      /* :80 / :263 */
          float total = 0.f;
          if (!sub) {
            float lane[32];                                                                  /* :75, lanes as in the 2-D kernel */
            for (int t = 0; t < 32; ++t) lane[t] = 0.f;
            for (int j = 0; j < K; ++j)
              for (int i = 0; i < K; ++i) {
                const long long ia = (((long long)item * H + y1 + j) * pW + x1 + i) * C;          /* :66 */
                const long long ib = (((long long)item * H + y1 + j) * pW + x1 + s2o + i) * C;    /* :89 */
                for (int ch = 0; ch < C; ++ch)
                  lane[ch & 31] = fmaf(flat_at(r0, ia + ch, count), flat_at(r1, ib + ch, count), lane[ch & 31]);   /* :91 */
              }
            for (int t = 0; t < 32; ++t) total += lane[t];                                   /* :100-103 */
          } else {
            for (int j = -kr; j <= kr; ++j)
              for (int i = -kr; i <= kr; ++i) {
                const long long ia = (((long long)item * H + y1 + kr + j) * pW + x1 + kr + i) * C;        /* :280 */
                const long long ib = (((long long)item * H + y1 + kr + j) * pW + x1 + kr + s2o + i) * C;  /* :281 */
                for (int l = 0; l < C; ++l) total += fabsf(flat_at(r0, ia + l, count) - flat_at(r1, ib + l, count));   /* :284 */
              }
          }
          top[(size_t)item * topcount + ((size_t)tc * g.topH + by) * g.topW + bx] = total / (float)sumelems;   /* :106 / :289 */
        }
      }
  free(r0);
  free(r1);
  return FN2_OK;
}

FN2_API int fn2_correlation1d_backward_cpu(const fn2_corr_params* p, const float* bottom0, const float* bottom1,
                                           const float* top_diff, float* bottom0_diff, float* bottom1_diff,
                                           int N, int C, int H, int W) {
  corr1d_geom g;
  int rc = corr1d_geometry(p, C, H, W, &g);
  if (rc) return rc;
  float* r0 = rearrange_padded_x(bottom0, N, C, H, W, p->pad);
  float* r1 = rearrange_padded_x(bottom1, N, C, H, W, p->pad);
  if (!r0 || !r1) { free(r0); free(r1); return FN2_ERR_WORKSPACE; }
  const int md = p->max_displacement, s1 = p->stride1, s2 = p->stride2, pad = p->pad, kr = g.kr;
  const int topH = g.topH, topW = g.topW, topC = g.topC;
  const long long pW = g.pW, count = (long long)N * H * pW * C;
  const int sumelems = (kr * 2 + 1) * (kr * 2 + 1) * C;
  const size_t bottomcount = (size_t)C * H * W;
  const int sub = (p->corr_type == FN2_CORR_SUBTRACT);

  if (bottom0_diff) {
#pragma omp parallel for collapse(3) schedule(static)
    for (int item = 0; item < N; ++item)
      for (int m = 0; m < H; ++m)
        for (int xx = 0; xx < W; ++xx) {
          const int l = xx + pad;                                                            /* :124 */
          int xmin = ceil_div_ro(l - 2 * kr - md, s1);                                       /* :134 */
          int ymin = ceil_div_ro(m - 2 * kr, s1);                                            /* :135 */
          int xmax = floor_div_ro(l - md, s1);                                               /* :138 */
          int ymax = floor_div_ro(m, s1);                                                    /* :139 */
          const int live = (xmax >= 0 && ymax >= 0 && xmin <= topW - 1 && ymin <= topH - 1); /* :143 */
          if (live) { xmin = imax(0, xmin); xmax = imin(topW - 1, xmax); ymin = imax(0, ymin); ymax = imin(topH - 1, ymax); }
          for (int n = 0; n < C; ++n) {
            float sum = 0.f;
            if (live)
              for (int o = g.xshift; o < g.xshift + g.ngw; ++o) {                            /* :152 */
                const int s2o = s2 * o;
                const long long idx = (((long long)item * H + m) * pW + (l + s2o)) * C + n;  /* :156 / :332 */
                float coef;
                if (!sub) coef = flat_at(r1, idx, count);                                    /* :157 */
                else coef = (flat_at(r0, idx, count) >= flat_at(r1, idx, count)) ? 1.f : -1.f;   /* :333-335 */
                const size_t off = (size_t)item * topC + (o - g.xshift);                     /* :160-161 */
                for (int y = ymin; y <= ymax; ++y)
                  for (int x = xmin; x <= xmax; ++x)
                    sum = fmaf(top_diff[(off * topH + y) * topW + x], coef, sum);            /* :166 */
              }
            bottom0_diff[(size_t)item * bottomcount + ((size_t)n * H + m) * W + xx] = sum / (float)sumelems;   /* :173-175 */
          }
        }
  }
  if (bottom1_diff) {
#pragma omp parallel for collapse(3) schedule(static)
    for (int item = 0; item < N; ++item)
      for (int m = 0; m < H; ++m)
        for (int xx = 0; xx < W; ++xx) {
          const int l = xx + pad;
          for (int n = 0; n < C; ++n) {
            float sum = 0.f;
            for (int o = g.xshift; o < g.xshift + g.ngw; ++o) {                              /* :204 */
              const int s2o = s2 * o;
              int xmin = ceil_div_ro(l - 2 * kr - md - s2o, s1);                             /* :210 */
              int ymin = ceil_div_ro(m - 2 * kr, s1);                                        /* :211 */
              int xmax = floor_div_ro(l - md - s2o, s1);                                     /* :214 */
              int ymax = floor_div_ro(m, s1);                                                /* :215 */
              if (xmax >= 0 && ymax >= 0 && xmin <= topW - 1 && ymin <= topH - 1) {          /* :217 */
                xmin = imax(0, xmin); xmax = imin(topW - 1, xmax);
                ymin = imax(0, ymin); ymax = imin(topH - 1, ymax);
                const long long idx = (((long long)item * H + m) * pW + (l - s2o)) * C + n;  /* :226 / :397 */
                float coef;
                if (!sub) coef = flat_at(r0, idx, count);                                    /* :227 */
                else coef = (flat_at(r0, idx, count) >= flat_at(r1, idx, count)) ? -1.f : 1.f;   /* :398-400 */
                const size_t off = (size_t)item * topC + (o - g.xshift);
                for (int y = ymin; y <= ymax; ++y)
                  for (int x = xmin; x <= xmax; ++x)
                    sum = fmaf(top_diff[(off * topH + y) * topW + x], coef, sum);            /* :236 */
              }
            }
            bottom1_diff[(size_t)item * bottomcount + ((size_t)n * H + m) * W + xx] = sum / (float)sumelems;   /* :243-245 */
          }
        }
  }
  free(r0);
  free(r1);
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * FlowWarp: flow_warp_layer.cpp:58-117 (Forward_cpu), GPU twin flow_warp_layer.cu:58-122.
 * The GPU kernel pre-multiplies the four bilinear coefficients (:93-96) and then sums
 * coeffTL*TL + coeffTR*TR + coeffBL*BL + coeffBR*BR (:110-114); the CPU twin writes the same
 * expression left-associated.  nvcc contracts that chain into mul + 3 fma; we do the same.
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

FN2_API int fn2_flow_warp_forward_cpu(const float* image, const float* flow, float* warped,
                                      int N, int C, int H, int W, int fill_value) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  if (fill_value != FN2_FILL_ZERO && fill_value != FN2_FILL_NAN) return FN2_ERR_INVALID_ARG;
  const size_t wh = (size_t)W * H, whc = wh * C;
  const float fill = (fill_value == FN2_FILL_ZERO) ? 0.f : u2f(0xFFE00000u);   /* cu:372-375 */
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t off = whc * n;
        const float fx = flow[2 * wh * n + (size_t)y * W + x];                 /* cpp:80 */
        const float fy = flow[2 * wh * n + wh + (size_t)y * W + x];            /* cpp:81 */
        const float x2 = (float)x + fx, y2 = (float)y + fy;
        if (x2 >= 0 && y2 >= 0 && x2 < W && y2 < H) {                           /* cpp:86 */
          const int ixL = (int)x2, iyT = (int)y2;
          const int ixR = imin(ixL + 1, W - 1), iyB = imin(iyT + 1, H - 1);
          const float alpha = x2 - ixL, beta = y2 - iyT;
          const float cTL = (1 - alpha) * (1 - beta), cTR = alpha * (1 - beta);
          const float cBL = (1 - alpha) * beta, cBR = alpha * beta;
          for (int c = 0; c < C; ++c) {
            const float* im = image + off + c * wh;
            const float TL = im[(size_t)iyT * W + ixL], TR = im[(size_t)iyT * W + ixR];
            const float BL = im[(size_t)iyB * W + ixL], BR = im[(size_t)iyB * W + ixR];
            warped[off + c * wh + (size_t)y * W + x] = fmaf(cBR, BR, fmaf(cBL, BL, fmaf(cTR, TR, cTL * TL)));
          }
        } else {
          for (int c = 0; c < C; ++c) warped[off + c * wh + (size_t)y * W + x] = fill;   /* cpp:110-114 */
        }
      }
  return FN2_OK;
}

/* flow_warp_layer.cu:169-229 (GPU backward; memsets at :478-479, propagate_down at :507-508).
 * Sequential scatter here: the reference's atomicAdd order is unspecified. */
FN2_API int fn2_flow_warp_backward_cpu(const float* image, const float* flow, const float* warped_diff,
                                       float* image_diff, float* flow_diff, int N, int C, int H, int W,
                                       int propagate_image, int propagate_flow) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t wh = (size_t)W * H, whc = wh * C;
  memset(image_diff, 0, sizeof(float) * whc * N);
  memset(flow_diff, 0, sizeof(float) * wh * 2 * N);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t off = whc * n;
        const float x2 = (float)x + flow[2 * wh * n + (size_t)y * W + x];
        const float y2 = (float)y + flow[2 * wh * n + wh + (size_t)y * W + x];
        if (x2 >= 0.f && y2 >= 0.f && x2 < W && y2 < H) {
          const int ixL = (int)x2, iyT = (int)y2;
          const int ixR = imin(ixL + 1, W - 1), iyB = imin(iyT + 1, H - 1);
          const float alpha = x2 - ixL, beta = y2 - iyT;
          for (int c = 0; c < C; ++c) {
            const float g = warped_diff[off + c * wh + (size_t)y * W + x];
            float* d = image_diff + off + c * wh;
            d[(size_t)iyT * W + ixL] += g * (1 - alpha) * (1 - beta);          /* cu:197 */
            d[(size_t)iyT * W + ixR] += g * alpha * (1 - beta);                /* cu:198 */
            d[(size_t)iyB * W + ixL] += g * (1 - alpha) * beta;                /* cu:199 */
            d[(size_t)iyB * W + ixR] += g * alpha * beta;                      /* cu:200 */
          }
          float gamma = iyB - y2;                                              /* cu:203 */
          float bot = 0.f;
          for (int c = 0; c < C; ++c) {
            const float* im = image + off + c * wh;
            float temp = 0.f;
            temp += gamma * (im[(size_t)iyT * W + ixR] - im[(size_t)iyT * W + ixL]);
            temp += (1 - gamma) * (im[(size_t)iyB * W + ixR] - im[(size_t)iyB * W + ixL]);
            bot += warped_diff[off + c * wh + (size_t)y * W + x] * temp;
          }
          flow_diff[2 * wh * n + (size_t)y * W + x] = bot;                     /* cu:214 */
          gamma = ixR - x2;                                                     /* cu:216 */
          bot = 0.f;
          for (int c = 0; c < C; ++c) {
            const float* im = image + off + c * wh;
            float temp = 0.f;
            temp += gamma * (im[(size_t)iyB * W + ixL] - im[(size_t)iyT * W + ixL]);
            temp += (1 - gamma) * (im[(size_t)iyB * W + ixR] - im[(size_t)iyT * W + ixR]);
            bot += warped_diff[off + c * wh + (size_t)y * W + x] * temp;
          }
          flow_diff[2 * wh * n + wh + (size_t)y * W + x] = bot;                /* cu:227 */
        }
      }
  if (!propagate_image) memset(image_diff, 0, sizeof(float) * whc * N);
  if (!propagate_flow) memset(flow_diff, 0, sizeof(float) * wh * 2 * N);
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Resample: resample_layer.cu:14-206.
 * ---------------------------------------------------------------------------------------------- */
static inline float bicubic_coeff(float x_) {                 /* :14-20 */
  const float x = fabsf(x_);
  if (x <= 1.0f) return x * x * (1.5f * x - 2.5f) + 1.0f;
  else if (x < 2.0f) return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
  else return 0.0f;
}
static inline float triangle_coeff(float x) {                 /* :28-33 */
  if (-1 <= x && x < 0) return x + 1;
  if (0 <= x && x <= 1) return 1 - x;
  return 0;
}

FN2_API int fn2_resample_forward_cpu(const float* in, float* out, int N, int C, int Hin, int Win,
                                     int Hout, int Wout, int type, int antialias_param) {
  if (N < 0 || C < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return FN2_ERR_INVALID_ARG;
  if (type != FN2_RESAMPLE_NEAREST && type != FN2_RESAMPLE_LINEAR && type != FN2_RESAMPLE_CUBIC)
    return FN2_ERR_UNSUPPORTED;                                                  /* resample_layer.cpp:17-20 */
  const float fx = (float)Win / (float)Wout;                                     /* :146 */
  const float fy = (float)Hin / (float)Hout;                                     /* :147 */
  const int in_cs = Win * Hin, out_cs = Wout * Hout;
  const long nthreads = (long)N * C * out_cs;
  if (type == FN2_RESAMPLE_NEAREST) {
#pragma omp parallel for schedule(static)
    for (long index = 0; index < nthreads; ++index) {
      const int c = (int)(index / out_cs);
      const int x_out = (int)(index % out_cs) % Wout, y_out = (int)(index % out_cs) / Wout;
      const float x_in = x_out * fx + fy / 2.0f - 0.5f;                          /* :117 (sic: fy) */
      const float y_in = y_out * fy + fx / 2.0f - 0.5f;                          /* :118 (sic: fx) */
      int xr = (int)roundf(x_in), yr = (int)roundf(y_in);                        /* :120-121 */
      /* The reference does not clamp (:123) and reads out of bounds when fx != fy pushes the
       * rounded index outside; clamp so the oracle itself stays defined. */
      xr = imin(imax(xr, 0), Win - 1);
      yr = imin(imax(yr, 0), Hin - 1);
      out[index] = in[(size_t)c * in_cs + (size_t)yr * Win + xr];
    }
    return FN2_OK;
  }
  const int cubic = (type == FN2_RESAMPLE_CUBIC);
  const int is_down = (fx > 1) || (fy > 1);                                      /* :179 */
  const int antialias = is_down && antialias_param;                              /* :180 */
  const int kernel_width = cubic ? 4 : 2;                                        /* :182-185 */
#pragma omp parallel for schedule(static)
  for (long index = 0; index < nthreads; ++index) {
    const int c = (int)(index / out_cs);
    const int x_out = (int)(index % out_cs) % Wout, y_out = (int)(index % out_cs) / Wout;
    const float x_in = x_out * fx + fy / 2.0f - 0.5f;                            /* :62 */
    const float y_in = y_out * fy + fx / 2.0f - 0.5f;                            /* :63 */
    const int xr = (int)roundf(x_in), yr = (int)roundf(y_in);                    /* :65-66 */
    float sum = 0, wsum = 0;
    const float ax = 1.0f / (antialias ? fx : 1.0f);                             /* :71 */
    const float ay = 1.0f / (antialias ? fy : 1.0f);                             /* :72 */
    const int rx = (fx < 1.0f) ? 2 : (int)ceilf((float)kernel_width / ax);       /* :73 */
    const int ry = (fy < 1.0f) ? 2 : (int)ceilf((float)kernel_width / ay);       /* :74 */
    for (int y = yr - ry; y <= yr + ry; ++y)
      for (int x = xr - rx; x <= xr + rx; ++x) {
        if (y < 0 || x < 0) continue;
        if (y >= Hin || x >= Win) continue;
        const float dx = x_in - x, dy = y_in - y;
        float w;
        if (cubic) w = ax * bicubic_coeff(ax * dx) * ay * bicubic_coeff(ay * dy);      /* :87 */
        else w = ax * triangle_coeff(ax * dx) * ay * triangle_coeff(ay * dy);          /* :89 */
        sum = fmaf(w, in[(size_t)c * in_cs + (size_t)y * Win + x], sum);                /* :90 */
        wsum += w;
      }
    out[index] = (!wsum) ? 0 : (sum / wsum);                                      /* :93 */
  }
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * L1Loss: l1loss_layer.cpp:11-90 (composition) + l1loss_layer.cu:67-190.
 * Sub-layers: Eltwise SUM coeff (+1,-1) (eltwise_layer.cpp:59-65), Power^2, 1x1 Convolution with a
 * constant filler (sum over channels), Power^0.5 with shift epsilon (power_layer.cu:9-30, :33-83).
 * The reduction is a cublasSdot in the reference (order unspecified); we accumulate in double.
 * ---------------------------------------------------------------------------------------------- */
FN2_API int fn2_l1loss_forward_cpu(const fn2_l1loss_params* p, const float* b0, const float* b1,
                                   float* loss_out, float* normalize_coeff_out,
                                   int N, int C, int H, int W) {
  if (!p || N < 1 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W, count = (size_t)N * C * hw;
  double nvalid = 0;
#pragma omp parallel for reduction(+ : nvalid) schedule(static)
  for (size_t i = 0; i < count; ++i) {
    const float d = b1 ? (b0[i] - b1[i]) : b0[i];
    nvalid += (d == d) ? 1.0 : 0.0;                                /* FindNotNaNs cu:20-24; dot(mask,mask) cu:87 */
  }
  float norm;
  if (p->normalize_by_num_entries) norm = (float)nvalid / (float)C;      /* cu:86-88 */
  else norm = (float)N;                                                  /* cu:90 */
  double dot = 0;
  if (p->l2_per_location) {
    const float wgt = p->l2_prescale_by_channels ? 1.f / (float)C : 1.f;    /* cpp:47-51 */
    const float plat2 = p->plateau * p->plateau;                             /* cu:104 */
#pragma omp parallel for reduction(+ : dot) schedule(static)
    for (size_t q = 0; q < (size_t)N * hw; ++q) {
      const size_t n = q / hw, s = q % hw;
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        const size_t i = (n * C + c) * hw + s;
        float d = b1 ? (b0[i] - b1[i]) : b0[i];
        d = (d == d) ? d : 0.f;                                     /* KillMasked cu:95-96 */
        acc += wgt * (d * d);                                       /* square cu:99, 1x1 conv cu:100 */
      }
      if (p->plateau > 0 && fabsf(acc) < plat2) acc = 0.f;          /* cu:103-114 */
      dot += (double)sqrtf(acc + p->epsilon);                       /* Power(0.5, shift eps) cu:117; dot with ones cu:119 */
    }
  } else {
#pragma omp parallel for reduction(+ : dot) schedule(static)
    for (size_t i = 0; i < count; ++i) {
      float d = b1 ? (b0[i] - b1[i]) : b0[i];
      int keep = (d == d);
      if (p->plateau > 0 && fabsf(d) < p->plateau) keep = 0;        /* MaskPlateauValues cu:52-56,123-126 (NaN: fabs(NaN)<p false) */
      d = keep ? d : 0.f;                                           /* KillMasked cu:132-134 */
      const float sign = d > 0 ? 1.f : -1.f;                        /* ComputeSign cu:11-15 */
      dot += (double)(d * sign);                                    /* cu:139 */
    }
  }
  if (loss_out) *loss_out = (float)dot / norm;                      /* cu:141 */
  if (normalize_coeff_out) *normalize_coeff_out = norm;
  return FN2_OK;
}

FN2_API int fn2_l1loss_backward_cpu(const fn2_l1loss_params* p, const float* b0, const float* b1,
                                    float top_diff, float normalize_coeff,
                                    float* b0_diff, float* b1_diff, int N, int C, int H, int W) {
  if (!p || N < 1 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  const float alpha = top_diff / normalize_coeff;                   /* cu:155 */
  if (p->l2_per_location) {
    const float wgt = p->l2_prescale_by_channels ? 1.f / (float)C : 1.f;
    const float plat2 = p->plateau * p->plateau;
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < (size_t)N * hw; ++q) {
      const size_t n = q / hw, s = q % hw;
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        const size_t i = (n * C + c) * hw + s;
        float d = b1 ? (b0[i] - b1[i]) : b0[i];
        d = (d == d) ? d : 0.f;
        acc += wgt * (d * d);
      }
      int plateau_kill = 0;
      if (p->plateau > 0 && fabsf(acc) < plat2) { acc = 0.f; plateau_kill = 1; }
      const float e = sqrtf(acc + p->epsilon);
      /* sqrt_output diff = alpha * 1 (cu:158); Power backward general branch (power_layer.cu:62-74):
       * diff = top_data / (x + shift) * 0.5 * top_diff */
      float ds = (e / (acc + p->epsilon)) * 0.5f * alpha;
      if (plateau_kill) ds = 0.f;                                   /* cu:162-166 */
      for (int c = 0; c < C; ++c) {
        const size_t i = (n * C + c) * hw + s;
        float d = b1 ? (b0[i] - b1[i]) : b0[i];
        const int valid = (d == d);
        d = valid ? d : 0.f;
        /* conv backward: wgt * ds; square backward (power_layer.cu:48-52): 2 * d * that */
        float g = (2.f * d) * (wgt * ds);
        g = valid ? g : 0.f;                                        /* KillMasked cu:179-180 */
        b0_diff[i] = g;                                             /* Eltwise backward coeff +1 */
        if (b1 && b1_diff) b1_diff[i] = -g;                         /* coeff -1 */
      }
    }
  } else {
    const size_t count = (size_t)N * C * hw;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < count; ++i) {
      float d = b1 ? (b0[i] - b1[i]) : b0[i];
      int keep = (d == d);
      if (p->plateau > 0 && fabsf(d) < p->plateau) keep = 0;
      d = keep ? d : 0.f;
      const float sign = d > 0 ? 1.f : -1.f;
      float g = alpha * sign;                                       /* cu:175-176 */
      g = keep ? g : 0.f;                                           /* cu:179-180 */
      b0_diff[i] = g;
      if (b1 && b1_diff) b1_diff[i] = -g;
    }
  }
  return FN2_OK;
}

/* Every L1Loss layer of a net at once (fn2_l1loss_forward_multi / _backward_multi): the layers one by one, then
 * Net::ForwardFromTo's  loss += layer_loss  (net.cpp:565-579) with layer_loss = caffe_cpu_dot(top_data, loss_weight) (layer.hpp:434-440)
 * -- a float product and a float sum per layer, in layer order.  norms[s] = the layer's normalize_coeff (what the HIP path keeps in
 * its workspace between forward and backward). */
FN2_API int fn2_l1loss_forward_multi_cpu(const fn2_l1loss_params* p, int nscales, const fn2_l1loss_scale* sc, float* losses, float* norms,
                                         float* total) {
  if (!p || !sc || nscales < 1) return FN2_ERR_INVALID_ARG;
  volatile float t = 0.f;
  for (int s = 0; s < nscales; ++s) {
    float loss = 0.f, norm = 0.f;
    const int rc = fn2_l1loss_forward_cpu(p, sc[s].bottom0, sc[s].bottom1, &loss, &norm, sc[s].N, sc[s].C, sc[s].H, sc[s].W);
    if (rc) return rc;
    if (losses) losses[s] = loss;
    if (norms) norms[s] = norm;
    volatile float prod = sc[s].loss_weight * loss;      /* rounded to float before the sum (no fused multiply-add) */
    t = t + prod;
  }
  if (total) *total = t;
  return FN2_OK;
}

FN2_API int fn2_l1loss_backward_multi_cpu(const fn2_l1loss_params* p, int nscales, const fn2_l1loss_scale* sc, float total_diff,
                                          const float* norms) {
  if (!p || !sc || !norms || nscales < 1) return FN2_ERR_INVALID_ARG;
  for (int s = 0; s < nscales; ++s) {
    volatile float top_diff = sc[s].loss_weight * total_diff;        /* the layer's top[0]->cpu_diff()[0] (l1loss_layer.cu:155) */
    const int rc = fn2_l1loss_backward_cpu(p, sc[s].bottom0, sc[s].bottom1, top_diff, norms[s], sc[s].bottom0_diff,
                                           sc[s].bottom1 ? sc[s].bottom1_diff : 0, sc[s].N, sc[s].C, sc[s].H, sc[s].W);
    if (rc) return rc;
  }
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * ChannelNorm: channel_norm_layer.cpp:43-69 / .cu:16-47.
 * ---------------------------------------------------------------------------------------------- */
FN2_API int fn2_channel_norm_forward_cpu(const float* bottom, float* top, int N, int C, int H, int W) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
#pragma omp parallel for schedule(static)
  for (size_t q = 0; q < (size_t)N * hw; ++q) {
    const size_t n = q / hw, s = q % hw;
    float norm = 0;
    for (int c = 0; c < C; ++c) {
      const float v = bottom[(n * C + c) * hw + s];
      norm = fmaf(v, v, norm);                                       /* cu:27-28 */
    }
    top[q] = sqrtf(norm);                                            /* cu:31-32 */
  }
  return FN2_OK;
}

FN2_API int fn2_channel_norm_backward_cpu(const float* bottom, const float* top, const float* top_diff,
                                          float* bottom_diff, int N, int C, int H, int W) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < (size_t)N * C * hw; ++i) {
    const size_t n = i / (C * hw), s = i % hw;
    /* cu:45: float * float / (float + 1e-9 [double]) -> evaluated in double, rounded on store */
    bottom_diff[i] = (float)((double)(top_diff[n * hw + s] * bottom[i]) / ((double)top[n * hw + s] + 1e-9));
  }
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Downsample: downsample_layer.cu:15-72, host :75-129.
 * ---------------------------------------------------------------------------------------------- */
FN2_API int fn2_downsample_forward_cpu(const float* bottom, float* top, int N, int C, int Hin, int Win,
                                       int Hout, int Wout) {
  if (N < 0 || C < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return FN2_ERR_INVALID_ARG;
  if (Hin == Hout && Win == Wout) {                                  /* downsample_layer.cpp:53-56 shares the blob */
    memcpy(top, bottom, sizeof(float) * (size_t)N * C * Hin * Win);
    return FN2_OK;
  }
  if (Hout < 2 || Wout < 2) return FN2_ERR_INVALID_ARG;              /* :104-105 divide by size - 1: inf scale, undefined radius */
  const float widthScale = (float)(Win - 1) / (float)(Wout - 1);      /* :104 */
  const float heightScale = (float)(Hin - 1) / (float)(Hout - 1);     /* :105 */
  const int wradius = (int)ceilf(widthScale), hradius = (int)ceilf(heightScale);   /* :107-108 */
  const long topcount = (long)N * C * Hout * Wout;
#pragma omp parallel for schedule(static)
  for (long index = 0; index < topcount; ++index) {
    const int destx = (int)(index % Wout), desty = (int)((index / Wout) % Hout);
    const long cn = index / Wout / Hout;
    const float botx = ((float)destx / (float)(Wout - 1)) * (float)(Win - 1);      /* :27 */
    const float boty = ((float)desty / (float)(Hout - 1)) * (float)(Hin - 1);      /* :28 */
    const int ibotx = (int)roundf(botx), iboty = (int)roundf(boty);                /* :30-31 */
    const float* src = bottom + (size_t)cn * Hin * Win;
    float accum_value = 0, accum_weight = 0, accum_nan = 0;
    for (int yoff = -hradius; yoff <= hradius; ++yoff) {
      const int by = iboty + yoff;
      for (int xoff = -wradius; xoff <= wradius; ++xoff) {
        const int bx = ibotx + xoff;
        if (bx >= 0 && by >= 0 && bx < Win && by < Hin) {
          float sample = src[(size_t)by * Win + bx];
          float weight = fmaxf(0.0f, 1.0f - (fabsf((float)bx - botx) / widthScale)) *
                         fmaxf(0.0f, 1.0f - (fabsf((float)by - boty) / heightScale));   /* :52 */
          if (sample != sample) { accum_nan += weight; sample = 0; weight = 0; }       /* :53-57 */
          accum_value = fmaf(sample, weight, accum_value);                              /* :59 */
          accum_weight += weight;                                                       /* :60 */
        }
      }
    }
    if (accum_nan / accum_weight > 0.5f) top[index] = u2f(0x7fffffffu);               /* :64-65 */
    else top[index] = accum_value / accum_weight;                                      /* :67 */
  }
  return FN2_OK;
}

/* Several top sizes of one bottom (fn2_downsample_forward_multi: the ground-truth pyramid of the multi-scale loss in one launch): the layers
 * one after another. */
FN2_API int fn2_downsample_forward_multi_cpu(const float* bottom, float* const* tops, const int* top_heights, const int* top_widths, int count,
                                             int N, int C, int Hin, int Win) {
  if (count < 1 || count > 8 || !tops || !top_heights || !top_widths) return FN2_ERR_INVALID_ARG;
  for (int j = 0; j < count; ++j) {
    if (top_heights[j] < 2 || top_widths[j] < 2 || (top_heights[j] == Hin && top_widths[j] == Win)) return FN2_ERR_INVALID_ARG;
    const int rc = fn2_downsample_forward_cpu(bottom, tops[j], N, C, Hin, Win, top_heights[j], top_widths[j]);
    if (rc) return rc;
  }
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Flow heads: stock Caffe Convolution / Deconvolution arithmetic (conv_layer.cpp:8-40 via im2col + GEMM,
 * deconv_layer.cpp:8-45), restated as direct loops.  The reference sums through cblas_sgemm / cublasSgemm
 * (order unspecified); tests compare at fp32 tolerance against torch-CPU conv2d / conv_transpose2d too.
 * ---------------------------------------------------------------------------------------------- */
FN2_API int fn2_predict_flow_conv_forward_cpu(const float* in, const float* weight, const float* bias, float* out,
                                              int N, int C, int H, int W) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t plane = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < 2; ++o)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          double acc = bias ? bias[o] : 0.0;
          for (int c = 0; c < C; ++c)
            for (int dy = 0; dy < 3; ++dy)
              for (int dx = 0; dx < 3; ++dx) {
                const int yy = y + dy - 1, xx = x + dx - 1;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                acc += (double)weight[(((size_t)o * C + c) * 3 + dy) * 3 + dx] * in[((size_t)n * C + c) * plane + (size_t)yy * W + xx];
              }
          out[((size_t)n * 2 + o) * plane + (size_t)y * W + x] = (float)acc;
        }
  return FN2_OK;
}

FN2_API int fn2_upsample_flow_deconv_forward_cpu(const float* in, const float* weight, const float* bias, float* out,
                                                 int N, int H, int W) {
  if (N < 0 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const int Ho = 2 * H, Wo = 2 * W;
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < 2; ++o)
      for (int Y = 0; Y < Ho; ++Y)
        for (int X = 0; X < Wo; ++X) {
          double acc = bias ? bias[o] : 0.0;
          for (int c = 0; c < 2; ++c)
            for (int ky = 0; ky < 4; ++ky)
              for (int kx = 0; kx < 4; ++kx) {
                const int ny = Y + 1 - ky, nx = X + 1 - kx;       /* Y = 2*iy - pad + ky */
                if (ny < 0 || nx < 0 || (ny & 1) || (nx & 1)) continue;
                const int iy = ny / 2, ix = nx / 2;
                if (iy >= H || ix >= W) continue;
                acc += (double)weight[((c * 2 + o) * 4 + ky) * 4 + kx] * in[(((size_t)n * 2 + c) * H + iy) * W + ix];
              }
          out[(((size_t)n * 2 + o) * Ho + Y) * Wo + X] = (float)acc;
        }
  return FN2_OK;
}

/* the same layer into channels [top_c0, top_c0 + 2) of a wider top blob (the last input of a refinement Concat) */
FN2_API int fn2_upsample_flow_deconv_forward_into_cpu(const float* in, const float* weight, const float* bias, float* top,
                                                      int N, int H, int W, int top_channels, int top_c0) {
  if (N < 0 || H < 1 || W < 1 || top_c0 < 0 || top_c0 + 2 > top_channels) return FN2_ERR_INVALID_ARG;
  const size_t plane = (size_t)4 * H * W;
  float* tmp = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * 2 * plane);
  if (!tmp) return FN2_ERR_INVALID_ARG;
  const int rc = fn2_upsample_flow_deconv_forward_cpu(in, weight, bias, tmp, N, H, W);
  if (rc == FN2_OK)
    for (int n = 0; n < N; ++n)
      memcpy(top + ((size_t)n * top_channels + top_c0) * plane, tmp + (size_t)n * 2 * plane, sizeof(float) * 2 * plane);
  free(tmp);
  return rc;
}

/* Convolution bias term + in-place leaky ReLU (base_conv_layer.cpp:343-348 forward_gpu_bias: top += bias[c];
 * relu_layer.cu:8-14 ReLUForward: out = in > 0 ? in : in * negative_slope), fp32 like the reference. */
FN2_API int fn2_bias_leaky_relu_forward_cpu(float* data, const float* bias, int N, int C, int H, int W, float negative_slope) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
#pragma omp parallel for
  for (long long p = 0; p < (long long)N * C; ++p) {
    const float b = bias ? bias[p % C] : 0.f;
    float* d = data + (size_t)p * hw;
    for (size_t i = 0; i < hw; ++i) {
      const float t = d[i] + b;
      d[i] = t > 0.f ? t : t * negative_slope;
    }
  }
  return FN2_OK;
}

/* conv1 + ReLU1 of the FlowNet encoders: Convolution{kernel_size 7, stride 2, pad 3} (conv_layer.cpp:25-40 /
 * base_conv_layer.cpp:255-318: im2col + GEMM + bias) and the in-place ReLU with negative_slope (relu_layer.cpp:23-30).
 * Direct loops; the contraction is accumulated in double (the reference's SGEMM order is library-defined). */
FN2_API int fn2_conv_k7s2_relu_forward_cpu(const float* in, const float* weight, const float* bias, float* out,
                                           int N, int Cin, int Hin, int Win, int Cout, float negative_slope) {
  if (N < 0 || Cin < 1 || Hin < 1 || Win < 1 || Cout < 1) return FN2_ERR_INVALID_ARG;
  const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
          double acc = bias ? bias[co] : 0.0;
          for (int c = 0; c < Cin; ++c)
            for (int ky = 0; ky < 7; ++ky) {
              const int yi = 2 * y - 3 + ky;
              if (yi < 0 || yi >= Hin) continue;
              for (int kx = 0; kx < 7; ++kx) {
                const int xi = 2 * x - 3 + kx;
                if (xi < 0 || xi >= Win) continue;
                acc += (double)weight[((co * Cin + c) * 7 + ky) * 7 + kx] * in[(((size_t)n * Cin + c) * Hin + yi) * Win + xi];
              }
            }
          const float t = (float)acc;
          out[(((size_t)n * Cout + co) * Ho + y) * Wo + x] = t > 0.f ? t : t * negative_slope;
        }
  return FN2_OK;
}


/* Correlation followed by the in-place ReLU (relu_layer.cu:8-27) and written as a channel slice of a wider blob (the Concat that
 * follows it, concat_layer.cu:8-52): the checker of fn2_correlation_forward_fused, composed from the plain forward above. */
FN2_API int fn2_correlation_forward_fused_cpu(const fn2_corr_params* p, const float* bottom0, const float* bottom1, float* top,
                                              int N, int C, int H, int W, int top_channels, int top_c0, int relu, float negative_slope) {
  int tc, th, tw;
  int rc = fn2_correlation_out_shape_cpu(p, C, H, W, &tc, &th, &tw);
  if (rc) return rc;
  if (top_channels <= 0) { top_channels = tc; top_c0 = 0; }
  if (top_c0 < 0 || top_c0 + tc > top_channels) return FN2_ERR_INVALID_ARG;
  const size_t per = (size_t)tc * th * tw;
  float* tmp = (float*)malloc(sizeof(float) * per * (size_t)(N > 0 ? N : 1));
  if (!tmp) return FN2_ERR_WORKSPACE;
  rc = fn2_correlation_forward_cpu(p, bottom0, bottom1, tmp, N, C, H, W);
  if (!rc)
    for (int n = 0; n < N; ++n)
      for (size_t i = 0; i < per; ++i) {
        float v = tmp[(size_t)n * per + i];
        if (relu) v = v > 0.f ? v : v * negative_slope;
        top[((size_t)n * top_channels + top_c0) * th * tw + i] = v;
      }
  free(tmp);
  return rc;
}

/* Direct convolution + bias + optional ReLU on PACKED weights: the CPU twin of csrc/conv_mfma.hip.
 * Reference arithmetic: Convolution{kernel_size, stride, pad} (conv_layer.cpp:25-40 / base_conv_layer.cpp:255-318: im2col + GEMM +
 * bias) and the in-place ReLU (relu_layer.cpp:23-30).  The reference's SGEMM summation order is library-defined; the HIP kernel
 * accumulates in the order (channel quad, ky, kx, channel within the quad) with fused multiply-adds (v_mfma_f32_16x16x4_f32 is a
 * k-ordered fma chain), zero padding included as 0-products -- restated here with fmaf in the same order, so the two agree bit for bit.
 * Packed layout (fn2_conv_mfma_pack_weights): [Cout/64][k-steps + 8 spare][lane 64][4], lane = 16 * kq + co, element j:
 * W[64 g + 16 j + co][4 cq + kq][ky][kx], k-step = (cq * k + ky) * k + kx; zero beyond Cin. */
static int conv_mfma_ksteps(int Cin, int k) {      /* channel quads padded to whole chunks: 2 quads, 8 for 1x1 kernels (csrc/conv_mfma.hip) */
  const int cq = k == 1 ? 8 : 2;
  return (((Cin + 3) / 4 + cq - 1) / cq) * cq * k * k;
}

FN2_API size_t fn2_conv_mfma_packed_floats_cpu(int Cout, int Cin, int kernel) {
  if (Cout <= 0 || Cout % 32 != 0 || Cin <= 0 || kernel <= 0) return 0;
  return (size_t)((Cout + 63) / 64) * (conv_mfma_ksteps(Cin, kernel) + 8) * 256;
}

FN2_API int fn2_conv_mfma_pack_weights_cpu(const float* weight, float* packed, int Cout, int Cin, int kernel) {
  if (!weight || !packed || Cout <= 0 || Cout % 32 != 0 || Cin <= 0 || (kernel != 1 && kernel != 3 && kernel != 4 && kernel != 5 && kernel != 7)) return FN2_ERR_INVALID_ARG;
  const int ksteps = conv_mfma_ksteps(Cin, kernel), kalloc = ksteps + 8, kk = kernel * kernel;
  for (int g = 0; g < (Cout + 63) / 64; ++g)
    for (int ks = 0; ks < kalloc; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int co = 64 * g + 16 * j + (lane & 15), ci = 4 * (ks / kk) + (lane >> 4), tap = ks % kk;
          packed[(((size_t)g * kalloc + ks) * 64 + lane) * 4 + j] =
              (ks < ksteps && ci < Cin && co < Cout) ? weight[((size_t)co * Cin + ci) * kk + tap] : 0.f;
        }
  return FN2_OK;
}

/* fn2_conv_mfma_pack_weights_view: the operand of the [Cout][Cin][k][k] VIEW of a blob -- element (co, ci, tap) =
 * weight[co * stride_cout + ci * stride_cin + (flip ? k*k - 1 - tap : tap)] for co < src_cout and ci < src_cin, 0 beyond. */
FN2_API int fn2_conv_mfma_pack_weights_view_cpu(const float* weight, float* packed, int Cout, int Cin, int kernel, int src_cout, int src_cin,
                                                long long stride_cout, long long stride_cin, int flip) {
  if (!weight || !packed || Cout <= 0 || Cout % 32 != 0 || Cin <= 0 || (kernel != 1 && kernel != 3 && kernel != 4 && kernel != 5 && kernel != 7)) return FN2_ERR_INVALID_ARG;
  if (src_cout < 1 || src_cout > Cout || src_cin < 1 || src_cin > Cin || stride_cout < 1 || stride_cin < 1) return FN2_ERR_INVALID_ARG;
  const int ksteps = conv_mfma_ksteps(Cin, kernel), kalloc = ksteps + 8, kk = kernel * kernel;
  for (int g = 0; g < (Cout + 63) / 64; ++g)
    for (int ks = 0; ks < kalloc; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int co = 64 * g + 16 * j + (lane & 15), ci = 4 * (ks / kk) + (lane >> 4), tap = ks % kk;
          packed[(((size_t)g * kalloc + ks) * 64 + lane) * 4 + j] =
              (ks < ksteps && ci < src_cin && co < src_cout) ? weight[(size_t)co * stride_cout + (size_t)ci * stride_cin + (flip ? kk - 1 - tap : tap)] : 0.f;
        }
  return FN2_OK;
}

FN2_API int fn2_conv_mfma_forward_cpu(const float* bottom, const float* packed, const float* bias, float* top,
                                      int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                      int Cout, int top_channels, int top_c0, int kernel, int stride, int pad,
                                      int relu, float negative_slope) {
  if (N < 0 || Cin < 1 || Hin < 1 || Win < 1 || Cout < 1 || Cout % 32 != 0 || kernel < 1 || stride < 1 || pad < 0) return FN2_ERR_INVALID_ARG;
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels) return FN2_ERR_INVALID_ARG;
  const int Ho = (Hin + 2 * pad - kernel) / stride + 1, Wo = (Win + 2 * pad - kernel) / stride + 1;
  const int quads = (Cin + 3) / 4, kalloc = conv_mfma_ksteps(Cin, kernel) + 8;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co) {
      const float* wg = packed + (size_t)(co / 64) * kalloc * 256 + ((co % 64) / 16) + 4 * (co % 16);
      for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
          float acc = 0.f;
          for (int cq = 0; cq < quads; ++cq)
            for (int ky = 0; ky < kernel; ++ky)
              for (int kx = 0; kx < kernel; ++kx) {
                const int ks = (cq * kernel + ky) * kernel + kx;
                const int yi = stride * y - pad + ky, xi = stride * x - pad + kx;
                for (int kq = 0; kq < 4; ++kq) {
                  const int ci = 4 * cq + kq;
                  float v = 0.f;
                  if (ci < Cin && yi >= 0 && yi < Hin && xi >= 0 && xi < Win)
                    v = bottom[(((size_t)n * bottom_channels + bottom_c0 + ci) * Hin + yi) * Win + xi];
                  acc = fmaf(v, wg[(size_t)ks * 256 + 64 * kq], acc);
                }
              }
          float t = acc + (bias ? bias[co] : 0.f);
          if (relu) t = t > 0.f ? t : t * negative_slope;
          top[(((size_t)n * top_channels + top_c0 + co) * Ho + y) * Wo + x] = t;
        }
    }
  return FN2_OK;
}

/* 3x3 convolution for small feature maps with the channel axis split into `ksplit` parts: the CPU twin of csrc/conv_plane.hip (MODE 0).
 * Same reference arithmetic and packed weights as fn2_conv_mfma_forward_cpu (kernel 3); per part a k-ordered fmaf chain over its
 * channel quads (channel quad, ky, kx, channel within the quad), then the parts added in part order, then bias and ReLU.  Part p
 * covers the 2-quad units [p U / ksplit, (p + 1) U / ksplit), U = ceil(quads / 2).
 * ksplit is what fn2_conv_plane_ksplit() reports for the layer (passed in: the oracle does not link the HIP library). */
FN2_API int fn2_conv_plane_k_forward_cpu(const float* bottom, const float* packed, const float* bias, float* top,
                                         int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                         int Cout, int top_channels, int top_c0, int kernel, int stride, int pad,
                                         int relu, float negative_slope, int ksplit) {
  if (N < 0 || Cin < 1 || Cin % 4 != 0 || Hin < 1 || Win < 1 || Cout < 1 || Cout % 64 != 0 || stride < 1 || pad < 0 || ksplit < 1) return FN2_ERR_INVALID_ARG;
  if (kernel != 3 && kernel != 4 && kernel != 5) return FN2_ERR_INVALID_ARG;
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels) return FN2_ERR_INVALID_ARG;
  const int quads = Cin / 4, units = (quads + 1) / 2, kalloc = conv_mfma_ksteps(Cin, kernel) + 8;
  if (ksplit > units) return FN2_ERR_INVALID_ARG;
  const int Ho = (Hin + 2 * pad - kernel) / stride + 1, Wo = (Win + 2 * pad - kernel) / stride + 1;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co) {
      const float* wg = packed + (size_t)(co / 64) * kalloc * 256 + ((co % 64) / 16) + 4 * (co % 16);
      for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
          float sum = 0.f;
          for (int part = 0; part < ksplit; ++part) {
            float acc = 0.f;
            const int q0 = 2 * (int)((long long)part * units / ksplit), q1 = 2 * (int)((long long)(part + 1) * units / ksplit);
            for (int cq = q0; cq < q1; ++cq)
              for (int ky = 0; ky < kernel; ++ky)
                for (int kx = 0; kx < kernel; ++kx) {
                  const int ks = (cq * kernel + ky) * kernel + kx;
                  const int yi = stride * y - pad + ky, xi = stride * x - pad + kx;
                  for (int kq = 0; kq < 4; ++kq) {
                    float v = 0.f;
                    if (4 * cq + kq < Cin && yi >= 0 && yi < Hin && xi >= 0 && xi < Win)
                      v = bottom[(((size_t)n * bottom_channels + bottom_c0 + 4 * cq + kq) * Hin + yi) * Win + xi];
                    acc = fmaf(v, wg[(size_t)ks * 256 + 64 * kq], acc);
                  }
                }
            sum = part == 0 ? acc : sum + acc;
          }
          float t = sum + (bias ? bias[co] : 0.f);
          if (relu) t = t > 0.f ? t : t * negative_slope;
          top[(((size_t)n * top_channels + top_c0 + co) * Ho + y) * Wo + x] = t;
        }
    }
  return FN2_OK;
}

FN2_API int fn2_conv_plane_forward_cpu(const float* bottom, const float* packed, const float* bias, float* top,
                                       int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                       int Cout, int top_channels, int top_c0, int stride, int pad,
                                       int relu, float negative_slope, int ksplit) {
  return fn2_conv_plane_k_forward_cpu(bottom, packed, bias, top, N, Cin, Hin, Win, bottom_channels, bottom_c0, Cout, top_channels, top_c0,
                                      3, stride, pad, relu, negative_slope, ksplit);
}

/* CPU twin of fn2_conv_k7s2_wgrad (csrc/conv_stem_wgrad.hip): weight gradient of the 7x7 / 2 / 3 stem convolution,
 * ConvolutionLayer::Backward_gpu -> weight_gpu_gemm (conv_layer.cu:40-52, base_conv_layer.cpp:368-384), in the kernel's summation order:
 * units (sample, pair of output rows, 32-pixel x segment) in order; part p covers units [p U / parts, (p + 1) U / parts) and is one fmaf
 * chain over its pixels in (unit, row, x) order -- pixels beyond the map count as zeros, like the kernel's zero-filled tile --; the parts
 * are added in 16 contiguous segments (part by part inside a segment, then the segment sums in order: stem_wgrad_finalize).
 * parts = fn2_conv_k7s2_wgrad_ksplit() (passed in: the oracle does not link the HIP library). */
FN2_API int fn2_conv_k7s2_wgrad_cpu(const float* top_diff, const float* bottom, float* weight_diff, int N, int Cin, int Hin, int Win, int Cout,
                                    int accumulate, int parts) {
  if (!top_diff || !bottom || !weight_diff || N < 1 || Cin < 1 || Hin < 1 || Win < 1 || Cout < 1 || parts < 1) return FN2_ERR_INVALID_ARG;
  const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1, R = 2, XT = 32;
  const int nyb = (Ho + R - 1) / R, nsx = (Wo + XT - 1) / XT, units = N * nyb * nsx, taps = Cin * 49;
  if (parts > units) return FN2_ERR_INVALID_ARG;
#pragma omp parallel for collapse(2)
  for (int co = 0; co < Cout; ++co)
    for (int t = 0; t < taps; ++t) {
      const int ci = t / 49, ky = (t % 49) / 7, kx = t % 7;
      float sum = 0.f;
      int first_seg = 1;
      for (int seg = 0; seg < 16; ++seg) {            /* 16 contiguous segments of parts: part by part inside, then the segment sums in order */
        const int p0 = (int)((long long)seg * parts / 16), p1 = (int)((long long)(seg + 1) * parts / 16);
        if (p0 >= p1) continue;
        float ssum = 0.f;
        for (int part = p0; part < p1; ++part) {
          const int u0 = (int)((long long)part * units / parts), u1 = (int)((long long)(part + 1) * units / parts);
          float acc = 0.f;
          for (int u = u0; u < u1; ++u) {
            const int sx = u % nsx, yb = (u / nsx) % nyb, n = u / (nsx * nyb);
            for (int r = 0; r < R; ++r)
              for (int xx = 0; xx < XT; ++xx) {
                const int y = R * yb + r, x = XT * sx + xx;
                const float d = (y < Ho && x < Wo) ? top_diff[(((size_t)n * Cout + co) * Ho + y) * Wo + x] : 0.f;
                const int by = 2 * y - 3 + ky, bx = 2 * x - 3 + kx;
                const float b = (by >= 0 && by < Hin && bx >= 0 && bx < Win) ? bottom[(((size_t)n * Cin + ci) * Hin + by) * Win + bx] : 0.f;
                acc = fmaf(d, b, acc);
              }
          }
          ssum = part == p0 ? acc : ssum + acc;
        }
        sum = first_seg ? ssum : sum + ssum;
        first_seg = 0;
      }
      float* o = weight_diff + (size_t)co * taps + t;
      *o = accumulate ? *o + sum : sum;
    }
  return FN2_OK;
}

/* Deploy head: top[n, top_c0 + c] = bottom[n, c] * scale + shift[c], product and sum rounded separately -- Eltwise{coeff}
 * (eltwise_layer.cpp:59-65: caffe_set(0) + caffe_axpy(coeff, bottom, top)) then the mean subtraction of the deploy-time
 * DataAugmentation layer (data_augmentation_layer.cu:592-621).  CPU twin of fn2_scale_shift_forward. */
FN2_API int fn2_scale_shift_forward_cpu(const float* bottom, float* top, const float* shift, int N, int C, int H, int W,
                                        int top_channels, int top_c0, float scale) {
  if (!bottom || !top || N < 0 || C < 1 || H < 1 || W < 1 || top_c0 < 0 || top_c0 + C > top_channels) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float* p = bottom + ((size_t)n * C + c) * hw;
      float* q = top + ((size_t)n * top_channels + top_c0 + c) * hw;
      const float sh = shift ? shift[c] : 0.f;
      for (size_t i = 0; i < hw; ++i) {
        volatile float prod = p[i] * scale;            /* volatile: the compiler must not contract the pair into an fma */
        q[i] = prod + sh;
      }
    }
  return FN2_OK;
}

/* Transposed convolution, stride 2 (+ bias + optional ReLU): the CPU twin of csrc/tconv_mfma.hip.
 * Reference arithmetic: DeconvolutionLayer::Forward_cpu (deconv_layer.cpp:8-26: weight^T x bottom, col2im, bias) / the data gradient of
 * ConvolutionLayer::Backward_cpu (conv_layer.cpp:57-62 -> backward_cpu_gemm, base_conv_layer.cpp:305-317):
 *     top[n][co][Y][X] = act(bias[co] + sum_{ci,ky,kx: Y = 2y - pad + ky, X = 2x - pad + kx} bottom[n][ci][y][x] W[ci][co][ky][kx]).
 * Summation order of the HIP kernel per output element: channel quads ascending; within a quad the taps ky (== Y + pad mod 2) and kx
 * (== X + pad mod 2) ascending; within a tap the 4 channels of the quad, with fmaf.  weight is the UNPACKED blob [Cin][Cout][k][k]. */
FN2_API int fn2_tconv_forward_cpu(const float* bottom, const float* weight, const float* bias, float* top,
                                  int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                  int Cout, int Hout, int Wout, int top_channels, int top_c0, int kernel, int pad,
                                  int relu, float negative_slope) {
  if (!bottom || !weight || !top || N < 0 || Cin < 1 || Hin < 1 || Win < 1 || Cout < 1 || Hout < 1 || Wout < 1 || kernel < 1 || pad < 0) return FN2_ERR_INVALID_ARG;
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels) return FN2_ERR_INVALID_ARG;
  const int quads = (Cin + 3) / 4;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int Y = 0; Y < Hout; ++Y)
        for (int X = 0; X < Wout; ++X) {
          float acc = 0.f;
          for (int cq = 0; cq < quads; ++cq)
            for (int ky = (Y + pad) & 1; ky < kernel; ky += 2)
              for (int kx = (X + pad) & 1; kx < kernel; kx += 2) {
                const int y2 = Y + pad - ky, x2 = X + pad - kx;          /* = 2 y, 2 x (even by construction) */
                if (y2 < 0 || x2 < 0 || (y2 >> 1) >= Hin || (x2 >> 1) >= Win) continue;
                for (int kq = 0; kq < 4; ++kq) {
                  const int ci = 4 * cq + kq;
                  if (ci >= Cin) break;
                  acc = fmaf(bottom[(((size_t)n * bottom_channels + bottom_c0 + ci) * Hin + (y2 >> 1)) * Win + (x2 >> 1)],
                             weight[(((size_t)ci * Cout + co) * kernel + ky) * kernel + kx], acc);
                }
              }
          float t = acc + (bias ? bias[co] : 0.f);
          if (relu) t = t > 0.f ? t : t * negative_slope;
          top[(((size_t)n * top_channels + top_c0 + co) * Hout + Y) * Wout + X] = t;
        }
  return FN2_OK;
}

/* Weight gradient of a Convolution / Deconvolution layer: the CPU twin of csrc/conv_wgrad.hip.
 * Reference arithmetic: ConvolutionLayer::Backward_cpu (conv_layer.cpp:43-66) -> weight_cpu_gemm (base_conv_layer.cpp:289-303: per sample
 * im2col + cblas_sgemm(top_diff x col^T) accumulated into weight_diff):
 *     dw[ca][cb][ky][kx] (+)= sum_{n,y,x} a[n][ca][y][x] * b[n][cb][stride y + ky - pad][stride x + kx - pad]     (0 outside b)
 * with a = top_diff, b = bottom for a Convolution; a = bottom, b = top_diff for a Deconvolution (deconv_layer.cpp:28-45, the roles swapped).
 * Summation order of the HIP kernel: the rows u = n * Ha + y of `a` are cut into `ksplit` contiguous parts [p U / ksplit, (p + 1) U /
 * ksplit); per part one fmaf chain over its pixels in (n, y, x) order; the parts added in part order.  ksplit is what
 * fn2_conv_wgrad_ksplit() reports (passed in: the oracle does not link the HIP library). */
FN2_API int fn2_conv_wgrad_cpu(const float* a, const float* b, float* dw,
                               int N, int Ca, int Ha, int Wa, int a_channels, int a_c0,
                               int Cb, int Hb, int Wb, int b_channels, int b_c0,
                               int kernel, int stride, int pad, int accumulate, int ksplit) {
  if (!a || !b || !dw || N < 0 || Ca < 1 || Ha < 1 || Wa < 1 || Cb < 1 || Hb < 1 || Wb < 1 || kernel < 1 || stride < 1 || pad < 0 || ksplit < 1)
    return FN2_ERR_INVALID_ARG;
  if (a_c0 < 0 || a_c0 + Ca > a_channels || b_c0 < 0 || b_c0 + Cb > b_channels) return FN2_ERR_INVALID_ARG;
  const int U = N * Ha;
  if (ksplit > (U > 0 ? U : 1)) return FN2_ERR_INVALID_ARG;
#pragma omp parallel for collapse(2)
  for (int ca = 0; ca < Ca; ++ca)
    for (int cb = 0; cb < Cb; ++cb)
      for (int ky = 0; ky < kernel; ++ky)
        for (int kx = 0; kx < kernel; ++kx) {
          float sum = 0.f;
          for (int part = 0; part < ksplit; ++part) {
            float acc = 0.f;
            const int u0 = (int)((long long)part * U / ksplit), u1 = (int)((long long)(part + 1) * U / ksplit);
            for (int u = u0; u < u1; ++u) {
              const int n = u / Ha, y = u % Ha, yi = stride * y + ky - pad;
              if (yi < 0 || yi >= Hb) continue;                                           /* fmaf(a, 0, acc) == acc */
              const float* ar = a + (((size_t)n * a_channels + a_c0 + ca) * Ha + y) * Wa;
              const float* br = b + (((size_t)n * b_channels + b_c0 + cb) * Hb + yi) * Wb;
              for (int x = 0; x < Wa; ++x) {
                const int xi = stride * x + kx - pad;
                if (xi >= 0 && xi < Wb) acc = fmaf(ar[x], br[xi], acc);
              }
            }
            sum = part == 0 ? acc : sum + acc;
          }
          float* o = dw + (((size_t)ca * Cb + cb) * kernel + ky) * kernel + kx;
          *o = accumulate ? *o + sum : sum;
        }
  return FN2_OK;
}

/* Deconvolution{4x4, stride 2, pad 1} + bias + optional ReLU on packed weights: the CPU twin of csrc/conv_plane.hip (MODE 1).
 * Reference arithmetic: DeconvolutionLayer::Forward_cpu (deconv_layer.cpp:8-26: weight^T x bottom, col2im, bias) and the in-place ReLU
 * (relu_layer.cpp:23-30): out[Y][X] = sum_ci sum_{ky, kx} in[y][x] W[ci][co][ky][kx] with Y = 2 y - 1 + ky, X = 2 x - 1 + kx.  Every
 * output pixel (Y, X) = (2 m + py, 2 l + px) has 2 x 2 contributing taps: rows in[m + py - 1 + a'] with ky = (3, 1) for py = 0 and
 * (2, 0) for py = 1 (a' = 0, 1), columns likewise; summation order (channel quad, a', b', channel within the quad) with fmaf, K split as
 * in fn2_conv_plane_forward_cpu.  Packed layout: [class 2 py + px][Cout/64][(quads padded to 2-quad units) * 4 + 8 spare][lane 64][4],
 * lane = 16 kq + co, element j: W[4 cq + kq][64 g + 16 j + co][ky(py, a')][kx(px, b')], k-step = (cq * 2 + a') * 2 + b'. */
static int deconv_plane_ksteps(int Cin) { return (((Cin + 3) / 4 + 1) / 2) * 2 * 4; }

FN2_API size_t fn2_deconv_plane_packed_floats_cpu(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cout % 64 != 0) return 0;
  return 4 * (size_t)(Cout / 64) * (deconv_plane_ksteps(Cin) + 8) * 256;
}

static int deconv_tap(int parity, int t) { return parity == 0 ? (t == 0 ? 3 : 1) : (t == 0 ? 2 : 0); }

/* src_kernel 4: the Deconvolution{4, 2, 1} blob; 3: a [Cin][Cout][3][3] blob read as the 4x4 one whose fourth tap row / column are zero */
FN2_API int fn2_deconv_plane_pack_weights_k_cpu(const float* weight, float* packed, int Cin, int Cout, int src_kernel) {
  if (!weight || !packed || Cin <= 0 || Cout <= 0 || Cout % 64 != 0 || (src_kernel != 3 && src_kernel != 4)) return FN2_ERR_INVALID_ARG;
  const int ksteps = deconv_plane_ksteps(Cin), kalloc = ksteps + 8, sk = src_kernel;
  for (int cls = 0; cls < 4; ++cls)
    for (int g = 0; g < Cout / 64; ++g)
      for (int ks = 0; ks < kalloc; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int co = 64 * g + 16 * j + (lane & 15), ci = 4 * (ks / 4) + (lane >> 4);
            const int ky = deconv_tap(cls >> 1, (ks >> 1) & 1), kx = deconv_tap(cls & 1, ks & 1);
            packed[((((size_t)cls * (Cout / 64) + g) * kalloc + ks) * 64 + lane) * 4 + j] =
                (ks < ksteps && ci < Cin && ky < sk && kx < sk) ? weight[(((size_t)ci * Cout + co) * sk + ky) * sk + kx] : 0.f;
          }
  return FN2_OK;
}

FN2_API int fn2_deconv_plane_pack_weights_cpu(const float* weight, float* packed, int Cin, int Cout) {
  return fn2_deconv_plane_pack_weights_k_cpu(weight, packed, Cin, Cout, 4);
}

FN2_API int fn2_deconv_plane_forward_cpu(const float* bottom, const float* packed, const float* bias, float* top,
                                         int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                         int Cout, int top_channels, int top_c0, int relu, float negative_slope, int ksplit) {
  if (N < 0 || Cin < 1 || Hin < 1 || Win < 1 || Cout < 1 || Cout % 64 != 0 || ksplit < 1) return FN2_ERR_INVALID_ARG;
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels) return FN2_ERR_INVALID_ARG;
  const int quads = (Cin + 3) / 4, units = (quads + 1) / 2, kalloc = deconv_plane_ksteps(Cin) + 8;
  if (ksplit > units) return FN2_ERR_INVALID_ARG;
  const int Ho = 2 * Hin, Wo = 2 * Win;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int Y = 0; Y < Ho; ++Y)
        for (int X = 0; X < Wo; ++X) {
          const int py = Y & 1, px = X & 1, m = Y >> 1, l = X >> 1, cls = 2 * py + px;
          const float* wg = packed + ((size_t)cls * (Cout / 64) + co / 64) * kalloc * 256 + ((co % 64) / 16) + 4 * (co % 16);
          float sum = 0.f;
          for (int part = 0; part < ksplit; ++part) {
            float acc = 0.f;
            const int q0 = 2 * (int)((long long)part * units / ksplit), q1 = 2 * (int)((long long)(part + 1) * units / ksplit);
            for (int cq = q0; cq < q1; ++cq)
              for (int ta = 0; ta < 2; ++ta)
                for (int tb = 0; tb < 2; ++tb) {
                  const int ks = (cq * 2 + ta) * 2 + tb;
                  const int yi = m + py - 1 + ta, xi = l + px - 1 + tb;
                  for (int kq = 0; kq < 4; ++kq) {
                    float v = 0.f;
                    if (4 * cq + kq < Cin && yi >= 0 && yi < Hin && xi >= 0 && xi < Win)
                      v = bottom[(((size_t)n * bottom_channels + bottom_c0 + 4 * cq + kq) * Hin + yi) * Win + xi];
                    acc = fmaf(v, wg[(size_t)ks * 256 + 64 * kq], acc);
                  }
                }
            sum = part == 0 ? acc : sum + acc;
          }
          float t = sum + (bias ? bias[co] : 0.f);
          if (relu) t = t > 0.f ? t : t * negative_slope;
          top[(((size_t)n * top_channels + top_c0 + co) * Ho + Y) * Wo + X] = t;
        }
  return FN2_OK;
}

/* 3x3 / stride 1 convolution as Winograd F(2x2, 3x3): the CPU twin of csrc/conv_wino.hip, operation for operation.
 * Reference arithmetic: conv_layer.cpp:25-40 / base_conv_layer.cpp:255-318 (+ relu_layer.cpp:23-30); the reference sums K = 9 Cin
 * products per output in a library-defined order, Winograd sums Cin products per transform-domain position and combines 16 of
 * them: both are fp32 roundings of the same exact value (tests compare against the reference's layer and fp64 at 1e-5 x scale).
 * This restatement follows the HIP kernel's order exactly -- U = G g G^T ((a + b) + c) * 0.5, V = B^T d B rows first, fmaf chains over
 * the channels in ascending order per position (padded channels contribute exact zeros), Y = A^T M A rows first -- so it is bit-identical.
 * Packed layout: [Cout/16][quads][position quad][lane = 16 kq + co][4]. */
static int wino_kquads(int Cin) { return (((Cin + 3) / 4 + 1) / 2) * 2; }

static void wino_u_cpu(const float g[3][3], float U[4][4]) {
  float tmp[4][3];
  for (int k = 0; k < 3; ++k) {
    tmp[0][k] = g[0][k];
    tmp[1][k] = ((g[0][k] + g[1][k]) + g[2][k]) * 0.5f;
    tmp[2][k] = ((g[0][k] - g[1][k]) + g[2][k]) * 0.5f;
    tmp[3][k] = g[2][k];
  }
  for (int i = 0; i < 4; ++i) {
    U[i][0] = tmp[i][0];
    U[i][1] = ((tmp[i][0] + tmp[i][1]) + tmp[i][2]) * 0.5f;
    U[i][2] = ((tmp[i][0] - tmp[i][1]) + tmp[i][2]) * 0.5f;
    U[i][3] = tmp[i][2];
  }
}

FN2_API size_t fn2_conv_wino_packed_floats_cpu(int Cout, int Cin) {
  if (Cout <= 0 || Cout % 16 != 0 || Cin <= 0) return 0;
  return (size_t)(Cout / 16) * wino_kquads(Cin) * 1024;
}

FN2_API int fn2_conv_wino_pack_weights_cpu(const float* weight, float* packed, int Cout, int Cin) {
  if (!weight || !packed || Cout <= 0 || Cout % 16 != 0 || Cin <= 0) return FN2_ERR_INVALID_ARG;
  const int kquads = wino_kquads(Cin);
  for (int grp = 0; grp < Cout / 16; ++grp)
    for (int cq = 0; cq < kquads; ++cq)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = 16 * grp + (lane & 15), ci = 4 * cq + (lane >> 4);
        float g[3][3], U[4][4];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) g[a][b] = ci < Cin ? weight[((size_t)co * Cin + ci) * 9 + a * 3 + b] : 0.f;
        wino_u_cpu(g, U);
        float* dst = packed + ((size_t)grp * kquads + cq) * 1024 + lane * 4;
        for (int pq = 0; pq < 4; ++pq)
          for (int e = 0; e < 4; ++e) dst[pq * 256 + e] = U[pq][e];
      }
  return FN2_OK;
}

FN2_API int fn2_conv_wino_forward_cpu(const float* bottom, const float* packed, const float* bias, float* top,
                                      int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                      int Cout, int top_channels, int top_c0, int pad, int relu, float negative_slope) {
  if (N < 0 || Cin < 1 || Hin < 1 || Win < 1 || Cout < 1 || Cout % 16 != 0 || pad != 1) return FN2_ERR_INVALID_ARG;
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels) return FN2_ERR_INVALID_ARG;
  const int Ho = Hin + 2 * pad - 2, Wo = Win + 2 * pad - 2, kquads = wino_kquads(Cin);
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co) {
      const float* ug = packed + (size_t)(co / 16) * kquads * 1024 + (co % 16) * 4;
      for (int ty = 0; 2 * ty < Ho; ++ty)
        for (int tx = 0; 2 * tx < Wo; ++tx) {
          float M[16];
          for (int p = 0; p < 16; ++p) M[p] = 0.f;
          for (int c = 0; c < 4 * kquads; ++c) {
            float d[4][4], w[4][4], v[4][4];
            for (int i = 0; i < 4; ++i)
              for (int j = 0; j < 4; ++j) {
                const int yi = 2 * ty - pad + i, xi = 2 * tx - pad + j;
                d[i][j] = (c < Cin && yi >= 0 && yi < Hin && xi >= 0 && xi < Win)
                              ? bottom[(((size_t)n * bottom_channels + bottom_c0 + c) * Hin + yi) * Win + xi] : 0.f;
              }
            for (int j = 0; j < 4; ++j) {
              w[0][j] = d[0][j] - d[2][j];
              w[1][j] = d[1][j] + d[2][j];
              w[2][j] = d[2][j] - d[1][j];
              w[3][j] = d[1][j] - d[3][j];
            }
            for (int i = 0; i < 4; ++i) {
              v[i][0] = w[i][0] - w[i][2];
              v[i][1] = w[i][1] + w[i][2];
              v[i][2] = w[i][2] - w[i][1];
              v[i][3] = w[i][1] - w[i][3];
            }
            const float* uc = ug + (size_t)(c / 4) * 1024 + (c % 4) * 64;          /* lane = 16 kq + co */
            for (int p = 0; p < 16; ++p) M[p] = fmaf(v[p >> 2][p & 3], uc[(p >> 2) * 256 + (p & 3)], M[p]);
          }
          float t[2][4], y[2][2];
          for (int k = 0; k < 4; ++k) {
            t[0][k] = M[k] + M[4 + k] + M[8 + k];
            t[1][k] = M[4 + k] - M[8 + k] - M[12 + k];
          }
          for (int h = 0; h < 2; ++h) {
            y[h][0] = t[h][0] + t[h][1] + t[h][2];
            y[h][1] = t[h][1] - t[h][2] - t[h][3];
          }
          for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 2; ++e) {
              const int oy = 2 * ty + h, ox = 2 * tx + e;
              if (oy >= Ho || ox >= Wo) continue;
              float s = y[h][e] + (bias ? bias[co] : 0.f);
              if (relu) s = s > 0.f ? s : s * negative_slope;
              top[(((size_t)n * top_channels + top_c0 + co) * Ho + oy) * Wo + ox] = s;
            }
        }
    }
  return FN2_OK;
}

/* Batched im2col / col2im of Caffe's GEMM convolution (src/caffe/util/im2col.cpp:20-50 im2col_cpu, :168-200 col2im_cpu;
 * GPU twins im2col.cu:8-72, 246-318).  col2im carries the deconvolution's bias and optional leaky ReLU like the HIP path;
 * the additions run over the column grid rows ascending, then columns ascending (the reference GPU kernel's order). */
FN2_API int fn2_im2col_forward_cpu(const float* im, float* col, int N, int C, int H, int W, int kernel, int pad, int stride) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || kernel < 1 || pad < 0 || stride < 1 || H + 2 * pad < kernel || W + 2 * pad < kernel)
    return FN2_ERR_INVALID_ARG;
  const int Hc = (H + 2 * pad - kernel) / stride + 1, Wc = (W + 2 * pad - kernel) / stride + 1;
#pragma omp parallel for
  for (long long pl = 0; pl < (long long)N * C; ++pl) {
    const float* src = im + (size_t)pl * H * W;
    float* dst = col + (size_t)pl * kernel * kernel * Hc * Wc;
    for (int i = 0; i < kernel; ++i)
      for (int j = 0; j < kernel; ++j)
        for (int yc = 0; yc < Hc; ++yc)
          for (int xc = 0; xc < Wc; ++xc) {
            const int y = yc * stride - pad + i, x = xc * stride - pad + j;
            dst[((size_t)(i * kernel + j) * Hc + yc) * Wc + xc] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0.f;
          }
  }
  return FN2_OK;
}

static int col2im_bias_relu_cpu(const float* col, const float* bias, float* im, int N, int C, int H, int W,
                                int kernel, int pad, int stride, int apply_relu, float negative_slope, int im_ctot, int im_c0);

FN2_API int fn2_col2im_bias_relu_forward_cpu(const float* col, const float* bias, float* im, int N, int C, int H, int W,
                                             int kernel, int pad, int stride, int apply_relu, float negative_slope) {
  return col2im_bias_relu_cpu(col, bias, im, N, C, H, W, kernel, pad, stride, apply_relu, negative_slope, C, 0);
}

/* the same pass into channels [top_c0, top_c0 + C) of a wider top blob */
FN2_API int fn2_col2im_bias_relu_forward_into_cpu(const float* col, const float* bias, float* top, int N, int C, int H, int W,
                                                  int kernel, int pad, int stride, int apply_relu, float negative_slope,
                                                  int top_channels, int top_c0) {
  if (top_c0 < 0 || top_c0 + C > top_channels) return FN2_ERR_INVALID_ARG;
  return col2im_bias_relu_cpu(col, bias, top, N, C, H, W, kernel, pad, stride, apply_relu, negative_slope, top_channels, top_c0);
}

static int col2im_bias_relu_cpu(const float* col, const float* bias, float* im, int N, int C, int H, int W,
                                int kernel, int pad, int stride, int apply_relu, float negative_slope, int im_ctot, int im_c0) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || kernel < 1 || pad < 0 || stride < 1 || H + 2 * pad < kernel || W + 2 * pad < kernel)
    return FN2_ERR_INVALID_ARG;
  const int Hc = (H + 2 * pad - kernel) / stride + 1, Wc = (W + 2 * pad - kernel) / stride + 1;
#pragma omp parallel for
  for (long long pl = 0; pl < (long long)N * C; ++pl) {
    const float* src = col + (size_t)pl * kernel * kernel * Hc * Wc;
    const float b = bias ? bias[pl % C] : 0.f;
    for (int yy = 0; yy < H; ++yy)
      for (int xx = 0; xx < W; ++xx) {
        const int y = yy + pad, x = xx + pad;
        float v = 0.f;
        for (int yc = 0; yc < Hc; ++yc) {
          const int i = y - yc * stride;
          if (i < 0 || i >= kernel) continue;
          for (int xc = 0; xc < Wc; ++xc) {
            const int j = x - xc * stride;
            if (j < 0 || j >= kernel) continue;
            v += src[((size_t)(i * kernel + j) * Hc + yc) * Wc + xc];
          }
        }
        v += b;
        im[((size_t)(pl / C) * im_ctot + im_c0 + pl % C) * H * W + (size_t)yy * W + xx] = (apply_relu && v <= 0.f) ? v * negative_slope : v;
      }
  }
  return FN2_OK;
}


/* Backward of bias + leaky ReLU: relu_layer.cpp:33-45 (bottom_diff = top_diff * ((data > 0) + slope * (data <= 0)), evaluated on
 * the in-place blob) and backward_cpu_bias (base_conv_layer.cpp:319-323: bias_diff += top_diff summed over the positions). */
FN2_API int fn2_bias_leaky_relu_backward_cpu(const float* top_data, const float* top_diff, float* bottom_diff, float* bias_diff,
                                             int N, int C, int H, int W, float negative_slope) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  for (int c = 0; c < C; ++c) {
    double acc = 0.0;
    for (int n = 0; n < N; ++n) {
      const size_t base = ((size_t)n * C + c) * hw;
      for (size_t i = 0; i < hw; ++i) {
        const float g = top_diff[base + i] * (top_data[base + i] > 0.f ? 1.f : negative_slope);
        bottom_diff[base + i] = g;
        acc += g;
      }
    }
    if (bias_diff) bias_diff[c] = (float)acc;
  }
  return FN2_OK;
}

FN2_API int fn2_bias_leaky_relu_backward_slices_cpu(const float* top_data, const float* top_diff, int diff_channels, int diff_c0,
                                                    float* bottom_diff, float* bias_diff, int N, int C, int H, int W, float negative_slope);
/* ... with top_data a channel slice of a wider blob as well. */
FN2_API int fn2_bias_leaky_relu_backward_slices2_cpu(const float* top_data, int data_channels, int data_c0, const float* top_diff,
                                                     int diff_channels, int diff_c0, float* bottom_diff, float* bias_diff, int N, int C, int H,
                                                     int W, float negative_slope) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || data_c0 < 0 || data_c0 + C > data_channels) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  float* y = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * C * hw);
  if (!y) return FN2_ERR_INVALID_ARG;
  for (int n = 0; n < N; ++n)
    memcpy(y + (size_t)n * C * hw, top_data + ((size_t)n * data_channels + data_c0) * hw, sizeof(float) * C * hw);
  const int rc = fn2_bias_leaky_relu_backward_slices_cpu(y, top_diff, diff_channels, diff_c0, bottom_diff, bias_diff, N, C, H, W, negative_slope);
  free(y);
  return rc;
}

/* Bias gradient alone: backward_cpu_bias (base_conv_layer.cpp:319-323: bias_diff += top_diff summed over the positions of every sample,
 * the GEMV's beta = 1); top_diff may be a channel slice of a wider blob. */
FN2_API int fn2_conv_backward_bias_cpu(const float* top_diff, int diff_channels, int diff_c0, float* bias_diff, int N, int C, int H, int W,
                                       int accumulate) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || diff_c0 < 0 || diff_c0 + C > diff_channels) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  for (int c = 0; c < C; ++c) {
    double acc = 0.0;
    for (int n = 0; n < N; ++n) {
      const size_t base = ((size_t)n * diff_channels + diff_c0 + c) * hw;
      for (size_t i = 0; i < hw; ++i) acc += top_diff[base + i];
    }
    bias_diff[c] = accumulate ? bias_diff[c] + (float)acc : (float)acc;
  }
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * CustomData sample format.
 *   Datum: src/caffe/proto/caffe.proto:30-41, encoded by libprotobuf (third-party, version unpinned by the reference's
 *   Makefile); the wire format is the published proto2 encoding: key = (field << 3) | wire_type as a base-128 varint, wire types
 *   0 varint, 1 fixed64, 2 length-delimited, 5 fixed32; int32 values are sign-extended to 64 bits; repeated scalars may arrive
 *   packed.  tests/test_sample_format.py pins these functions against the protobuf runtime installed here.
 *   Writer: ImagePair::read_data, tools/convert_imageset_and_flow.cpp:142-206.
 *   Reader: DecodeData (src/caffe/layers/custom_data_layer.cpp:44-136) + the slice copy of CustomDataLayerPrefetch (:209-300).
 * ---------------------------------------------------------------------------------------------- */
static int pb_varint(const unsigned char** p, const unsigned char* end, uint64_t* v) {
  uint64_t r = 0;
  int shift = 0;
  while (*p < end && shift < 64) {
    const unsigned char b = *(*p)++;
    r |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) { *v = r; return 1; }
    shift += 7;
  }
  return 0;
}

/* unknown group: everything up to the END_GROUP key (wire type 4) of the same field number; groups nest (recursion limit 100) */
static int pb_skip_group(const unsigned char** p, const unsigned char* end, unsigned field, int depth) {
  if (depth > 100) return 0;
  while (*p < end) {
    uint64_t key, x;
    if (!pb_varint(p, end, &key) || key > 0xffffffffu || (key >> 3) == 0) return 0;
    const unsigned f = (unsigned)(key >> 3), wt = (unsigned)(key & 7);
    if (wt == 0) { if (!pb_varint(p, end, &x)) return 0; }
    else if (wt == 1) { if ((size_t)(end - *p) < 8) return 0; *p += 8; }
    else if (wt == 2) { if (!pb_varint(p, end, &x) || (uint64_t)(end - *p) < x) return 0; *p += x; }
    else if (wt == 3) { if (!pb_skip_group(p, end, f, depth + 1)) return 0; }
    else if (wt == 4) return f == field;
    else if (wt == 5) { if ((size_t)(end - *p) < 4) return 0; *p += 4; }
    else return 0;
  }
  return 0;
}

static int datum_walk_cpu(const void* buf, size_t len, fn2_datum_view* out, float* fdst, size_t fcap) {
  const unsigned char* p = (const unsigned char*)buf;
  const unsigned char* end = p + len;
  fn2_datum_view v;
  memset(&v, 0, sizeof(v));
  if (!buf && len) return FN2_ERR_INVALID_ARG;
  while (p < end) {
    uint64_t key, x;
    if (!pb_varint(&p, end, &key) || key > 0xffffffffu) return FN2_ERR_INVALID_ARG;     /* a key is a 32-bit varint */
    const unsigned field = (unsigned)(key >> 3), wt = (unsigned)(key & 7);
    if (field == 0) return FN2_ERR_INVALID_ARG;
    if (wt == 0) {
      if (!pb_varint(&p, end, &x)) return FN2_ERR_INVALID_ARG;
      switch (field) {
        case 1: v.channels = (int)(int64_t)x; break;
        case 2: v.height = (int)(int64_t)x; break;
        case 3: v.width = (int)(int64_t)x; break;
        case 5: v.label = (int)(int64_t)x; break;
        case 7: v.encoded = x != 0; break;
        default: break;
      }
    } else if (wt == 1) {
      if ((size_t)(end - p) < 8) return FN2_ERR_INVALID_ARG;
      p += 8;
    } else if (wt == 2) {
      if (!pb_varint(&p, end, &x) || (uint64_t)(end - p) < x) return FN2_ERR_INVALID_ARG;
      if (field == 4) { v.data = p; v.data_bytes = (size_t)x; }
      if (field == 6) {
        if (x % 4) return FN2_ERR_INVALID_ARG;
        for (uint64_t i = 0; i < x / 4; ++i) {
          if (fdst && v.float_data_count < fcap) memcpy(fdst + v.float_data_count, p + 4 * i, 4);
          v.float_data_count++;
        }
      }
      p += x;
    } else if (wt == 5) {
      if ((size_t)(end - p) < 4) return FN2_ERR_INVALID_ARG;
      if (field == 6) {
        if (fdst && v.float_data_count < fcap) memcpy(fdst + v.float_data_count, p, 4);
        v.float_data_count++;
      }
      p += 4;
    } else if (wt == 3) {
      if (!pb_skip_group(&p, end, field, 1)) return FN2_ERR_INVALID_ARG;                 /* an unknown group, skipped as a whole */
    } else {
      return FN2_ERR_INVALID_ARG;       /* END_GROUP without a group, wire types 6 / 7 */
    }
  }
  if (out) *out = v;
  return FN2_OK;
}

FN2_API int fn2_datum_parse_cpu(const void* buf, size_t len, fn2_datum_view* out) {
  if (!out) return FN2_ERR_INVALID_ARG;
  return datum_walk_cpu(buf, len, out, NULL, 0);
}

FN2_API int fn2_datum_float_data_cpu(const void* buf, size_t len, float* dst, size_t count) {
  fn2_datum_view v;
  int rc = datum_walk_cpu(buf, len, &v, dst, count);
  if (rc) return rc;
  return v.float_data_count == count ? FN2_OK : FN2_ERR_INVALID_ARG;
}

static size_t pb_put(unsigned char* dst, size_t pos, uint64_t v) {      /* appends a varint; counts only when dst == NULL */
  do {
    unsigned char b = (unsigned char)(v & 0x7f);
    v >>= 7;
    if (v) b |= 0x80;
    if (dst) dst[pos] = b;
    pos++;
  } while (v);
  return pos;
}

FN2_API long long fn2_datum_serialize_cpu(int channels, int height, int width, const void* data, size_t data_bytes, int label,
                                          void* dst, size_t dst_bytes) {
  if (!data && data_bytes) return FN2_ERR_INVALID_ARG;
  for (int pass = 0; pass < 2; ++pass) {
    unsigned char* d = pass ? (unsigned char*)dst : NULL;
    size_t pos = 0;
    pos = pb_put(d, pos, (1u << 3) | 0); pos = pb_put(d, pos, (uint64_t)(int64_t)channels);
    pos = pb_put(d, pos, (2u << 3) | 0); pos = pb_put(d, pos, (uint64_t)(int64_t)height);
    pos = pb_put(d, pos, (3u << 3) | 0); pos = pb_put(d, pos, (uint64_t)(int64_t)width);
    pos = pb_put(d, pos, (4u << 3) | 2); pos = pb_put(d, pos, (uint64_t)data_bytes);
    if (d && data_bytes) memcpy(d + pos, data, data_bytes);
    pos += data_bytes;
    pos = pb_put(d, pos, (5u << 3) | 0); pos = pb_put(d, pos, (uint64_t)(int64_t)label);
    if (!dst) return (long long)pos;
    if (!pass && dst_bytes < pos) return FN2_ERR_WORKSPACE;
    if (pass) return (long long)pos;
  }
  return FN2_ERR_INVALID_ARG;
}

/* byte size and channel range of every slice, custom_data_layer.cpp:66-86; returns the number of slices or a negative status */
typedef struct cd_slice { int c0, cc, enc; size_t off; } cd_slice;
static int cd_slices(int channels, int H, int W, const int* sp, int nsp, const int* enc, int nenc, int float_data,
                     cd_slice* out, size_t* total) {
  if (channels < 1 || H < 1 || W < 1 || nsp < 0 || nsp >= 32 || nenc < 0) return FN2_ERR_INVALID_ARG;
  if (float_data && nenc) return FN2_ERR_INVALID_ARG;                                            /* :55 */
  const size_t hw = (size_t)H * W;
  int channel_end = 0;
  size_t off = 0;
  for (int slice = 0; slice <= nsp; ++slice) {
    const int channel_start = channel_end;                                                       /* :70 */
    channel_end = (slice == nsp) ? channels : sp[slice];                                         /* :72-75 */
    const int channel_count = channel_end - channel_start;                                       /* :77 */
    if (channel_count < 1 || channel_end > channels) return FN2_ERR_INVALID_ARG;                 /* CHECK_GT :519 */
    const int format = float_data ? 0 : (nenc <= slice ? FN2_ENC_UINT8 : enc[slice]);            /* :79-83 */
    out[slice].c0 = channel_start; out[slice].cc = channel_count; out[slice].enc = format; out[slice].off = off;
    if (float_data) off += 4 * hw * channel_count;
    else if (format == FN2_ENC_UINT8) off += hw * channel_count;
    else if (format == FN2_ENC_UINT16FLOW) off += 2 * hw * channel_count;
    else if (format == FN2_ENC_BOOL1) { if (channel_count != 1) return FN2_ERR_INVALID_ARG; off += (hw - 1) / 8 + 1; }   /* :116, assert :135 */
    else return FN2_ERR_INVALID_ARG;                                                             /* :129-131 */
  }
  *total = off;
  return nsp + 1;
}

FN2_API size_t fn2_custom_data_sample_bytes_cpu(int channels, int H, int W, const int* slice_points, int n_slice_points,
                                                const int* encodings, int n_encodings) {
  cd_slice sl[32];
  size_t total = 0;
  return cd_slices(channels, H, W, slice_points, n_slice_points, encodings, n_encodings, 0, sl, &total) < 0 ? 0 : total;
}

FN2_API int fn2_custom_data_encode_sample_cpu(const unsigned char* img0, const unsigned char* img1, const float* flow,
                                              const unsigned char* occ, int H, int W, unsigned char* dst, size_t dst_bytes) {
  if (H < 1 || W < 1 || !img0 || !img1 || !dst) return FN2_ERR_INVALID_ARG;
  const int width = W, height = H;
  const size_t data_size = (size_t)3 * width * height + (size_t)3 * width * height + (size_t)2 * 2 * width * height +
                           ((size_t)width * height - 1) / 8 + 1;                                 /* tool :142-145 */
  if (dst_bytes < data_size) return FN2_ERR_WORKSPACE;
  memset(dst, 0, data_size);                                                                     /* :147 */
  unsigned char* ptr = dst;
  for (int c = 0; c < 3; ++c)                                                                    /* :151-156 */
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) *(ptr++) = img0[((size_t)y * width + x) * 3 + c];          /* cv::Vec3b at(y,x)[c] */
  for (int c = 0; c < 3; ++c)                                                                    /* :160-165 */
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) *(ptr++) = img1[((size_t)y * width + x) * 3 + c];
  for (size_t j = 0; j < (size_t)2 * width * height; j++) {                                      /* :169-181 */
    short value = 0;
    if (flow) {
      if (isnan(flow[j])) value = 32767;                                                         /* numeric_limits<short>::max() */
      else {
        /* `value = flo_data[j]*32`: float -> short conversion truncates toward zero; outside the range of short it is undefined
         * behaviour in C++ (flows beyond +-1024 px do not occur in the data sets): saturate */
        const float t = flow[j] * 32;
        value = t >= 32767.f ? 32767 : (t <= -32768.f ? -32768 : (short)t);
      }
    }
    *(ptr++) = *((unsigned char*)&value);                                                        /* host byte order (little-endian) */
    *(ptr++) = *((unsigned char*)&value + 1);
  }
  unsigned char current = 0;                                                                     /* :185-203 */
  int current_idx = 0;
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      unsigned char value = 0;
      if (occ) value = occ[(size_t)y * width + x] > 0;
      if (value) current |= 1 << current_idx;
      current_idx++;
      if (current_idx == 8) { *(ptr++) = current; current_idx = 0; current = 0; }
    }
  if (current_idx > 0) *(ptr++) = current;
  return (size_t)(ptr - dst) == data_size ? FN2_OK : FN2_ERR_INVALID_ARG;                        /* assert :204 */
}

/* The staging step in front of the decode: what CustomDataLayerPrefetch does per item before DecodeData (custom_data_layer.cpp:170-207:
 * fetch the value, Datum::ParseFromArray) plus the label (:297), with the payload copied out instead of decoded. */
FN2_API int fn2_custom_data_stage_records_cpu(const void* const* records, const size_t* record_bytes, int N, void* staging, size_t sample_stride,
                                              int* channels, int* height, int* width, size_t* sample_bytes, int* labels) {
  if (N < 1 || !records || !record_bytes) return FN2_ERR_INVALID_ARG;
  fn2_datum_view first;
  for (int item_id = 0; item_id < N; ++item_id) {
    fn2_datum_view d;
    int rc = datum_walk_cpu(records[item_id], record_bytes[item_id], &d, NULL, 0);
    if (rc) return rc;
    if (!d.data) return FN2_ERR_INVALID_ARG;
    if (item_id == 0) first = d;
    if (d.channels != first.channels || d.height != first.height || d.width != first.width || d.data_bytes != first.data_bytes) return FN2_ERR_INVALID_ARG;
    if (labels) labels[item_id] = d.label;
    if (staging) {
      if (sample_stride < d.data_bytes) return FN2_ERR_WORKSPACE;
      memcpy((unsigned char*)staging + (size_t)item_id * sample_stride, d.data, d.data_bytes);
    }
  }
  if (channels) *channels = first.channels;
  if (height) *height = first.height;
  if (width) *width = first.width;
  if (sample_bytes) *sample_bytes = first.data_bytes;
  return FN2_OK;
}

/* All pointers are HOST pointers here.  Restates DecodeData (whole datum -> floats, :44-136) and then the slice copy with the mean
 * and the scale (:209-300, the crop_size == 0 branch :274-284). */
FN2_API int fn2_custom_data_decode_forward_cpu(const void* samples, size_t sample_stride, int N, int channels, int H, int W,
                                               const int* slice_points, int n_slice_points, const int* encodings, int n_encodings,
                                               int float_data, const float* mean, float scale, float* const* tops) {
  cd_slice sl[32];
  size_t total = 0;
  const int ns = cd_slices(channels, H, W, slice_points, n_slice_points, encodings, n_encodings, float_data, sl, &total);
  if (ns < 0) return ns;
  if (N < 0 || (N && (!samples || !tops)) || sample_stride < total) return FN2_ERR_INVALID_ARG;
  const int width = W, height = H;
  const size_t heightwidth = (size_t)H * W, count = heightwidth * channels;
  float* decoded = (float*)malloc(count * sizeof(float));
  float* zero_mean = (float*)calloc(count, sizeof(float));                                       /* data_mean_ "all-empty", :612-615 */
  if (!decoded || !zero_mean) { free(decoded); free(zero_mean); return FN2_ERR_WORKSPACE; }
  const float* mean_data = mean ? mean : zero_mean;
  for (int item_id = 0; item_id < N; ++item_id) {
    const unsigned char* srcptr = (const unsigned char*)samples + (size_t)item_id * sample_stride;
    float* destptr = decoded;
    if (float_data) {
      memcpy(decoded, srcptr, count * sizeof(float));                                            /* :57-58 */
    } else {
      for (int slice = 0; slice < ns; ++slice) {
        const int channel_count = sl[slice].cc;
        switch (sl[slice].enc) {
          case FN2_ENC_UINT8:                                                                    /* :88-92 */
            for (int c = 0; c < channel_count; c++)
              for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) *(destptr++) = (float)(*(srcptr++));
            break;
          case FN2_ENC_UINT16FLOW:                                                               /* :94-111 */
            for (int c = 0; c < channel_count; c++)
              for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                  short v;
                  *((unsigned char*)&v) = *(srcptr++);
                  *((unsigned char*)&v + 1) = *(srcptr++);
                  float value;
                  if (v == 32767) { const uint32_t snan = 0x7fa00000u; memcpy(&value, &snan, 4); }   /* signaling_NaN(), :105 */
                  else value = ((float)v) / 32.0f;                                               /* :107 */
                  *(destptr++) = value;
                }
            break;
          default: {                                                                             /* BOOL1, :113-128 */
            size_t j = 0;
            for (size_t i = 0; i < (heightwidth - 1) / 8 + 1; i++) {
              const unsigned char data = *(srcptr++);
              for (int k = 0; k < 8; k++) {
                const float value = (data & (1 << k)) == (1 << k);
                if (j < heightwidth) *(destptr++) = value ? 1.0f : 0.f;
                j++;
              }
            }
          }
        }
      }
      if (destptr != decoded + count) { free(decoded); free(zero_mean); return FN2_ERR_INVALID_ARG; }   /* assert :135 */
    }
    for (int slice = 0; slice < ns; ++slice) {                                                   /* :216-284 */
      float* top_data = tops[slice];
      const int slice_channel_count = sl[slice].cc, src_channel_start = sl[slice].c0;
      for (int c = 0; c < slice_channel_count; ++c)
        for (size_t hw = 0; hw < heightwidth; ++hw) {
          const size_t top_index = ((size_t)item_id * slice_channel_count + c) * heightwidth + hw;   /* :278 */
          const size_t data_index = (size_t)(src_channel_start + c) * heightwidth + hw;              /* :279 */
          volatile float datum_element = decoded[data_index];                                       /* keeps the sNaN -> qNaN subtraction */
          top_data[top_index] = (datum_element - mean_data[data_index]) * scale;                    /* :282 */
        }
    }
  }
  free(decoded);
  free(zero_mean);
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * FlowAugmentation: flow_augmentation_layer.cpp:30-72, flow_augmentation_layer.cu:23-160; matrix helpers
 * augmentation_layer_base.cpp:14-68; coefficient arrays :352-380 (index = field order of AugmentationCoeff, caffe.proto:436-486;
 * fields with a non-zero default are stored as log and come back through exp()).
 * Number types follow the reference: tTransMat holds floats, leftMultiply takes floats, its call sites compute in double
 * (.5 * float, cos(double), 1.0 / float) and convert.
 * ---------------------------------------------------------------------------------------------- */
typedef struct aug_mat { float t0, t2, t4, t1, t3, t5; } aug_mat;

static void aug_left_multiply(aug_mat* m, float u0, float u1, float u2, float u3, float u4, float u5) {   /* cpp:22-35 */
  const float t0 = m->t0, t2 = m->t2, t4 = m->t4;
  const float t1 = m->t1, t3 = m->t3, t5 = m->t5;
  m->t0 = t0 * u0 + t1 * u2;
  m->t1 = t0 * u1 + t1 * u3;
  m->t2 = t2 * u0 + t3 * u2;
  m->t3 = t2 * u1 + t3 * u3;
  m->t4 = t4 * u0 + t5 * u2 + u4;
  m->t5 = t4 * u1 + t5 * u3 + u5;
}

static aug_mat aug_from_array(const float* in, int width, int height, int bottomwidth, int bottomheight) {
  /* array_to_coeff, cpp:368-380: defaults of mirror, dx, dy, angle are 0 (copied), of zoom_x, zoom_y 1 (exp) */
  const float mirror = in[0], dx = in[1], dy = in[2], angle = in[3];
  const float zoom_x = (float)exp((double)in[4]), zoom_y = (float)exp((double)in[5]);
  aug_mat m;
  m.t0 = 1; m.t2 = 0; m.t4 = 0; m.t1 = 0; m.t3 = 1; m.t5 = 0;                                                    /* toIdentity, cpp:15-19 */
  /* fromCoeff, cpp:38-49 (every has_*() is true after array_to_coeff) */
  if (mirror) aug_left_multiply(&m, -1, 0, 0, 1, (float)(.5 * (float)width), (float)(-.5 * (float)height));
  else aug_left_multiply(&m, 1, 0, 0, 1, (float)(-.5 * (float)width), (float)(-.5 * (float)height));
  aug_left_multiply(&m, (float)cos(angle), (float)sin(angle), (float)-sin(angle), (float)cos(angle), 0, 0);
  aug_left_multiply(&m, 1, 0, 0, 1, dx * (float)width, dy * (float)height);
  aug_left_multiply(&m, (float)(1.0 / zoom_x), 0, 0, (float)(1.0 / zoom_y), 0, 0);
  aug_left_multiply(&m, 1, 0, 0, 1, (float)(.5 * (float)bottomwidth), (float)(.5 * (float)bottomheight));
  return m;
}

static aug_mat aug_inverse(aug_mat s) {                                                                           /* cpp:52-68 */
  const float a = s.t0, c = s.t2, e = s.t4;
  const float b = s.t1, d = s.t3, f = s.t5;
  const float denom = a * d - b * c;
  aug_mat r;
  r.t0 = d / denom;
  r.t1 = -b / denom;
  r.t2 = -c / denom;
  r.t3 = a / denom;
  r.t4 = (c * f - d * e) / denom;
  r.t5 = (b * e - a * f) / denom;
  return r;
}

FN2_API int fn2_augmentation_matrix_cpu(const float* coeffs, int crop_width, int crop_height, int bottom_width, int bottom_height,
                                        int invert, float* mat6) {
  if (!coeffs || !mat6 || crop_width < 1 || crop_height < 1 || bottom_width < 1 || bottom_height < 1) return FN2_ERR_INVALID_ARG;
  aug_mat m = aug_from_array(coeffs, crop_width, crop_height, bottom_width, bottom_height);
  if (invert) m = aug_inverse(m);
  mat6[0] = m.t0; mat6[1] = m.t1; mat6[2] = m.t2; mat6[3] = m.t3; mat6[4] = m.t4; mat6[5] = m.t5;
  return FN2_OK;
}

/* WarpData, flow_augmentation_layer.cu:23-88; all pointers are host pointers here. */
FN2_API int fn2_flow_augmentation_forward_cpu(const float* flow, const float* coeffs1, const float* coeffs2, float* top,
                                              int N, int H, int W, int crop_height, int crop_width) {
  if (crop_width < 1 || crop_height < 1 || N < 0 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;               /* cpp:33-34 */
  if (N && (!flow || !coeffs1 || !coeffs2 || !top)) return FN2_ERR_INVALID_ARG;
  const int width = W, height = H, dest_width = crop_width, dest_height = crop_height;
  const long long src_count = (long long)N * 2 * H * W;
  for (int n = 0; n < N; ++n) {
    const aug_mat m1 = aug_from_array(coeffs1 + (size_t)n * FN2_AUG_NUM_PARAMS, crop_width, crop_height, W, H);             /* cu:131-137 */
    const aug_mat m2 = aug_inverse(aug_from_array(coeffs2 + (size_t)n * FN2_AUG_NUM_PARAMS, crop_width, crop_height, W, H)); /* cu:139-142 */
    for (int yi = 0; yi < dest_height; ++yi)
      for (int xi = 0; xi < dest_width; ++xi) {
        const float x = (float)xi, y = (float)yi;
        /* the device compiler contracts a*b + c*d + e into fma(a, b, fma(c, d, e)) (AMDGPU fuses aggressively; the position decides
         * which source pixel is read, so the rounding is kept) */
        const float xpos1 = fmaf(x, m1.t0, fmaf(y, m1.t2, m1.t4));                                            /* :41 */
        const float ypos1 = fmaf(x, m1.t1, fmaf(y, m1.t3, m1.t5));                                            /* :42 */
        const long long ix = (long long)width * ((long long)height * (2 * n + 0) + (int)(ypos1 + 0.5f)) + (int)(xpos1 + 0.5f);   /* :45-47 */
        const long long iy = (long long)width * ((long long)height * (2 * n + 1) + (int)(ypos1 + 0.5f)) + (int)(xpos1 + 0.5f);   /* :48-50 */
        /* the reference reads src_data[min(idx, src_count)]: unchecked below 0 and one element past the end above; both read 0 here */
        const float u = (ix >= 0 && ix < src_count) ? flow[ix] : 0.f;
        const float v = (iy >= 0 && iy < src_count) ? flow[iy] : 0.f;
        const float xpos2 = xpos1 + u, ypos2 = ypos1 + v;                                                     /* :52-53 */
        const float xpos3 = fmaf(xpos2, m2.t0, fmaf(ypos2, m2.t2, m2.t4));                                    /* :56 */
        const float ypos3 = fmaf(xpos2, m2.t1, fmaf(ypos2, m2.t3, m2.t5));                                    /* :57 */
        top[(size_t)dest_width * ((size_t)dest_height * (2 * n + 0) + yi) + xi] = xpos3 - x;                  /* :60 */
        top[(size_t)dest_width * ((size_t)dest_height * (2 * n + 1) + yi) + xi] = ypos3 - y;                  /* :61 */
      }
  }
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * DataAugmentation for given coefficients: data_augmentation_layer.cu:320-637 from the point where the coefficient blob exists
 * (:452), kernels :24-317, coefficient structs include/caffe/layers/augmentation_layer_base.hpp:37-127, clear_defaults
 * augmentation_layer_base.cpp:340-350.  The passes run one after the other over the whole batch, as in the reference.
 * ---------------------------------------------------------------------------------------------- */
enum { OA_MIRROR, OA_DX, OA_DY, OA_ANGLE, OA_ZOOM_X, OA_ZOOM_Y, OA_GAMMA, OA_BRIGHTNESS, OA_CONTRAST, OA_COLOR1, OA_COLOR2, OA_COLOR3,
       OA_POW_NOMEAN0, OA_ADD_NOMEAN0 = OA_POW_NOMEAN0 + 3, OA_MULT_NOMEAN0 = OA_ADD_NOMEAN0 + 3, OA_POW_WITHMEAN0 = OA_MULT_NOMEAN0 + 3,
       OA_ADD_WITHMEAN0 = OA_POW_WITHMEAN0 + 3, OA_MULT_WITHMEAN0 = OA_ADD_WITHMEAN0 + 3, OA_LMULT_POW = OA_MULT_WITHMEAN0 + 3, OA_LMULT_ADD,
       OA_LMULT_MULT, OA_COL_ANGLE, OA_FOG_AMOUNT, OA_FOG_SIZE, OA_MOTION_BLUR_ANGLE, OA_MOTION_BLUR_SIZE, OA_SHADOW_ANGLE,
       OA_SHADOW_DISTANCE, OA_SHADOW_STRENGTH, OA_NOISE, OA_COUNT };
static const float oa_default[42] = {0, 0, 0, 0, 1, 1,  1, 0, 1, 1, 1, 1,  1, 1, 1, 0, 0, 0,  1, 1, 1, 1, 1, 1,
                                     0, 0, 0, 1, 1, 1,  1, 0, 1, 0,  0, 0, 0, 0, 0, 0, 0, 0};   /* caffe.proto:436-486 */

typedef struct oa_coeff { float v[42]; int has[42]; } oa_coeff;

static void oa_from_array(const float* in, oa_coeff* c) {                       /* array_to_coeff, cpp:368-380 */
  for (int fn = 0; fn < 42; ++fn) {
    if (fabs(oa_default[fn]) < 1e-3) c->v[fn] = in[fn];
    else c->v[fn] = (float)exp((double)in[fn]);
    c->has[fn] = 1;
  }
}

static void oa_clear_defaults(oa_coeff* c) {                                    /* cpp:340-350 */
  for (int fn = 0; fn < 42; ++fn)
    if (fabs(oa_default[fn] - c->v[fn]) < 1e-3) { c->v[fn] = oa_default[fn]; c->has[fn] = 0; }
}

static aug_mat oa_matrix(const oa_coeff* c, int width, int height, int bottomwidth, int bottomheight) {   /* toIdentity + fromCoeff, cpp:15-49 */
  aug_mat m;
  m.t0 = 1; m.t2 = 0; m.t4 = 0; m.t1 = 0; m.t3 = 1; m.t5 = 0;
  if (c->v[OA_MIRROR]) aug_left_multiply(&m, -1, 0, 0, 1, (float)(.5 * (float)width), (float)(-.5 * (float)height));
  else aug_left_multiply(&m, 1, 0, 0, 1, (float)(-.5 * (float)width), (float)(-.5 * (float)height));
  if (c->has[OA_ANGLE]) aug_left_multiply(&m, (float)cos(c->v[OA_ANGLE]), (float)sin(c->v[OA_ANGLE]), (float)-sin(c->v[OA_ANGLE]), (float)cos(c->v[OA_ANGLE]), 0, 0);
  if (c->has[OA_DX] || c->has[OA_DY]) aug_left_multiply(&m, 1, 0, 0, 1, c->v[OA_DX] * (float)width, c->v[OA_DY] * (float)height);
  if (c->has[OA_ZOOM_X] || c->has[OA_ZOOM_Y]) aug_left_multiply(&m, (float)(1.0 / c->v[OA_ZOOM_X]), 0, 0, (float)(1.0 / c->v[OA_ZOOM_Y]), 0, 0);
  aug_left_multiply(&m, 1, 0, 0, 1, (float)(.5 * (float)bottomwidth), (float)(.5 * (float)bottomheight));
  return m;
}

static inline float oa_clamp(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }               /* cu:20-22 */

typedef struct oa_eigenspace { float mean_eig[3], mean_rgb[3], max_abs_eig[3], max_rgb[3], min_rgb[3], max_l, eigvec[9]; } oa_eigenspace;

/* Philox4x32-10 (Salmon et al., SC'11; Random123's constants), restated from csrc/philox.hpp: counter (c0..c3), key (k0, k1) -> 4 words */
static void oa_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static float oa_unit(uint32_t x) { return ((float)x + 1.0f) * 2.3283064365386963e-10f; }

/* the raw generator, for the known-answer test against numpy's Philox and Random123's published vectors */
FN2_API int fn2_philox4x32_10_cpu(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  oa_philox4x32_10(counter[0], counter[1], counter[2], counter[3], key[0], key[1], out);
  return FN2_OK;
}

FN2_API int fn2_data_augmentation_forward_cpu(const fn2_data_aug_params* p, const float* bottom, const float* coeffs, const float* mean,
                                              float* top, int N, int C, int H, int W) {
  if (!p || N < 0 || C < 1 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const int do_cropping = p->crop_width > 0 && p->crop_height > 0;                                      /* cpp:94 */
  const int cw = do_cropping ? p->crop_width : W, ch = do_cropping ? p->crop_height : H;
  if (W < cw || H < ch) return FN2_ERR_INVALID_ARG;                                                     /* cpp:103-104 */
  if (p->mean_mode < FN2_MEAN_NONE || p->mean_mode > FN2_MEAN_PER_PIXEL) return FN2_ERR_INVALID_ARG;
  if (N == 0) return FN2_OK;
  if (!bottom || !top || (p->mean_mode != FN2_MEAN_NONE && !mean)) return FN2_ERR_INVALID_ARG;
  const int width = W, height = H, channels = C;
  const long long src_count = (long long)N * C * H * W;
  const size_t area = (size_t)ch * cw, count = area * C;
  const float max_multiplier = p->max_multiplier;
  static const float zeros[42] = {0};

  if (!do_cropping) {
    memcpy(top, bottom, sizeof(float) * (size_t)src_count);                                             /* :590 */
  } else {
    oa_coeff* co = (oa_coeff*)malloc(sizeof(oa_coeff) * (size_t)N);
    aug_mat* mats = (aug_mat*)malloc(sizeof(aug_mat) * (size_t)N);
    if (!co || !mats) { free(co); free(mats); return FN2_ERR_WORKSPACE; }
    int has_chromatic = 0, has_eigen = 0, has_effect = 0;
    for (int n = 0; n < N; ++n) {                                                                       /* :452-476 */
      oa_from_array(coeffs ? coeffs + (size_t)n * 42 : zeros, &co[n]);
      oa_clear_defaults(&co[n]);
      mats[n] = oa_matrix(&co[n], cw, ch, W, H);
      const float* v = co[n].v;
      if (v[OA_GAMMA] != 1 || v[OA_BRIGHTNESS] != 0 || v[OA_CONTRAST] != 1 || v[OA_COLOR1] != 1 || v[OA_COLOR2] != 1 || v[OA_COLOR3] != 1) has_chromatic = 1;   /* hpp:48 */
      for (int f = OA_POW_NOMEAN0; f <= OA_COL_ANGLE; ++f) if (v[f] != oa_default[f]) has_eigen = 1;  /* hpp:86-94 */
      if ((v[OA_FOG_AMOUNT] != 0 && v[OA_FOG_SIZE] != 0) || v[OA_MOTION_BLUR_SIZE] > 0 || v[OA_SHADOW_STRENGTH] > 0 || v[OA_NOISE] > 0) has_effect = 1;   /* hpp:111 */
    }
    if ((has_chromatic || has_eigen || has_effect) && C != 3) { free(co); free(mats); return FN2_ERR_INVALID_ARG; }   /* :489,:548,:556 */
    if (has_eigen && !p->has_chromatic_eigvec) { free(co); free(mats); return FN2_ERR_INVALID_ARG; }   /* :493-494 (LOG(ERROR), then reads an empty list) */

    oa_eigenspace es;
    if (has_eigen) {                                                                                    /* :488-536 + ComputeChromaticEigenspace :147-187 */
      memset(&es, 0, sizeof(es));
      for (int i = 0; i < 9; ++i) es.eigvec[i] = p->chromatic_eigvec[i];
      for (int c = 0; c < 3; ++c) es.min_rgb[c] = FLT_MAX;
      for (int n = 0; n < N; ++n)
        for (size_t px = 0; px < (size_t)H * W; ++px) {
          float rgb[3];
          for (int c = 0; c < 3; ++c) rgb[c] = bottom[((size_t)n * 3 + c) * H * W + px];
          for (int c = 0; c < 3; ++c) {
            const float eig = es.eigvec[3 * c] * rgb[0] + es.eigvec[3 * c + 1] * rgb[1] + es.eigvec[3 * c + 2] * rgb[2];
            if (fabsf(eig) > es.max_abs_eig[c]) es.max_abs_eig[c] = fabsf(eig);
            if (rgb[c] > es.max_rgb[c]) es.max_rgb[c] = rgb[c];
            if (rgb[c] < es.min_rgb[c]) es.min_rgb[c] = rgb[c];
            es.mean_rgb[c] += rgb[c] / width / height;                                                  /* atomicAdd of per-thread terms, :176-179 */
          }
        }
      for (int c = 0; c < 3; ++c) es.mean_rgb[c] = es.mean_rgb[c] / N;                                  /* :517-518 */
      for (int c = 0; c < 3; ++c) {
        es.mean_eig[c] = es.eigvec[3 * c] * es.mean_rgb[0] + es.eigvec[3 * c + 1] * es.mean_rgb[1] + es.eigvec[3 * c + 2] * es.mean_rgb[2];
        if (es.max_abs_eig[c] > 1e-2) es.mean_eig[c] = es.mean_eig[c] / es.max_abs_eig[c];
      }
      es.max_l = sqrtf(es.max_abs_eig[0] * es.max_abs_eig[0] + es.max_abs_eig[1] * es.max_abs_eig[1] + es.max_abs_eig[2] * es.max_abs_eig[2]);
    }

    /* SpatialAugmentation, :24-69 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int cn = 0; cn < N * C; ++cn)
      for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
          const int n = cn / channels;
          const aug_mat* m = &mats[n];
          float xpos = fmaf((float)x, m->t0, fmaf((float)y, m->t2, m->t4));                             /* :41, contracted on the device */
          float ypos = fmaf((float)x, m->t1, fmaf((float)y, m->t3, m->t5));                             /* :42 */
          xpos = oa_clamp(xpos, 0.0f, (float)(width) - 1.05f);                                          /* :44 */
          ypos = oa_clamp(ypos, 0.0f, (float)(height) - 1.05f);                                         /* :45 */
          const float tlx = floorf(xpos), tly = floorf(ypos);
          const long long off = (long long)width * ((long long)height * cn + (long long)tly) + (long long)tlx;   /* :51 */
          const long long last = src_count - 1;          /* the reference clamps to src_count, one past the end; unreachable inside the image */
          const float sTL = bottom[off];
          const float sTR = bottom[off + 1 <= last ? off + 1 : last];
          const float sBL = bottom[off + width <= last ? off + width : last];
          const float sBR = bottom[off + 1 + width <= last ? off + 1 + width : last];
          const float xdist = xpos - tlx, ydist = ypos - tly;
          /* (1-xd)(1-yd) TL + xd yd BR + (1-xd) yd BL + xd (1-yd) TR, :61-64.  The device compiler contracts the sum of products
           * into an fma chain: the second product is rounded, the others are fused */
          float sample = xdist * ydist * sBR;
          sample = fmaf((1 - xdist) * (1 - ydist), sTL, sample);
          sample = fmaf((1 - xdist) * ydist, sBL, sample);
          sample = fmaf(xdist * (1 - ydist), sTR, sample);
          top[((size_t)cn * ch + y) * cw + x] = sample;                                                 /* :67 */
        }

    if (has_eigen) {                                                                                    /* ChromaticEigenAugmentation, :192-291 */
      for (int n = 0; n < N; ++n) {
        const float* k = co[n].v;
        for (size_t px = 0; px < area; ++px) {
          float rgb[3], eig[3], s, s1, l = 0, l1 = 0;
          for (int c = 0; c < 3; ++c) rgb[c] = top[((size_t)n * 3 + c) * area + px] - es.mean_rgb[c];
          for (int c = 0; c < 3; ++c) {
            eig[c] = es.eigvec[3 * c] * rgb[0] + es.eigvec[3 * c + 1] * rgb[1] + es.eigvec[3 * c + 2] * rgb[2];
            if (es.max_abs_eig[c] > 1e-2f) {
              eig[c] = eig[c] / es.max_abs_eig[c];
              eig[c] = copysignf(powf(fabsf(eig[c]), k[OA_POW_NOMEAN0 + c]), eig[c]);
              eig[c] = eig[c] + k[OA_ADD_NOMEAN0 + c];
              eig[c] = eig[c] * k[OA_MULT_NOMEAN0 + c];
            }
          }
          for (int c = 0; c < 3; ++c) eig[c] = eig[c] + es.mean_eig[c];
          if (es.max_abs_eig[0] > 1e-2f) {
            eig[0] = copysignf(powf(fabsf(eig[0]), k[OA_POW_WITHMEAN0]), eig[0]);
            eig[0] = eig[0] + k[OA_ADD_WITHMEAN0];
            eig[0] = eig[0] * k[OA_MULT_WITHMEAN0];
          }
          s = sqrtf(eig[1] * eig[1] + eig[2] * eig[2]);
          s1 = s;
          if (s > 1e-2f) {
            s1 = powf(s1, k[OA_POW_WITHMEAN0 + 1]);
            s1 = fmaxf(s1 + k[OA_ADD_WITHMEAN0 + 1], 0.f);
            s1 = s1 * k[OA_MULT_WITHMEAN0 + 1];
          }
          if (k[OA_COL_ANGLE] != 0) {
            const float t1 = cosf(k[OA_COL_ANGLE]) * eig[1] - sinf(k[OA_COL_ANGLE]) * eig[2];
            const float t2 = sinf(k[OA_COL_ANGLE]) * eig[1] + cosf(k[OA_COL_ANGLE]) * eig[2];
            eig[1] = t1;
            eig[2] = t2;
          }
          for (int c = 0; c < 3; ++c) if (es.max_abs_eig[c] > 1e-2f) eig[c] = eig[c] * es.max_abs_eig[c];
          if (es.max_l > 1e-2f) { l1 = sqrtf(eig[0] * eig[0] + eig[1] * eig[1] + eig[2] * eig[2]); l1 = l1 / es.max_l; }
          if (s > 1e-2f) { eig[1] = eig[1] / s * s1; eig[2] = eig[2] / s * s1; }
          if (es.max_l > 1e-2f) {
            l = sqrtf(eig[0] * eig[0] + eig[1] * eig[1] + eig[2] * eig[2]);
            l1 = powf(l1, k[OA_LMULT_POW]);
            l1 = fmaxf(l1 + k[OA_LMULT_ADD], 0.f);
            l1 = l1 * k[OA_LMULT_MULT];
            l1 = l1 * es.max_l;
            if (l > 1e-2f)
              for (int c = 0; c < 3; ++c) {
                eig[c] = eig[c] / l * l1;
                if (eig[c] > es.max_abs_eig[c]) eig[c] = es.max_abs_eig[c];
              }
          }
          for (int c = 0; c < 3; ++c) {
            float v = es.eigvec[c] * eig[0] + es.eigvec[3 + c] * eig[1] + es.eigvec[6 + c] * eig[2];
            v = v < max_multiplier ? v : max_multiplier;
            v = v > 0 ? v : 0;
            top[((size_t)n * 3 + c) * area + px] = v;
          }
        }
      }
    }

    if (has_chromatic) {                                                                                /* ColorContrastAugmentation, :72-116 */
      for (int n = 0; n < N; ++n) {
        const float* k = co[n].v;
        for (size_t px = 0; px < area; ++px) {
          float rgb[3], mean_in = 0, mean_out = 0;
          for (int c = 0; c < 3; ++c) {
            rgb[c] = top[((size_t)n * 3 + c) * area + px];
            mean_in += rgb[c];
            rgb[c] *= k[OA_COLOR1 + c];
            mean_out += rgb[c];
          }
          const float brightness_coeff = mean_in / (mean_out + 0.01f);
          for (int c = 0; c < 3; ++c) {
            rgb[c] = oa_clamp(rgb[c] * brightness_coeff, 0.f, 1.f);
            rgb[c] = powf(rgb[c], k[OA_GAMMA]);
            rgb[c] = rgb[c] + k[OA_BRIGHTNESS];
            rgb[c] = 0.5f + (rgb[c] - 0.5f) * k[OA_CONTRAST];
            top[((size_t)n * 3 + c) * area + px] = oa_clamp(rgb[c], 0.f, max_multiplier);
          }
        }
      }
    }

    if (has_effect) {                                                                                   /* ApplyEffects, :295-317 */
      for (int n = 0; n < N; ++n) {
        const float nx = (float)cos(co[n].v[OA_SHADOW_ANGLE]), ny = (float)sin(co[n].v[OA_SHADOW_ANGLE]);   /* hpp:110 */
        for (int c = 0; c < C; ++c)
          for (int y = 0; y < ch; ++y)
            for (int x = 0; x < cw; ++x) {
              float sample = top[(((size_t)n * C + c) * ch + y) * cw + x];
              if ((x - cw / 2) * nx + (y - ch / 2) * ny - co[n].v[OA_SHADOW_DISTANCE] > 0) sample -= co[n].v[OA_SHADOW_STRENGTH];
              top[(((size_t)n * C + c) * ch + y) * cw + x] = oa_clamp(sample, 0.f, max_multiplier);
            }
        /* the noise effect, :578-587: N(0, noise^2) per element.  The reference draws from cuRAND's stream; the HIP kernel (and this twin)
         * from Philox4x32-10 -- counter (pixel, sample * 4 + channel triple, stream), key = seed -- through Box-Muller. */
        const float sigma = co[n].v[OA_NOISE];
        if (sigma > 0)
          for (int y = 0; y < ch; ++y)
            for (int x = 0; x < cw; ++x) {
              const long long pix = (long long)y * cw + x;
              uint32_t r[4];
              oa_philox4x32_10((uint32_t)pix, (uint32_t)(pix >> 32) ^ ((uint32_t)n * 4u), (uint32_t)p->noise_stream, (uint32_t)(p->noise_stream >> 32),
                               (uint32_t)p->noise_seed, (uint32_t)(p->noise_seed >> 32), r);
              const float r0 = sqrtf(-2.0f * logf(oa_unit(r[0]))), t0 = 6.283185307179586f * oa_unit(r[1]);
              const float r1 = sqrtf(-2.0f * logf(oa_unit(r[2]))), t1 = 6.283185307179586f * oa_unit(r[3]);
              const float z[3] = {r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1)};
              for (int c = 0; c < 3; ++c) top[(((size_t)n * C + c) * ch + y) * cw + x] += sigma * z[c];
            }
      }
    }
    free(co);
    free(mats);
  }

  /* mean subtraction, :592-635: per pixel (axpy with -1, :613-616) or per channel (rank-1 gemm with -1, :617-634) */
  if (p->mean_mode != FN2_MEAN_NONE)
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c)
        for (size_t px = 0; px < area; ++px)
          top[(size_t)n * count + (size_t)c * area + px] -= (p->mean_mode == FN2_MEAN_PER_PIXEL) ? mean[(size_t)c * area + px] : mean[c];
  return FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * .caffemodel reader twin (checker for csrc/caffemodel.cpp): NetParameter.layer (100) / .layers (2) -> blobs, as
 * Net::CopyTrainedLayersFrom (net.cpp:752-800) and Blob::FromProto (blob.cpp:459-508) see them.  Written as a table of
 * (message, field) handlers over one generic field iterator -- a different shape from the product code on purpose.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const unsigned char* p; const unsigned char* end; } cm_span;

static int cm_varint(cm_span* s, uint64_t* v) {
  uint64_t r = 0; int shift = 0;
  while (s->p < s->end && shift < 64) {
    unsigned char b = *s->p++;
    r |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) { *v = r; return 1; }
    shift += 7;
  }
  return 0;
}

/* returns 0 at end, 1 on a field, -1 on malformed input; payload = value bytes for wire types 1, 2, 5 */
static int cm_next(cm_span* s, unsigned* num, unsigned* wt, uint64_t* val, cm_span* payload) {
  if (s->p >= s->end) return 0;
  uint64_t key, l;
  if (!cm_varint(s, &key) || (key >> 3) == 0 || (key >> 3) > 0x1fffffffull) return -1;
  *num = (unsigned)(key >> 3); *wt = (unsigned)(key & 7);
  if (*wt == 0) return cm_varint(s, val) ? 1 : -1;
  if (*wt == 1) l = 8; else if (*wt == 5) l = 4;
  else if (*wt == 2) { if (!cm_varint(s, &l)) return -1; }
  else return -1;
  if ((uint64_t)(s->end - s->p) < l) return -1;
  payload->p = s->p; payload->end = s->p + l; s->p += l;
  return 1;
}

static int cm_blob_meta(cm_span b, fn2_caffemodel_entry* e) {
  long long legacy[4] = {0, 0, 0, 0};
  int shape_seen = 0, legacy_seen = 0, rc;
  size_t nfloat = 0, ndouble = 0;
  unsigned num, wt; uint64_t v; cm_span pl;
  e->num_axes = 0;
  while ((rc = cm_next(&b, &num, &wt, &v, &pl)) == 1) {
    if (wt == 0 && num >= 1 && num <= 4) { legacy[num - 1] = (int32_t)v; legacy_seen = 1; }
    if (num == 7 && wt == 2) {
      unsigned sn, sw; uint64_t sv; cm_span spl; int src;
      shape_seen = 1;
      while ((src = cm_next(&pl, &sn, &sw, &sv, &spl)) == 1) {
        if (sn != 1) continue;
        if (sw == 0) { if (e->num_axes >= 8) return 0; e->dim[e->num_axes++] = (long long)sv; }
        if (sw == 2) while (spl.p < spl.end) { uint64_t x; if (!cm_varint(&spl, &x) || e->num_axes >= 8) return 0; e->dim[e->num_axes++] = (long long)x; }
      }
      if (src < 0) return 0;
    }
    if (num == 5) nfloat += wt == 2 ? (size_t)(pl.end - pl.p) / 4 : (wt == 5 ? 1 : 0);
    if (num == 8) ndouble += wt == 2 ? (size_t)(pl.end - pl.p) / 8 : (wt == 1 ? 1 : 0);
  }
  if (rc < 0) return 0;
  if (!shape_seen && legacy_seen) { e->num_axes = 4; memcpy(e->dim, legacy, sizeof(legacy)); }
  e->is_double = ndouble > 0;
  e->count = ndouble > 0 ? ndouble : nfloat;
  return 1;
}

FN2_API int fn2_caffemodel_index_cpu(const void* buf, size_t len, fn2_caffemodel_entry* entries, int max_entries, int* num_entries) {
  if (!buf || !num_entries || (max_entries > 0 && !entries)) return FN2_ERR_INVALID_ARG;
  const unsigned char* base = (const unsigned char*)buf;
  cm_span net = {base, base + len}, layer;
  unsigned num, wt; uint64_t v; int rc, count = 0;
  while ((rc = cm_next(&net, &num, &wt, &v, &layer)) == 1) {
    if (wt != 2 || (num != 100 && num != 2)) continue;
    const int v1 = num == 2;
    fn2_caffemodel_entry head;
    memset(&head, 0, sizeof(head));
    head.v1 = v1; head.v1_type = -1;
    const int first = count;
    int bi = 0, lrc;
    unsigned ln, lw; uint64_t lv; cm_span lp;
    while ((lrc = cm_next(&layer, &ln, &lw, &lv, &lp)) == 1) {
      if (lw == 2 && ln == (v1 ? 4u : 1u)) { head.name_off = (size_t)(lp.p - base); head.name_len = (size_t)(lp.end - lp.p); }
      else if (!v1 && lw == 2 && ln == 2) { head.type_off = (size_t)(lp.p - base); head.type_len = (size_t)(lp.end - lp.p); }
      else if (v1 && lw == 0 && ln == 5) head.v1_type = (long long)lv;
      else if (lw == 2 && ln == (v1 ? 6u : 7u)) {
        fn2_caffemodel_entry e;
        memset(&e, 0, sizeof(e));
        e.blob_index = bi++; e.blob_off = (size_t)(lp.p - base); e.blob_len = (size_t)(lp.end - lp.p);
        if (!cm_blob_meta(lp, &e)) return FN2_ERR_INVALID_ARG;
        if (count < max_entries) entries[count] = e;
        ++count;
      }
    }
    if (lrc < 0) return FN2_ERR_INVALID_ARG;
    for (int i = first; i < count && i < max_entries; ++i) {
      entries[i].name_off = head.name_off; entries[i].name_len = head.name_len; entries[i].type_off = head.type_off;
      entries[i].type_len = head.type_len; entries[i].v1 = head.v1; entries[i].v1_type = head.v1_type;
    }
  }
  if (rc < 0) return FN2_ERR_INVALID_ARG;
  *num_entries = count;
  return FN2_OK;
}

FN2_API int fn2_caffemodel_read_blob_cpu(const void* buf, size_t len, const fn2_caffemodel_entry* e, float* dst, size_t dst_floats) {
  if (!buf || !e || !dst || e->blob_off > len || e->blob_len > len - e->blob_off || dst_floats < e->count) return FN2_ERR_INVALID_ARG;
  long long want = 1;
  for (int i = 0; i < e->num_axes; ++i) want *= e->dim[i];
  if ((size_t)want != e->count) return FN2_ERR_INVALID_ARG;
  cm_span b = {(const unsigned char*)buf + e->blob_off, (const unsigned char*)buf + e->blob_off + e->blob_len}, pl;
  unsigned num, wt; uint64_t v; int rc; size_t k = 0;
  while ((rc = cm_next(&b, &num, &wt, &v, &pl)) == 1) {
    if (wt == 0) continue;
    if (!e->is_double && num == 5) for (; pl.p + 4 <= pl.end && k < e->count; pl.p += 4) memcpy(dst + k++, pl.p, 4);
    if (e->is_double && num == 8) for (; pl.p + 8 <= pl.end && k < e->count; pl.p += 8) { double d; memcpy(&d, pl.p, 8); dst[k++] = (float)d; }
  }
  return rc < 0 || k != e->count ? FN2_ERR_INVALID_ARG : FN2_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Channel-slice twins (round 3).  The *_slices entry points are the plain layers reading / writing channel ranges of wider blobs,
 * with the Eltwise layers the FlowNet2 graphs put around them folded in.  The twins restate that literally: gather the slice,
 * apply the Eltwise scaling (eltwise_layer.cpp:59-65: top = coeff * bottom, one rounding), run the plain layer twin, scatter.
 * ---------------------------------------------------------------------------------------------- */
static float* slice_gather(const float* blob, int N, int ctot, int c0, int C, size_t hw) {
  float* t = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * C * hw);
  if (!t) return NULL;
  for (int n = 0; n < N; ++n) memcpy(t + (size_t)n * C * hw, blob + ((size_t)n * ctot + c0) * hw, sizeof(float) * C * hw);
  return t;
}
static void slice_scatter(const float* t, float* blob, int N, int ctot, int c0, int C, size_t hw) {
  for (int n = 0; n < N; ++n) memcpy(blob + ((size_t)n * ctot + c0) * hw, t + (size_t)n * C * hw, sizeof(float) * C * hw);
}

FN2_API int fn2_flow_warp_forward_slices_cpu(const float* image, int image_channels, int image_c0,
                                             const float* flow, int flow_channels, int flow_c0,
                                             float* warped, int top_channels, int top_c0, int N, int C, int H, int W, int fill_value) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || image_c0 < 0 || image_c0 + C > image_channels || top_c0 < 0 || top_c0 + C > top_channels ||
      flow_c0 < 0 || flow_c0 + 2 > flow_channels)
    return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  float* im = slice_gather(image, N, image_channels, image_c0, C, hw);
  float* fl = slice_gather(flow, N, flow_channels, flow_c0, 2, hw);
  float* out = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * C * hw);
  if (!im || !fl || !out) { free(im); free(fl); free(out); return FN2_ERR_INVALID_ARG; }
  const int rc = fn2_flow_warp_forward_cpu(im, fl, out, N, C, H, W, fill_value);
  if (rc == FN2_OK) slice_scatter(out, warped, N, top_channels, top_c0, C, hw);
  free(im); free(fl); free(out);
  return rc;
}

FN2_API int fn2_channel_norm_forward_slices_cpu(const float* bottom, int bottom_channels, int bottom_c0,
                                                const float* minus, int minus_channels, int minus_c0,
                                                float* top, int top_channels, int top_c0, int N, int C, int H, int W) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || bottom_c0 < 0 || bottom_c0 + C > bottom_channels || top_c0 < 0 || top_c0 + 1 > top_channels ||
      (minus && (minus_c0 < 0 || minus_c0 + C > minus_channels)))
    return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  float* b = slice_gather(bottom, N, bottom_channels, bottom_c0, C, hw);
  float* t = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * hw);
  if (!b || !t) { free(b); free(t); return FN2_ERR_INVALID_ARG; }
  if (minus) {                                                        /* Eltwise{SUM, coeff 1, -1}: eltwise_layer.cpp:59-65 */
    for (int n = 0; n < N; ++n)
      for (size_t i = 0; i < (size_t)C * hw; ++i) b[(size_t)n * C * hw + i] -= minus[((size_t)n * minus_channels + minus_c0) * hw + i];
  }
  const int rc = fn2_channel_norm_forward_cpu(b, t, N, C, H, W);
  if (rc == FN2_OK) slice_scatter(t, top, N, top_channels, top_c0, 1, hw);
  free(b); free(t);
  return rc;
}

FN2_API int fn2_resample_forward_slices_cpu(const float* in, float in_scale, float* out, int top_channels, int top_c0,
                                            float* out2, int top2_channels, int top2_c0, float out2_scale,
                                            int N, int C, int Hin, int Win, int Hout, int Wout, int type, int antialias) {
  if (N < 0 || C < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1 || top_c0 < 0 || top_c0 + C > top_channels ||
      (out2 && (top2_c0 < 0 || top2_c0 + C > top2_channels)))
    return FN2_ERR_INVALID_ARG;
  const size_t hwi = (size_t)Hin * Win, hwo = (size_t)Hout * Wout, ni = (size_t)(N > 0 ? N : 1) * C * hwi, no = (size_t)(N > 0 ? N : 1) * C * hwo;
  float* s = (float*)malloc(sizeof(float) * ni);
  float* t = (float*)malloc(sizeof(float) * no);
  if (!s || !t) { free(s); free(t); return FN2_ERR_INVALID_ARG; }
  for (size_t i = 0; i < (size_t)N * C * hwi; ++i) s[i] = in[i] * in_scale;
  const int rc = fn2_resample_forward_cpu(s, t, N, C, Hin, Win, Hout, Wout, type, antialias);
  if (rc == FN2_OK) {
    slice_scatter(t, out, N, top_channels, top_c0, C, hwo);
    if (out2) {
      for (size_t i = 0; i < (size_t)N * C * hwo; ++i) t[i] = t[i] * out2_scale;
      slice_scatter(t, out2, N, top2_channels, top2_c0, C, hwo);
    }
  }
  free(s); free(t);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Backward of the 2-channel flow heads (CPU twins of csrc/flow_head_bwd.hip): ConvolutionLayer::Backward_cpu (conv_layer.cpp:42-70) with
 * weight [2, C, 3, 3] and DeconvolutionLayer::Backward_cpu (deconv_layer.cpp:28-60) with weight [2, 2, 4, 4], written as the sums they
 * are, accumulated in double (the HIP kernels use fixed-order fp32 partial sums: compared at 1e-5 * scale).
 * ---------------------------------------------------------------------------------------------- */
FN2_API int fn2_predict_flow_conv_backward_cpu(const float* bottom, int bottom_channels, int bottom_c0, const float* weight, const float* top_diff,
                                               float* bottom_diff, float* weight_diff, float* bias_diff, int N, int C, int H, int W, int accumulate) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || bottom_c0 < 0 || bottom_c0 + C > bottom_channels) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  if (weight_diff) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < 2; ++co)
      for (int c = 0; c < C; ++c)
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n)
              for (int y = 0; y < H; ++y) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= H) continue;
                for (int x = 0; x < W; ++x) {
                  const int xx = x + kx - 1;
                  if (xx < 0 || xx >= W) continue;
                  acc += (double)top_diff[((size_t)n * 2 + co) * hw + (size_t)y * W + x] *
                         (double)bottom[((size_t)n * bottom_channels + bottom_c0 + c) * hw + (size_t)yy * W + xx];
                }
              }
            float* d = weight_diff + (((size_t)co * C + c) * 3 + ky) * 3 + kx;
            *d = (accumulate ? *d : 0.f) + (float)acc;
          }
  }
  if (bias_diff)
    for (int co = 0; co < 2; ++co) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n)
        for (size_t i = 0; i < hw; ++i) acc += top_diff[((size_t)n * 2 + co) * hw + i];
      bias_diff[co] = (accumulate ? bias_diff[co] : 0.f) + (float)acc;
    }
  if (bottom_diff) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            double acc = 0.0;
            for (int co = 0; co < 2; ++co)
              for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                  const int yy = y - ky + 1, xx = x - kx + 1;
                  if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                  acc += (double)top_diff[((size_t)n * 2 + co) * hw + (size_t)yy * W + xx] * (double)weight[(((size_t)co * C + c) * 3 + ky) * 3 + kx];
                }
            bottom_diff[((size_t)n * C + c) * hw + (size_t)y * W + x] = (float)acc;
          }
  }
  return FN2_OK;
}

FN2_API int fn2_upsample_flow_deconv_backward_cpu(const float* bottom, const float* weight, const float* top_diff, float* bottom_diff,
                                                  float* weight_diff, float* bias_diff, int N, int H, int W, int accumulate) {
  if (N < 0 || H < 1 || W < 1) return FN2_ERR_INVALID_ARG;
  const int Ho = 2 * H, Wo = 2 * W;
  const size_t hw = (size_t)H * W, hwo = (size_t)Ho * Wo;
  if (weight_diff)
    for (int ci = 0; ci < 2; ++ci)
      for (int co = 0; co < 2; ++co)
        for (int ky = 0; ky < 4; ++ky)
          for (int kx = 0; kx < 4; ++kx) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n)
              for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                  const int Y = 2 * y - 1 + ky, X = 2 * x - 1 + kx;
                  if (Y < 0 || Y >= Ho || X < 0 || X >= Wo) continue;
                  acc += (double)bottom[((size_t)n * 2 + ci) * hw + (size_t)y * W + x] * (double)top_diff[((size_t)n * 2 + co) * hwo + (size_t)Y * Wo + X];
                }
            float* d = weight_diff + ((ci * 2 + co) * 4 + ky) * 4 + kx;
            *d = (accumulate ? *d : 0.f) + (float)acc;
          }
  if (bias_diff)
    for (int co = 0; co < 2; ++co) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n)
        for (size_t i = 0; i < hwo; ++i) acc += top_diff[((size_t)n * 2 + co) * hwo + i];
      bias_diff[co] = (accumulate ? bias_diff[co] : 0.f) + (float)acc;
    }
  if (bottom_diff)
    for (int n = 0; n < N; ++n)
      for (int ci = 0; ci < 2; ++ci)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            double acc = 0.0;
            for (int co = 0; co < 2; ++co)
              for (int ky = 0; ky < 4; ++ky)
                for (int kx = 0; kx < 4; ++kx) {
                  const int Y = 2 * y - 1 + ky, X = 2 * x - 1 + kx;
                  if (Y < 0 || Y >= Ho || X < 0 || X >= Wo) continue;
                  acc += (double)top_diff[((size_t)n * 2 + co) * hwo + (size_t)Y * Wo + X] * (double)weight[((ci * 2 + co) * 4 + ky) * 4 + kx];
                }
            bottom_diff[((size_t)n * 2 + ci) * hw + (size_t)y * W + x] = (float)acc;
          }
  return FN2_OK;
}

FN2_API int fn2_bias_leaky_relu_backward_slices_cpu(const float* top_data, const float* top_diff, int diff_channels, int diff_c0,
                                                    float* bottom_diff, float* bias_diff, int N, int C, int H, int W, float negative_slope) {
  if (N < 0 || C < 1 || H < 1 || W < 1 || diff_c0 < 0 || diff_c0 + C > diff_channels) return FN2_ERR_INVALID_ARG;
  const size_t hw = (size_t)H * W;
  float* g = slice_gather(top_diff, N, diff_channels, diff_c0, C, hw);
  if (!g) return FN2_ERR_INVALID_ARG;
  const int rc = fn2_bias_leaky_relu_backward_cpu(top_data, g, bottom_diff, bias_diff, N, C, H, W, negative_slope);
  free(g);
  return rc;
}
