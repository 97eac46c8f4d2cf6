// Correlation (cost volume) for gfx950.
//
// Replaces CorrelationLayer::Forward_gpu / Backward_gpu (reference:
// src/caffe/layers/correlation_layer.cu:431-603) behind fn2_correlation_{forward,backward}.
//
// Unlike the reference there is no padded NHWC scratch copy (blob_rearrange_kernel2 + two
// cudaMemsets, correlation_layer.cu:447-458): every kernel reads the NCHW inputs directly and
// treats out-of-image positions as the zero padding.
//
// Kernels in this file
//   corr_fwd_generic / corr_bwd{0,1}_generic : any (kernel_size, stride_1, stride_2, pad, type);
//       one thread per output element, x fastest (coalesced).  Fallback + cross-check.
//   corr_fwd_mfma (correlation_mfma.hip), corr_bwd_mfma (correlation_bwd_mfma.hip): the FlowNetC fast paths,
//       kernel_size 1, stride_1 1, MULTIPLY: 2-D banded GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32).
#include "correlation.hpp"

#include <cmath>

namespace fn2 {

// CorrelationLayer::LayerSetUp + Reshape, correlation_layer.cpp:13-84.
int corr_geometry(const fn2_corr_params* p, int N, int C, int H, int W, CorrGeom* g) {
  if (!p) return fail(FN2_ERR_INVALID_ARG, "correlation: params == NULL");
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "correlation: bad bottom shape [%d,%d,%d,%d]", N, C, H, W);
  if (p->kernel_size < 1 || p->kernel_size % 2 == 0)
    return fail(FN2_ERR_INVALID_ARG, "correlation: Odd kernel size required (got %d)", p->kernel_size);
  if (p->stride1 < 1 || p->stride2 < 1) return fail(FN2_ERR_INVALID_ARG, "correlation: strides must be >= 1");
  if (p->max_displacement < 0 || p->pad < 0) return fail(FN2_ERR_INVALID_ARG, "correlation: negative pad / max_displacement");
  if (p->corr_type != FN2_CORR_MULTIPLY && p->corr_type != FN2_CORR_SUBTRACT)
    return fail(FN2_ERR_INVALID_ARG, "correlation: unknown correlation_type %d", p->corr_type);
  g->N = N; g->C = C; g->H = H; g->W = W;
  g->pad = p->pad; g->K = p->kernel_size; g->md = p->max_displacement; g->s1 = p->stride1; g->s2 = p->stride2;
  g->type = p->corr_type;
  g->relu = 0; g->slope = 0.f; g->top_c0 = 0;
  g->kr = (g->K - 1) / 2;
  const int border = g->md + g->kr;
  g->topW = (int)std::ceil((float)(W + 2 * g->pad - border * 2) / (float)g->s1);
  g->topH = (int)std::ceil((float)(H + 2 * g->pad - border * 2) / (float)g->s1);
  if (g->topW < 1 || g->topH < 1)
    return fail(FN2_ERR_INVALID_ARG, "Correlation cannot be done with current settings. Neighborhood and kernel don't fit in blob");
  g->ngr = g->md / g->s2;
  g->ngw = 2 * g->ngr + 1;
  g->topC = g->ngw * g->ngw;
  g->top_ctot = g->topC;
  if (g->pad < g->md)
    return fail(FN2_ERR_INVALID_ARG, "correlation: pad (%d) < max_displacement (%d) reads outside the padded blob in the reference; refused", g->pad, g->md);
  return FN2_OK;
}

// ---------------------------------------------------------------------------------------------
// Generic forward: thread per top element.
// ---------------------------------------------------------------------------------------------
template <bool SUB>
__global__ void __launch_bounds__(256) corr_fwd_generic(const float* __restrict__ b0, const float* __restrict__ b1,
                                                        float* __restrict__ top, CorrGeom g) {
  const long long total = (long long)g.N * g.topC * g.topH * g.topW;
  const size_t plane = (size_t)g.H * g.W;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % g.topW);
    const int y = (int)((idx / g.topW) % g.topH);
    const int tc = (int)((idx / g.topW / g.topH) % g.topC);
    const int n = (int)(idx / g.topW / g.topH / g.topC);
    const int s2o = (tc % g.ngw - g.ngr) * g.s2;
    const int s2p = (tc / g.ngw - g.ngr) * g.s2;
    const float* a_n = b0 + (size_t)n * g.C * plane;
    const float* b_n = b1 + (size_t)n * g.C * plane;
    float sum = 0.f;
    for (int j = 0; j < g.K; ++j) {
      const int ya = y * g.s1 + g.md + j - g.pad, yb = ya + s2p;
      for (int i = 0; i < g.K; ++i) {
        const int xa = x * g.s1 + g.md + i - g.pad, xb = xa + s2o;
        const bool a_in = (ya >= 0) & (ya < g.H) & (xa >= 0) & (xa < g.W);
        const bool b_in = (yb >= 0) & (yb < g.H) & (xb >= 0) & (xb < g.W);
        const float* ap = a_n + (size_t)ya * g.W + xa;
        const float* bp = b_n + (size_t)yb * g.W + xb;
        if (!SUB) {
          if (a_in && b_in)
            for (int c = 0; c < g.C; ++c) sum = fmaf(ap[c * plane], bp[c * plane], sum);
        } else {
          for (int c = 0; c < g.C; ++c) {
            const float av = a_in ? ap[c * plane] : 0.f;
            const float bv = b_in ? bp[c * plane] : 0.f;
            sum += fabsf(av - bv);
          }
        }
      }
    }
    float v = sum / (float)(g.K * g.K * g.C);
    if (g.relu) v = v > 0.f ? v : v * g.slope;
    top[(((size_t)n * g.top_ctot + g.top_c0 + tc) * g.topH + y) * g.topW + x] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Generic backward: thread per bottom element (x fastest), gather over displacements -- no atomics.
// Window arithmetic as CorrelateDataBackward0/1 (correlation_layer.cu:131-150, :212-224).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ceil_div(int a, int s) { return (a >= 0) ? (a + s - 1) / s : -((-a) / s); }
__device__ __forceinline__ int floor_div(int a, int s) { return (a >= 0) ? a / s : -((-a + s - 1) / s); }

__device__ __forceinline__ float padded_at(const float* plane_ptr, int m, int l, const CorrGeom& g) {
  const int y = m - g.pad, x = l - g.pad;
  return ((y >= 0) & (y < g.H) & (x >= 0) & (x < g.W)) ? plane_ptr[(size_t)y * g.W + x] : 0.f;
}

template <bool SUB, int WHICH>
__global__ void __launch_bounds__(256) corr_bwd_generic(const float* __restrict__ b0, const float* __restrict__ b1,
                                                        const float* __restrict__ top_diff,
                                                        float* __restrict__ bdiff, CorrGeom g) {
  const long long total = (long long)g.N * g.C * g.H * g.W;
  const size_t plane = (size_t)g.H * g.W;
  const size_t tplane = (size_t)g.topH * g.topW;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % g.W);
    const int y = (int)((idx / g.W) % g.H);
    const int c = (int)((idx / g.W / g.H) % g.C);
    const int n = (int)(idx / g.W / g.H / g.C);
    const int l = x + g.pad, m = y + g.pad;
    const float* p0 = b0 + ((size_t)n * g.C + c) * plane;
    const float* p1 = b1 + ((size_t)n * g.C + c) * plane;
    const float* td = top_diff + (size_t)n * g.topC * tplane;
    float sum = 0.f;
    for (int pp = -g.ngr; pp <= g.ngr; ++pp) {
      for (int o = -g.ngr; o <= g.ngr; ++o) {
        const int s2o = g.s2 * o, s2p = g.s2 * pp;
        const int sx = (WHICH == 0) ? 0 : s2o, sy = (WHICH == 0) ? 0 : s2p;
        int xmin = ceil_div(l - 2 * g.kr - g.md - sx, g.s1);
        int ymin = ceil_div(m - 2 * g.kr - g.md - sy, g.s1);
        int xmax = floor_div(l - g.md - sx, g.s1);
        int ymax = floor_div(m - g.md - sy, g.s1);
        if (!(xmax >= 0 && ymax >= 0 && xmin <= g.topW - 1 && ymin <= g.topH - 1)) continue;
        xmin = max(0, xmin); xmax = min(g.topW - 1, xmax);
        ymin = max(0, ymin); ymax = min(g.topH - 1, ymax);
        const int mm = (WHICH == 0) ? m + s2p : m - s2p;
        const int ll = (WHICH == 0) ? l + s2o : l - s2o;
        float coef;
        if (!SUB) {
          coef = padded_at((WHICH == 0) ? p1 : p0, mm, ll, g);
        } else {
          const float v0 = padded_at(p0, mm, ll, g), v1 = padded_at(p1, mm, ll, g);
          coef = (WHICH == 0) ? ((v0 >= v1) ? 1.f : -1.f) : ((v0 >= v1) ? -1.f : 1.f);
        }
        const float* t = td + (size_t)((pp + g.ngr) * g.ngw + (o + g.ngr)) * tplane;
        for (int yy = ymin; yy <= ymax; ++yy)
          for (int xx = xmin; xx <= xmax; ++xx) sum = fmaf(t[(size_t)yy * g.topW + xx], coef, sum);
      }
    }
    bdiff[idx] = sum / (float)((g.kr * 2 + 1) * (g.kr * 2 + 1) * g.C);
  }
}

// Set by the tests / bench through fn2_debug_set_correlation_impl: 0 = auto, 1 = force generic.
static int g_force_generic = 0;

}  // namespace fn2

using namespace fn2;

namespace fn2 { extern int g_corr_units; extern int g_corr_units_lds; extern int g_corr_units_abl; extern int g_corr_ablation; extern int g_corr_force_dword; extern int g_corr_proj; extern int g_corr_skip_dead; extern int g_corr_simd_plan; extern int g_corr1d_force_generic; extern int g_corr1d_no_mfma; extern unsigned long long* g_corr_dbg; namespace bwd { extern int g_corr_bwd_first_gen; extern int g_corr_bwd_gen; extern int g_corr_bwd_separate; } }

FN2_API int fn2_debug_set_correlation_trace(void* device_buffer) {
  fn2::g_corr_dbg = reinterpret_cast<unsigned long long*>(device_buffer);
  return FN2_OK;
}

// impl: 0 = automatic, 1 = generic kernels, 3 = general (dword LDS-DMA) MFMA forward even where the paired-parity kernel applies,
// 5 / 6 = first / second generation of the MFMA backward, 7 / 8 / 9 = profiling builds of the paired-parity forward (3/8 of the MFMAs, no
// MFMAs, half the staging: wrong results), 13 = paired-parity forward without the SIMD plan, 14 = no zero-fill workgroups (profiling, wrong
// output), 64 + bits = ablation of the general MFMA forward (FN2_ABLATION builds: 1 no MFMA, 2 no staging loads, 4 no stores);
// 19 = corr_fwd_pair (second generation) where the unit kernel applies, 20 + policy = the unit kernel with a task policy
// (correlation_units.hip: 0 automatic, k = tasks per image row, + 16 image order), 60 + policy = the same with 16 KB of extra LDS (two workgroups per CU)
FN2_API int fn2_debug_set_correlation_impl(int impl) {
  fn2::g_corr_units = impl == 19 ? 0 : (impl >= 20 && impl < 52) ? 1 + (impl - 20) : 1;
  fn2::g_corr_units_lds = impl == 60 ? 16384 : impl == 61 ? 65536 : 0;          // 60 / 61: two / one workgroup per CU (extra dynamic LDS)
  fn2::g_corr_units_abl = (impl >= 100 && impl < 164) ? impl - 100 : 0;
  g_force_generic = (impl == 1);
  fn2::g_corr1d_force_generic = (impl == 1);
  fn2::g_corr1d_no_mfma = (impl == 17);                      // Correlation1D: the LDS-tiled VALU forward instead of the MFMA one
  fn2::g_corr_force_dword = (impl == 3);
  fn2::g_corr_proj = impl == 7 ? 1 : impl == 8 ? 2 : impl == 9 ? 3 : 0;   // profiling builds of the paired-parity forward: 7 = 3/8 of the MFMAs (bf16 x 3 projection), 8 = none (wrong results)
  fn2::bwd::g_corr_bwd_first_gen = (impl == 5);
  fn2::bwd::g_corr_bwd_gen = (impl == 6) ? 2 : (impl == 15) ? 3 : 0;      // 15 = third generation (G through LDS, one slab ahead) where the fourth applies;        // 6 = second-generation MFMA backward (LDS-DMA staging, gathered G)       // 5 = first-generation (register-staged) MFMA backward where the LDS-DMA one applies
  fn2::bwd::g_corr_bwd_separate = (impl == 16);           // 16 = one backward launch per bottom where the merged launch applies
  fn2::g_corr_skip_dead = (impl == 14);                  // 14 (profiling, wrong output): no zero-fill workgroups
  fn2::g_corr_simd_plan = (impl != 13);                  // 13 = corr_fwd_pair without the SIMD plan (wave w takes patch column w)
  fn2::g_corr_ablation = (impl >= 64 && impl < 100) ? impl - 64 : 0;
  return FN2_OK;
}

FN2_API int fn2_debug_correlation_units_plan(int N, int H, int W, int policy, unsigned* out_words, int max_words) {
  return fn2::corr_fwd_units_plan_words(N, H, W, policy, out_words, max_words);
}

FN2_API int fn2_correlation_out_shape(const fn2_corr_params* p, int C, int H, int W, int* topC, int* topH, int* topW) {
  CorrGeom g;
  int rc = corr_geometry(p, 1, C, H, W, &g);
  if (rc) return rc;
  if (topC) *topC = g.topC;
  if (topH) *topH = g.topH;
  if (topW) *topW = g.topW;
  return FN2_OK;
}

FN2_API size_t fn2_correlation_workspace_bytes(const fn2_corr_params*, int, int, int, int) { return 0; }

FN2_API int fn2_correlation_forward(const fn2_corr_params* p, const float* bottom0, const float* bottom1, float* top,
                                    int N, int C, int H, int W, void* ws, size_t ws_bytes, void* stream) {
  return fn2_correlation_forward_fused(p, bottom0, bottom1, top, N, C, H, W, 0, 0, 0, 0.f, ws, ws_bytes, stream);
}

FN2_API int fn2_correlation_forward_fused(const fn2_corr_params* p, const float* bottom0, const float* bottom1, float* top,
                                          int N, int C, int H, int W, int top_channels, int top_c0, int relu, float negative_slope,
                                          void*, size_t, void* stream) {
  CorrGeom g;
  int rc = corr_geometry(p, N, C, H, W, &g);
  if (rc) return rc;
  if (top_channels > 0) {
    if (top_c0 < 0 || top_c0 + g.topC > top_channels) return fail(FN2_ERR_INVALID_ARG, "correlation_forward: channel slice outside the top blob");
    g.top_ctot = top_channels; g.top_c0 = top_c0;
  }
  g.relu = relu != 0; g.slope = negative_slope;
  if (N == 0) return FN2_OK;
  if (!bottom0 || !bottom1 || !top) return fail(FN2_ERR_INVALID_ARG, "correlation_forward: NULL blob pointer");
  hipStream_t st = as_stream(stream);
  const bool plain = fn2::g_corr_force_dword == 0 && fn2::g_corr_proj == 0 && fn2::g_corr_ablation == 0 && fn2::g_corr_skip_dead == 0 && fn2::g_corr_simd_plan != 0;
  if (!g_force_generic && plain && corr_fwd_units_supported(g, bottom0, bottom1, top)) return corr_fwd_units_launch(g, bottom0, bottom1, top, st);
  if (!g_force_generic && corr_fwd_mfma_supported(g)) return corr_fwd_mfma_launch(g, bottom0, bottom1, top, st);
  const long long total = (long long)N * g.topC * g.topH * g.topW;
  const unsigned blocks = blocks_for(total, 256);
  if (g.type == FN2_CORR_MULTIPLY)
    hipLaunchKernelGGL(corr_fwd_generic<false>, dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top, g);
  else
    hipLaunchKernelGGL(corr_fwd_generic<true>, dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top, g);
  return check_launch("correlation_forward");
}

FN2_API int fn2_correlation_backward(const fn2_corr_params* p, const float* bottom0, const float* bottom1,
                                     const float* top_diff, float* bottom0_diff, float* bottom1_diff,
                                     int N, int C, int H, int W, void*, size_t, void* stream) {
  CorrGeom g;
  int rc = corr_geometry(p, N, C, H, W, &g);
  if (rc) return rc;
  if (N == 0) return FN2_OK;
  if (!bottom0 || !bottom1 || !top_diff) return fail(FN2_ERR_INVALID_ARG, "correlation_backward: NULL blob pointer");
  hipStream_t st = as_stream(stream);
  if (!g_force_generic && corr_bwd_mfma_supported(g)) {
    if (bottom0_diff && bottom1_diff) {
      rc = corr_bwd_mfma_launch_both(g, bottom0, bottom1, top_diff, bottom0_diff, bottom1_diff, st);
      if (rc != FN2_ERR_UNSUPPORTED) return rc;
    }
    if (bottom0_diff) { rc = corr_bwd_mfma_launch(g, 0, bottom1, top_diff, bottom0_diff, st); if (rc) return rc; }
    if (bottom1_diff) { rc = corr_bwd_mfma_launch(g, 1, bottom0, top_diff, bottom1_diff, st); if (rc) return rc; }
    return FN2_OK;
  }
  const long long total = (long long)N * C * H * W;
  const unsigned blocks = blocks_for(total, 256);
  const bool sub = (g.type == FN2_CORR_SUBTRACT);
  if (bottom0_diff) {
    if (!sub) hipLaunchKernelGGL((corr_bwd_generic<false, 0>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom0_diff, g);
    else hipLaunchKernelGGL((corr_bwd_generic<true, 0>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom0_diff, g);
  }
  if (bottom1_diff) {
    if (!sub) hipLaunchKernelGGL((corr_bwd_generic<false, 1>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom1_diff, g);
    else hipLaunchKernelGGL((corr_bwd_generic<true, 1>), dim3(blocks), dim3(256), 0, st, bottom0, bottom1, top_diff, bottom1_diff, g);
  }
  return check_launch("correlation_backward");
}
