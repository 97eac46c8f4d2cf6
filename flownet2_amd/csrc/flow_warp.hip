// FlowWarp for gfx950: bilinear backward-warp of an image by a flow field.
//
// Replaces FlowWarpLayer::Forward_gpu / Backward_gpu (reference:
// src/caffe/layers/flow_warp_layer.cu:357-514).  The reference first transposes the image to NHWC
// (flow_warp_rearrange_kernel, :23-52) and memsets the output (:377); here one thread owns one
// (n, y, x) pixel, computes the four tap addresses / weights once, and walks the channels of the
// NCHW planes directly: per channel the wave reads four nearly-contiguous runs and writes one
// fully-coalesced run, so HBM traffic is the algorithmic 4*N*H*W*(2C+2) bytes.
#include "fn2_common.hpp"

namespace fn2 {

// Channels handled by one thread.  Small so that C=256 feature maps still fill the chip.
constexpr int kWarpChPerThread = 4;

// grid: (pixel blocks, n * cgroups + channel group); 32-bit pixel index (a 64-bit flat index cost more than the warp itself)
// image / flow / warped may be channel slices of wider blobs: sample n starts ictot (fctot, octot) planes after sample n - 1, the
// slice at plane ic0 (fc0, oc0).
template <bool SLICED>     // false: the plain layer (dense [N,C,H,W] image / [N,2,H,W] flow / top): its own instantiation, the indexing of rounds 1-2
__global__ void __launch_bounds__(256) flow_warp_fwd(const float* __restrict__ image, const float* __restrict__ flow,
                                                     float* __restrict__ warped, int N, int C, int H, int W,
                                                     int cgroups, float fill, int ictot, int ic0, int octot, int oc0, int fctot, int fc0) {
  const unsigned wh = (unsigned)H * W;
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= wh) return;
  const int y = pix / W, x = pix - y * W;
  for (unsigned g = blockIdx.y; g < (unsigned)N * cgroups; g += gridDim.y) {
    const int n = g / cgroups, cg = g - n * cgroups;
    const float x2 = (float)x + flow[(SLICED ? (size_t)n * fctot + fc0 : (size_t)(2 * n)) * wh + pix];           // flow_warp_layer.cu:73
    const float y2 = (float)y + flow[(SLICED ? (size_t)n * fctot + fc0 + 1 : (size_t)(2 * n + 1)) * wh + pix];   // :74
    const int c0 = cg * kWarpChPerThread;
    const int c1 = min(C, c0 + kWarpChPerThread);
    float* out = warped + (SLICED ? (size_t)n * octot + oc0 : (size_t)n * C) * wh + pix;
    if (x2 >= 0.f && y2 >= 0.f && x2 < (float)W && y2 < (float)H) {    // :108
      const int ixL = (int)x2, iyT = (int)y2;                           // :81-82
      const int ixR = min(ixL + 1, W - 1), iyB = min(iyT + 1, H - 1);   // :83-84
      const float alpha = x2 - ixL, beta = y2 - iyT;                    // :91-92
      const float cTL = (1 - alpha) * (1 - beta), cTR = alpha * (1 - beta);   // :93-96
      const float cBL = (1 - alpha) * beta, cBR = alpha * beta;
      const unsigned oTL = (unsigned)iyT * W + ixL, oTR = (unsigned)iyT * W + ixR;
      const unsigned oBL = (unsigned)iyB * W + ixL, oBR = (unsigned)iyB * W + ixR;
      const float* im = image + (SLICED ? (size_t)n * ictot + ic0 : (size_t)n * C) * wh;
      for (int c = c0; c < c1; ++c) {
        const float* p = im + (size_t)c * wh;
        // :110-114, contracted like nvcc does: mul + 3 fma
        out[(size_t)c * wh] = fmaf(cBR, p[oBR], fmaf(cBL, p[oBL], fmaf(cTR, p[oTR], cTL * p[oTL])));
      }
    } else {
      for (int c = c0; c < c1; ++c) out[(size_t)c * wh] = fill;          // :103 via the smem buffer
    }
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------
// Reference: flow_warp_backward_kernel_no_smem, flow_warp_layer.cu:169-229 -- one thread per pixel, four float
// atomicAdds per (pixel, channel) into the image gradient.  The L2 atomic units retire ~30 G lane-atomics/s on this
// chip however local the targets are (565 us for a [4,3,384,768] blob, 630 us for [4,256,48,96]).  Here the scatter
// is inverted once per PIXEL instead of per (pixel, channel):
//   1. warp_bwd_link   every in-image source pixel hangs itself on the list of its top-left tap's cell: ONE int
//                      atomicExch per pixel on a head array (N*H*W ints; entry = source pixel index);
//   2. warp_bwd_gather one thread per (target pixel, channel group) collects the sources on the lists of its own cell
//                      and of the cells left / above / above-left (a source whose top-left tap is there may reach this
//                      cell with another tap), sorts them (up to 24 per cell), recomputes their bilinear weights from
//                      their flow and writes the image gradient with plain stores -- a FIXED summation order wherever
//                      at most 24 sources meet: bit-reproducible there, which the reference never is;
//   3. warp_bwd_flow   the flow gradient (:203-227) is a gather already: thread per (pixel, channel group).
struct WarpTap {
  bool inside;
  unsigned o[4];      // TL, TR, BL, BR offsets inside the plane
  float w[4];         // their bilinear weights, association as in :197-200
  float gx, gy;
};

__device__ __forceinline__ WarpTap warp_taps(int x, int y, float fu, float fv, int H, int W) {
  WarpTap t;
  const float x2 = (float)x + fu, y2 = (float)y + fv;
  t.inside = x2 >= 0.f && y2 >= 0.f && x2 < (float)W && y2 < (float)H;
  const int ixL = (int)x2, iyT = (int)y2;
  const int ixR = min(ixL + 1, W - 1), iyB = min(iyT + 1, H - 1);
  const float alpha = x2 - ixL, beta = y2 - iyT;
  t.o[0] = (unsigned)iyT * W + ixL; t.o[1] = (unsigned)iyT * W + ixR;
  t.o[2] = (unsigned)iyB * W + ixL; t.o[3] = (unsigned)iyB * W + ixR;
  t.w[0] = (1 - alpha) * (1 - beta); t.w[1] = alpha * (1 - beta);
  t.w[2] = (1 - alpha) * beta;       t.w[3] = alpha * beta;
  t.gy = iyB - y2;   // :203
  t.gx = ixR - x2;   // :216
  return t;
}

// grid: (pixel blocks, n).  head[] is pre-set to -1.
__global__ void __launch_bounds__(256) warp_bwd_link(const float* __restrict__ flow, int* __restrict__ head, int* __restrict__ next,
                                                     int N, int H, int W) {
  const unsigned wh = (unsigned)H * W;
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= wh) return;
  const int y = pix / W, x = pix - y * W;
  for (unsigned n = blockIdx.y; n < (unsigned)N; n += gridDim.y) {
    const WarpTap t = warp_taps(x, y, flow[(size_t)(2 * n) * wh + pix], flow[(size_t)(2 * n + 1) * wh + pix], H, W);
    if (t.inside) next[(size_t)n * wh + pix] = atomicExch(head + (size_t)n * wh + t.o[0], (int)pix);
  }
}

constexpr int kWarpGatherThreads = 128;
constexpr int kWarpListMax = 24;      // sources sorted in LDS per target; longer lists fall back to selection passes

// grid: (pixel blocks of 128, n * cgroups + channel group)
// with_flow (round 6, image-like blobs with ONE channel group): the thread also computes the flow gradient of its own pixel -- the
// arithmetic of warp_bwd_flow, flow_warp_layer.cu:203-227 -- whose loads are in flight while the lists are chased: one launch less and the
// gather's idle latency used (84.9 -> see profiles/r06_layer_microbench.md)
__global__ void __launch_bounds__(kWarpGatherThreads) warp_bwd_gather(const float* __restrict__ flow, const float* __restrict__ warped_diff,
                                                                      const int* __restrict__ head, const int* __restrict__ next,
                                                                      float* __restrict__ image_diff, int N, int C, int H, int W,
                                                                      int cgroups, int cpg, const float* __restrict__ image,
                                                                      float* __restrict__ flow_diff, int with_flow) {
  __shared__ int srcs[kWarpListMax][kWarpGatherThreads];
  const unsigned wh = (unsigned)H * W;
  const unsigned pix = blockIdx.x * (unsigned)kWarpGatherThreads + threadIdx.x;
  if (pix >= wh) return;
  const int y = pix / W, x = pix - y * W;
  const int tid = threadIdx.x;
  for (unsigned g = blockIdx.y; g < (unsigned)N * cgroups; g += gridDim.y) {
    const int n = g / cgroups, cg = g - n * cgroups;
    const int c0 = cg * cpg, c1 = min(C, c0 + cpg);
    const int* hd = head + (size_t)n * wh;
    const int* nx = next + (size_t)n * wh;
    const float* fl = flow + (size_t)(2 * n) * wh;
    float du = 0.f, dv = 0.f;
    if (with_flow) {
      const WarpTap t = warp_taps(x, y, fl[pix], fl[wh + pix], H, W);
      if (t.inside) {                                                      // diffs are 0 otherwise (:478-479)
        for (int c = 0; c < C; ++c) {
          const size_t ch = ((size_t)n * C + c) * wh;
          const float g0 = warped_diff[ch + pix];
          const float* p = image + ch;
          const float TL = p[t.o[0]], TR = p[t.o[1]], BL = p[t.o[2]], BR = p[t.o[3]];
          float tu = 0.f;
          tu += t.gy * (TR - TL);
          tu += (1 - t.gy) * (BR - BL);
          du += g0 * tu;                                                   // :211
          float tv = 0.f;
          tv += t.gx * (BL - TL);
          tv += (1 - t.gx) * (BR - TR);
          dv += g0 * tv;                                                   // :225
        }
      }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    auto add_source = [&](int sp) {
      const int sy = sp / W, sx = sp - sy * W;
      const WarpTap t = warp_taps(sx, sy, fl[sp], fl[wh + sp], H, W);
      float w = 0.f;                       // taps of this source that land on this cell, TL..BR order
#pragma unroll
      for (int k = 0; k < 4; ++k) if (t.o[k] == pix) w += t.w[k];
      const float* gsrc = warped_diff + ((size_t)n * C + c0) * wh + sp;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (c0 + j < c1) acc[j] += gsrc[(size_t)j * wh] * w;
    };
    // collect the candidate sources: lists of the cells (y, x), (y, x-1), (y-1, x), (y-1, x-1)
    // (the four list heads are loaded together and the lists walked side by side: one memory latency per list LEVEL instead of one per
    // list element -- the order the candidates arrive in does not matter, they are sorted below)
    int cnt = 0;
    bool overflow = false;
    int e[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int cy = y - (d >> 1), cx = x - (d & 1);
      e[d] = (cy < 0 || cx < 0) ? -1 : hd[(unsigned)cy * W + cx];
    }
    while ((e[0] & e[1] & e[2] & e[3]) != -1 && (e[0] >= 0 || e[1] >= 0 || e[2] >= 0 || e[3] >= 0)) {
      int nxt[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) nxt[d] = e[d] >= 0 ? nx[e[d]] : -1;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        if (e[d] >= 0) {
          if (cnt < kWarpListMax) srcs[cnt][tid] = e[d]; else overflow = true;
          ++cnt;
        }
#pragma unroll
      for (int d = 0; d < 4; ++d) e[d] = nxt[d];
    }
    if (!overflow) {
      for (int i = 1; i < cnt; ++i) {                  // insertion sort, ascending source index
        const int v = srcs[i][tid];
        int j = i - 1;
        while (j >= 0 && srcs[j][tid] > v) { srcs[j + 1][tid] = srcs[j][tid]; --j; }
        srcs[j + 1][tid] = v;
      }
      for (int i = 0; i < cnt; ++i) add_source(srcs[i][tid]);
    } else {
      // more than kWarpListMax sources collapse onto this cell (a sink of the flow field): walk the lists once, in
      // list order -- for these cells only, the summation order depends on the order the atomics were served
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int cy = y - (d >> 1), cx = x - (d & 1);
        if (cy < 0 || cx < 0) continue;
        for (int e = hd[(unsigned)cy * W + cx]; e >= 0; e = nx[e]) add_source(e);
      }
    }
    float* dst = image_diff + ((size_t)n * C + c0) * wh + pix;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (c0 + j < c1) dst[(size_t)j * wh] = acc[j];
    if (with_flow) {
      flow_diff[(size_t)(2 * n) * wh + pix] = du;
      flow_diff[(size_t)(2 * n + 1) * wh + pix] = dv;
    }
  }
}

// grid: (pixel blocks, n * cgroups + channel group); with more than one group the partial sums meet in atomics
__global__ void __launch_bounds__(256) warp_bwd_flow(const float* __restrict__ image, const float* __restrict__ flow,
                                                     const float* __restrict__ warped_diff, float* __restrict__ flow_diff,
                                                     int N, int C, int H, int W, int cgroups, int cpg) {
  const unsigned wh = (unsigned)H * W;
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= wh) return;
  const int y = pix / W, x = pix - y * W;
  for (unsigned g = blockIdx.y; g < (unsigned)N * cgroups; g += gridDim.y) {
    const int n = g / cgroups, cg = g - n * cgroups;
    const WarpTap t = warp_taps(x, y, flow[(size_t)(2 * n) * wh + pix], flow[(size_t)(2 * n + 1) * wh + pix], H, W);
    if (!t.inside) {                                                     // diffs are 0 (:478-479)
      if (cgroups == 1) { flow_diff[(size_t)(2 * n) * wh + pix] = 0.f; flow_diff[(size_t)(2 * n + 1) * wh + pix] = 0.f; }      // (no memset in front of this form)
      continue;
    }
    float du = 0.f, dv = 0.f;
    const int c0 = cg * cpg, c1 = min(C, c0 + cpg);
    for (int c = c0; c < c1; ++c) {
      const size_t ch = ((size_t)n * C + c) * wh;
      const float g0 = warped_diff[ch + pix];
      const float* p = image + ch;
      const float TL = p[t.o[0]], TR = p[t.o[1]], BL = p[t.o[2]], BR = p[t.o[3]];
      float tu = 0.f;
      tu += t.gy * (TR - TL);
      tu += (1 - t.gy) * (BR - BL);
      du += g0 * tu;                                                       // :211
      float tv = 0.f;
      tv += t.gx * (BL - TL);
      tv += (1 - t.gx) * (BR - TR);
      dv += g0 * tv;                                                       // :225
    }
    float* fu = flow_diff + (size_t)(2 * n) * wh + pix;
    float* fv = flow_diff + (size_t)(2 * n + 1) * wh + pix;
    if (cgroups == 1) { *fu = du; *fv = dv; }
    else { unsafeAtomicAdd(fu, du); unsafeAtomicAdd(fv, dv); }
  }
}

}  // namespace fn2

using namespace fn2;

static int warp_check(const char* what, int N, int C, int H, int W) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "%s: bad shape [%d,%d,%d,%d]", what, N, C, H, W);
  return FN2_OK;
}

FN2_API int fn2_flow_warp_forward_slices(const float* image, int image_channels, int image_c0,
                                         const float* flow, int flow_channels, int flow_c0, float* warped, int top_channels, int top_c0, int N, int C, int H, int W,
                                         int fill_value, void* stream) {
  int rc = warp_check("flow_warp_forward", N, C, H, W);
  if (rc) return rc;
  if (fill_value != FN2_FILL_ZERO && fill_value != FN2_FILL_NAN)
    return fail(FN2_ERR_INVALID_ARG, "flow_warp_forward: fill_value must be ZERO(1) or NOT_A_NUMBER(2)");
  if (image_c0 < 0 || image_c0 + C > image_channels || top_c0 < 0 || top_c0 + C > top_channels || flow_c0 < 0 || flow_c0 + 2 > flow_channels)
    return fail(FN2_ERR_INVALID_ARG, "flow_warp_forward: channel slice [%d,+%d) of %d / [%d,+2) of %d / [%d,+%d) of %d", image_c0, C, image_channels,
                flow_c0, flow_channels, top_c0, C, top_channels);
  if (N == 0) return FN2_OK;
  if (!image || !flow || !warped) return fail(FN2_ERR_INVALID_ARG, "flow_warp_forward: NULL blob pointer");
  if ((long long)H * W >= (1ll << 31)) return fail(FN2_ERR_UNSUPPORTED, "flow_warp_forward: plane too large");
  const int cgroups = (C + kWarpChPerThread - 1) / kWarpChPerThread;
  const float fill = (fill_value == FN2_FILL_ZERO) ? 0.f : __builtin_bit_cast(float, 0xFFE00000u);   // flow_warp_layer.cu:372-375
  const long long groups = (long long)N * cgroups;
  const dim3 grid((unsigned)(((long long)H * W + 255) / 256), (unsigned)(groups < 65535 ? groups : 65535));
  const bool sliced = image_channels != C || top_channels != C || flow_channels != 2;
  if (sliced) hipLaunchKernelGGL((flow_warp_fwd<true>), grid, dim3(256), 0, as_stream(stream), image, flow, warped, N, C, H, W, cgroups, fill,
                                 image_channels, image_c0, top_channels, top_c0, flow_channels, flow_c0);
  else hipLaunchKernelGGL((flow_warp_fwd<false>), grid, dim3(256), 0, as_stream(stream), image, flow, warped, N, C, H, W, cgroups, fill,
                          image_channels, image_c0, top_channels, top_c0, flow_channels, flow_c0);
  return check_launch("flow_warp_forward");
}

FN2_API int fn2_flow_warp_forward(const float* image, const float* flow, float* warped, int N, int C, int H, int W,
                                  int fill_value, void* stream) {
  return fn2_flow_warp_forward_slices(image, C, 0, flow, 2, 0, warped, C, 0, N, C, H, W, fill_value, stream);
}

FN2_API size_t fn2_flow_warp_backward_workspace_bytes(int N, int C, int H, int W) {
  (void)C;
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return sizeof(int) * (size_t)N * H * W * 2;      // list heads + next links, one of each per pixel
}

FN2_API int fn2_flow_warp_backward(const float* image, const float* flow, const float* warped_diff, float* image_diff,
                                   float* flow_diff, int N, int C, int H, int W, int propagate_image, int propagate_flow,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  int rc = warp_check("flow_warp_backward", N, C, H, W);
  if (rc) return rc;
  if (!image || !flow || !warped_diff || !image_diff || !flow_diff)
    return fail(FN2_ERR_INVALID_ARG, "flow_warp_backward: NULL blob pointer");
  if (N == 0) return FN2_OK;
  if ((long long)H * W >= (1ll << 29)) return fail(FN2_ERR_UNSUPPORTED, "flow_warp_backward: plane too large");
  if (!workspace || workspace_bytes < fn2_flow_warp_backward_workspace_bytes(N, C, H, W))
    return fail(FN2_ERR_WORKSPACE, "flow_warp_backward: workspace of %zu bytes needed (fn2_flow_warp_backward_workspace_bytes)",
                fn2_flow_warp_backward_workspace_bytes(N, C, H, W));
  hipStream_t st = as_stream(stream);
  const size_t wh = (size_t)H * W;
  const unsigned bx = (unsigned)((wh + 255) / 256);
  const unsigned ny = (unsigned)(N < 65535 ? N : 65535);
  if (propagate_image) {
    int* head = reinterpret_cast<int*>(workspace);
    int* next = head + (size_t)N * wh;
    if (hipMemsetAsync(head, 0xff, sizeof(int) * (size_t)N * wh, st) != hipSuccess)       // every list empty (-1)
      return fail(FN2_ERR_LAUNCH, "flow_warp_backward: hipMemsetAsync failed");
    hipLaunchKernelGGL(warp_bwd_link, dim3(bx, ny), dim3(256), 0, st, flow, head, next, N, H, W);
    const int cpg = C < 8 ? C : 8;
    const int cgroups = (C + cpg - 1) / cpg;
    const long long groups = (long long)N * cgroups;
    const int with_flow = propagate_flow && cgroups == 1 && groups <= 65535;
    hipLaunchKernelGGL(warp_bwd_gather, dim3((unsigned)((wh + kWarpGatherThreads - 1) / kWarpGatherThreads), (unsigned)(groups < 65535 ? groups : 65535)),
                       dim3(kWarpGatherThreads), 0, st, flow, warped_diff, head, next, image_diff, N, C, H, W, cgroups, cpg, image, flow_diff, with_flow);
    if (with_flow) return check_launch("flow_warp_backward");
  } else {
    if (hipMemsetAsync(image_diff, 0, sizeof(float) * wh * C * N, st) != hipSuccess)       // :507
      return fail(FN2_ERR_LAUNCH, "flow_warp_backward: hipMemsetAsync failed");
  }
  // channels per thread: all of them for image-like blobs (plain stores: every pixel is written, no memset), 8 for feature blobs
  const int cpg = C <= 16 ? C : 8;
  const int cgroups = (C + cpg - 1) / cpg;
  if (!propagate_flow || cgroups > 1 || N > 65535)                                           // :479 / :508
    if (hipMemsetAsync(flow_diff, 0, sizeof(float) * wh * 2 * N, st) != hipSuccess)
      return fail(FN2_ERR_LAUNCH, "flow_warp_backward: hipMemsetAsync failed");
  if (propagate_flow) {
    const long long groups = (long long)N * cgroups;
    hipLaunchKernelGGL(warp_bwd_flow, dim3(bx, (unsigned)(groups < 65535 ? groups : 65535)), dim3(256), 0, st, image, flow, warped_diff,
                       flow_diff, N, C, H, W, cgroups, cpg);
  }
  return check_launch("flow_warp_backward");
}
