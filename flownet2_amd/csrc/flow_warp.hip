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
__global__ void __launch_bounds__(256) flow_warp_fwd(const float* __restrict__ image, const float* __restrict__ flow,
                                                     float* __restrict__ warped, int N, int C, int H, int W,
                                                     int cgroups, float fill) {
  const unsigned wh = (unsigned)H * W;
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= wh) return;
  const int y = pix / W, x = pix - y * W;
  for (unsigned g = blockIdx.y; g < (unsigned)N * cgroups; g += gridDim.y) {
    const int n = g / cgroups, cg = g - n * cgroups;
    const float x2 = (float)x + flow[(size_t)(2 * n) * wh + pix];       // flow_warp_layer.cu:73
    const float y2 = (float)y + flow[(size_t)(2 * n + 1) * wh + pix];   // :74
    const int c0 = cg * kWarpChPerThread;
    const int c1 = min(C, c0 + kWarpChPerThread);
    float* out = warped + ((size_t)n * C) * wh + pix;
    if (x2 >= 0.f && y2 >= 0.f && x2 < (float)W && y2 < (float)H) {    // :108
      const int ixL = (int)x2, iyT = (int)y2;                           // :81-82
      const int ixR = min(ixL + 1, W - 1), iyB = min(iyT + 1, H - 1);   // :83-84
      const float alpha = x2 - ixL, beta = y2 - iyT;                    // :91-92
      const float cTL = (1 - alpha) * (1 - beta), cTR = alpha * (1 - beta);   // :93-96
      const float cBL = (1 - alpha) * beta, cBR = alpha * beta;
      const unsigned oTL = (unsigned)iyT * W + ixL, oTR = (unsigned)iyT * W + ixR;
      const unsigned oBL = (unsigned)iyB * W + ixL, oBR = (unsigned)iyB * W + ixR;
      const float* im = image + ((size_t)n * C) * wh;
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

// flow_warp_backward_kernel_no_smem, flow_warp_layer.cu:169-229.  The image-diff scatter uses the hardware fp32 atomic
// add (global_atomic_add_f32) like the reference's atomicAdd (:197-200), so the summation order -- and the last bit --
// is not deterministic.  Thread = one pixel x one group of channels (the reference walks all channels in one thread:
// 18k threads for a [4,256,48,96] blob); with more than one group the flow gradient is accumulated atomically as well.
__global__ void __launch_bounds__(256) flow_warp_bwd(const float* __restrict__ image, const float* __restrict__ flow,
                                                     const float* __restrict__ warped_diff,
                                                     float* __restrict__ image_diff, float* __restrict__ flow_diff,
                                                     int N, int C, int H, int W, int cgroups, int cpg) {
  const unsigned wh = (unsigned)H * W;
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= wh) return;
  const int y = pix / W, x = pix - y * W;
  for (unsigned g = blockIdx.y; g < (unsigned)N * cgroups; g += gridDim.y) {
    const int n = g / cgroups, cg = g - n * cgroups;
    const float x2 = (float)x + flow[(size_t)(2 * n) * wh + pix];
    const float y2 = (float)y + flow[(size_t)(2 * n + 1) * wh + pix];
    if (!(x2 >= 0.f && y2 >= 0.f && x2 < (float)W && y2 < (float)H)) continue;   // diffs stay 0 (:478-479)
    const int ixL = (int)x2, iyT = (int)y2;
    const int ixR = min(ixL + 1, W - 1), iyB = min(iyT + 1, H - 1);
    const float alpha = x2 - ixL, beta = y2 - iyT;
    const unsigned oTL = (unsigned)iyT * W + ixL, oTR = (unsigned)iyT * W + ixR;
    const unsigned oBL = (unsigned)iyB * W + ixL, oBR = (unsigned)iyB * W + ixR;
    const float gy = iyB - y2;   // :203
    const float gx = ixR - x2;   // :216
    float du = 0.f, dv = 0.f;
    const int c0 = cg * cpg, c1 = min(C, c0 + cpg);
    for (int c = c0; c < c1; ++c) {
      const size_t ch = ((size_t)n * C + c) * wh;
      const float g0 = warped_diff[ch + pix];
      float* d = image_diff + ch;
      unsafeAtomicAdd(d + oTL, g0 * (1 - alpha) * (1 - beta));
      unsafeAtomicAdd(d + oTR, g0 * alpha * (1 - beta));
      unsafeAtomicAdd(d + oBL, g0 * (1 - alpha) * beta);
      unsafeAtomicAdd(d + oBR, g0 * alpha * beta);
      const float* p = image + ch;
      const float TL = p[oTL], TR = p[oTR], BL = p[oBL], BR = p[oBR];
      float tu = 0.f;
      tu += gy * (TR - TL);
      tu += (1 - gy) * (BR - BL);
      du += g0 * tu;                                                       // :211
      float tv = 0.f;
      tv += gx * (BL - TL);
      tv += (1 - gx) * (BR - TR);
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

FN2_API int fn2_flow_warp_forward(const float* image, const float* flow, float* warped, int N, int C, int H, int W,
                                  int fill_value, void* stream) {
  int rc = warp_check("flow_warp_forward", N, C, H, W);
  if (rc) return rc;
  if (fill_value != FN2_FILL_ZERO && fill_value != FN2_FILL_NAN)
    return fail(FN2_ERR_INVALID_ARG, "flow_warp_forward: fill_value must be ZERO(1) or NOT_A_NUMBER(2)");
  if (N == 0) return FN2_OK;
  if (!image || !flow || !warped) return fail(FN2_ERR_INVALID_ARG, "flow_warp_forward: NULL blob pointer");
  if ((long long)H * W >= (1ll << 31)) return fail(FN2_ERR_UNSUPPORTED, "flow_warp_forward: plane too large");
  const int cgroups = (C + kWarpChPerThread - 1) / kWarpChPerThread;
  const float fill = (fill_value == FN2_FILL_ZERO) ? 0.f : __builtin_bit_cast(float, 0xFFE00000u);   // flow_warp_layer.cu:372-375
  const long long groups = (long long)N * cgroups;
  const dim3 grid((unsigned)(((long long)H * W + 255) / 256), (unsigned)(groups < 65535 ? groups : 65535));
  hipLaunchKernelGGL(flow_warp_fwd, grid, dim3(256), 0, as_stream(stream), image, flow, warped, N, C, H, W, cgroups, fill);
  return check_launch("flow_warp_forward");
}

FN2_API int fn2_flow_warp_backward(const float* image, const float* flow, const float* warped_diff, float* image_diff,
                                   float* flow_diff, int N, int C, int H, int W, int propagate_image, int propagate_flow,
                                   void* stream) {
  int rc = warp_check("flow_warp_backward", N, C, H, W);
  if (rc) return rc;
  if (!image || !flow || !warped_diff || !image_diff || !flow_diff)
    return fail(FN2_ERR_INVALID_ARG, "flow_warp_backward: NULL blob pointer");
  if (N == 0) return FN2_OK;
  hipStream_t st = as_stream(stream);
  const size_t wh = (size_t)H * W;
  if (hipMemsetAsync(image_diff, 0, sizeof(float) * wh * C * N, st) != hipSuccess ||      // :478
      hipMemsetAsync(flow_diff, 0, sizeof(float) * wh * 2 * N, st) != hipSuccess)         // :479
    return fail(FN2_ERR_LAUNCH, "flow_warp_backward: hipMemsetAsync failed");
  if ((long long)H * W >= (1ll << 31)) return fail(FN2_ERR_UNSUPPORTED, "flow_warp_backward: plane too large");
  // channels per thread: all of them for image-like blobs (plain flow-gradient stores), 8 for feature blobs
  const int cpg = C <= 16 ? C : 8;
  const int cgroups = (C + cpg - 1) / cpg;
  const long long groups = (long long)N * cgroups;
  const dim3 grid((unsigned)(((long long)H * W + 255) / 256), (unsigned)(groups < 65535 ? groups : 65535));
  hipLaunchKernelGGL(flow_warp_bwd, grid, dim3(256), 0, st, image, flow, warped_diff, image_diff, flow_diff, N, C, H, W, cgroups, cpg);
  rc = check_launch("flow_warp_backward");
  if (rc) return rc;
  if (!propagate_image) (void)hipMemsetAsync(image_diff, 0, sizeof(float) * wh * C * N, st);   // :507
  if (!propagate_flow) (void)hipMemsetAsync(flow_diff, 0, sizeof(float) * wh * 2 * N, st);     // :508
  return FN2_OK;
}
