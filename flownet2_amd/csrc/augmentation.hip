// FlowAugmentation: the ground-truth flow field under the spatial augmentation applied to the two images.
//
// Reference: FlowAugmentationLayer (src/caffe/layers/flow_augmentation_layer.{cpp,cu}) and the matrix helpers of
// AugmentationLayerBase (src/caffe/layers/augmentation_layer_base.cpp:14-68, :352-380).  The reference builds the per-sample
// matrices on the host from the coefficient blobs, uploads them through two SyncedMemory buffers and launches WarpData; here the
// matrices travel as kernel arguments (12 floats per sample, 64 samples per launch): no workspace, no copy.
// HBM-bound: one gathered read of (u, v) and one coalesced write of (u', v') per output pixel.
#include "augmentation.hpp"

namespace fn2 {

// FlowAugmentationLayer::Forward_gpu, flow_augmentation_layer.cu:126-143: array_to_coeff (every field set) -> toIdentity -> fromCoeff
static TransMat matrix_from_coeffs(const float* in, int width, int height, int bottomwidth, int bottomheight) {
  AugCoeff c;
  c.from_array(in);
  TransMat m;
  m.identity();
  m.from_coeff(c, width, height, bottomwidth, bottomheight);
  return m;
}

constexpr int kAugChunk = 64;
struct FlowAugArgs {
  const float* flow;
  float* top;
  int n0, n_chunk, H, W, ch, cw;
  long long src_count;
  TransMat m1[kAugChunk], m2[kAugChunk];
};

// WarpData, flow_augmentation_layer.cu:23-88.  Thread per output pixel, x fastest.
__global__ void __launch_bounds__(256) flow_aug_warp(FlowAugArgs a) {
  const long long per = (long long)a.ch * a.cw;
  const long long total = per * a.n_chunk;
  for (long long index = blockIdx.x * (long long)blockDim.x + threadIdx.x; index < total; index += (long long)gridDim.x * blockDim.x) {
    const int xi = (int)(index % a.cw), yi = (int)((index / a.cw) % a.ch), k = (int)(index / per);
    const int n = a.n0 + k;
    const float x = (float)xi, y = (float)yi;
    const TransMat& m1 = a.m1[k];
    const TransMat& m2 = a.m2[k];
    const float xpos1 = x * m1.t0 + y * m1.t2 + m1.t4;                                        // :41-42
    const float ypos1 = x * m1.t1 + y * m1.t3 + m1.t5;
    const long long off = (long long)(int)(ypos1 + 0.5f) * a.W + (int)(xpos1 + 0.5f);         // :45-50, flat index
    const long long ix = (long long)a.W * a.H * (2 * n + 0) + off;
    const long long iy = (long long)a.W * a.H * (2 * n + 1) + off;
    const float u = (ix >= 0 && ix < a.src_count) ? a.flow[ix] : 0.f;                         // reference: min(idx, count), unchecked below 0
    const float v = (iy >= 0 && iy < a.src_count) ? a.flow[iy] : 0.f;
    const float xpos2 = xpos1 + u, ypos2 = ypos1 + v;                                         // :52-53
    const float xpos3 = xpos2 * m2.t0 + ypos2 * m2.t2 + m2.t4;                                // :56-57
    const float ypos3 = xpos2 * m2.t1 + ypos2 * m2.t3 + m2.t5;
    a.top[((long long)(2 * n + 0) * a.ch + yi) * a.cw + xi] = xpos3 - x;                      // :60-61
    a.top[((long long)(2 * n + 1) * a.ch + yi) * a.cw + xi] = ypos3 - y;
  }
}

}  // namespace fn2

using namespace fn2;

FN2_API int fn2_augmentation_matrix(const float* coeffs, int crop_width, int crop_height, int bottom_width, int bottom_height,
                                    int invert, float* mat6) {
  if (!coeffs || !mat6) return fail(FN2_ERR_INVALID_ARG, "augmentation_matrix: NULL pointer");
  if (crop_width < 1 || crop_height < 1 || bottom_width < 1 || bottom_height < 1) return fail(FN2_ERR_INVALID_ARG, "augmentation_matrix: sizes must be positive");
  TransMat m = matrix_from_coeffs(coeffs, crop_width, crop_height, bottom_width, bottom_height);
  if (invert) m = m.inverse();
  mat6[0] = m.t0; mat6[1] = m.t1; mat6[2] = m.t2; mat6[3] = m.t3; mat6[4] = m.t4; mat6[5] = m.t5;
  return FN2_OK;
}

FN2_API int fn2_flow_augmentation_forward(const float* flow, const float* coeffs1, const float* coeffs2, float* top,
                                          int N, int H, int W, int crop_height, int crop_width, void* stream) {
  if (crop_width < 1) return fail(FN2_ERR_INVALID_ARG, "Please enter crop width if you want to perform augmentation");     // cpp:33
  if (crop_height < 1) return fail(FN2_ERR_INVALID_ARG, "Please enter crop height if you want to perform augmentation");   // cpp:34
  if (N < 0 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "flow_augmentation: bad flow shape [%d,2,%d,%d]", N, H, W);
  if (N == 0) return FN2_OK;
  if (!flow || !coeffs1 || !coeffs2 || !top) return fail(FN2_ERR_INVALID_ARG, "flow_augmentation: NULL pointer");
  FlowAugArgs a;
  a.flow = flow; a.top = top; a.H = H; a.W = W; a.ch = crop_height; a.cw = crop_width;
  a.src_count = (long long)N * 2 * H * W;
  for (int n0 = 0; n0 < N; n0 += kAugChunk) {
    a.n0 = n0;
    a.n_chunk = N - n0 < kAugChunk ? N - n0 : kAugChunk;
    for (int k = 0; k < a.n_chunk; ++k) {                                                    // cu:131-143
      a.m1[k] = matrix_from_coeffs(coeffs1 + (size_t)(n0 + k) * FN2_AUG_NUM_PARAMS, crop_width, crop_height, W, H);
      a.m2[k] = matrix_from_coeffs(coeffs2 + (size_t)(n0 + k) * FN2_AUG_NUM_PARAMS, crop_width, crop_height, W, H).inverse();
    }
    const long long total = (long long)a.n_chunk * crop_height * crop_width;
    hipLaunchKernelGGL(flow_aug_warp, dim3(blocks_for(total, 256)), dim3(256), 0, as_stream(stream), a);
  }
  return check_launch("flow_augmentation_forward");
}
