// FlowAugmentation: the ground-truth flow field under the spatial augmentation applied to the two images.
//
// Reference: FlowAugmentationLayer (src/caffe/layers/flow_augmentation_layer.{cpp,cu}) and the matrix helpers of
// AugmentationLayerBase (src/caffe/layers/augmentation_layer_base.cpp:14-68, :352-380).  The reference builds the per-sample
// matrices on the host from the coefficient blobs, uploads them through two SyncedMemory buffers and launches WarpData; here the
// matrices travel as kernel arguments (12 floats per sample, 64 samples per launch): no workspace, no copy.
// HBM-bound: one gathered read of (u, v) and one coalesced write of (u', v') per output pixel.
#include "fn2_common.hpp"

#include <cmath>

namespace fn2 {

// tTransMat, include/caffe/layers/augmentation_layer_base.hpp:20-35:  | t0 t2 t4 |
//                                                                     | t1 t3 t5 |
struct TransMat {
  float t0, t1, t2, t3, t4, t5;
  void identity() { t0 = 1; t2 = 0; t4 = 0; t1 = 0; t3 = 1; t5 = 0; }                        // cpp:15-19
  void left_multiply(float u0, float u1, float u2, float u3, float u4, float u5) {           // cpp:22-35
    const float a0 = t0, a2 = t2, a4 = t4, a1 = t1, a3 = t3, a5 = t5;
    t0 = a0 * u0 + a1 * u2;
    t1 = a0 * u1 + a1 * u3;
    t2 = a2 * u0 + a3 * u2;
    t3 = a2 * u1 + a3 * u3;
    t4 = a4 * u0 + a5 * u2 + u4;
    t5 = a4 * u1 + a5 * u3 + u5;
  }
  TransMat inverse() const {                                                                 // cpp:52-68
    const float a = t0, c = t2, e = t4, b = t1, d = t3, f = t5;
    const float denom = a * d - b * c;
    TransMat r;
    r.t0 = d / denom;
    r.t1 = -b / denom;
    r.t2 = -c / denom;
    r.t3 = a / denom;
    r.t4 = (c * f - d * e) / denom;
    r.t5 = (b * e - a * f) / denom;
    return r;
  }
};

// array_to_coeff (cpp:368-380: fields with a non-zero default come back through exp) followed by fromCoeff (cpp:38-49).  After
// array_to_coeff every field is "set", so each factor is applied.  The double -> float conversions are the reference's: the
// arguments of leftMultiply are floats, its call sites compute them in double.
static TransMat matrix_from_coeffs(const float* in, int width, int height, int bottomwidth, int bottomheight) {
  const float mirror = in[0], dx = in[1], dy = in[2], angle = in[3];
  const float zoom_x = (float)std::exp((double)in[4]), zoom_y = (float)std::exp((double)in[5]);   // exp(Dtype) resolves to ::exp(double), cpp:376
  TransMat m;
  m.identity();
  if (mirror) m.left_multiply(-1, 0, 0, 1, (float)(.5 * (double)(float)width), (float)(-.5 * (double)(float)height));
  else m.left_multiply(1, 0, 0, 1, (float)(-.5 * (double)(float)width), (float)(-.5 * (double)(float)height));
  m.left_multiply((float)std::cos((double)angle), (float)std::sin((double)angle), (float)-std::sin((double)angle), (float)std::cos((double)angle), 0, 0);
  m.left_multiply(1, 0, 0, 1, dx * (float)width, dy * (float)height);
  m.left_multiply((float)(1.0 / (double)zoom_x), 0, 0, (float)(1.0 / (double)zoom_y), 0, 0);
  m.left_multiply(1, 0, 0, 1, (float)(.5 * (double)(float)bottomwidth), (float)(.5 * (double)(float)bottomheight));
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
