// Bias + leaky ReLU, in place, one pass over the convolution output (HBM-bound: 8 bytes per element).
//
// Reference: the bias term of ConvolutionLayer / DeconvolutionLayer (BaseConvolutionLayer::forward_gpu_bias,
// src/caffe/layers/base_conv_layer.cpp:343-348: top += bias[c]) followed by the in-place ReLULayer with
// negative_slope (ReLUForward, src/caffe/layers/relu_layer.cu:8-14: out = in > 0 ? in : in * negative_slope) --
// two launches and two passes over the blob there; every conv/deconv + ReLU pair of the FlowNet prototxts.
#include "fn2_common.hpp"

namespace fn2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// grid: (blocks over hw / 4, N * C); VEC4 when hw % 4 == 0 and the blob is 16-byte aligned
template <bool VEC4>
__global__ void __launch_bounds__(256) bias_leaky_relu(float* __restrict__ data, const float* __restrict__ bias,
                                                       int C, unsigned hw, float slope) {
  const unsigned plane = blockIdx.y;
  const float b = bias ? bias[plane % (unsigned)C] : 0.f;
  float* p = data + (size_t)plane * hw;
  if constexpr (VEC4) {
    const unsigned n4 = hw / 4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
      f32x4 v = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float t = v[j] + b; v[j] = t > 0.f ? t : t * slope; }
      reinterpret_cast<f32x4*>(p)[i] = v;
    }
  } else {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
      const float t = p[i] + b;
      p[i] = t > 0.f ? t : t * slope;
    }
  }
}

}  // namespace fn2

using namespace fn2;

FN2_API int fn2_bias_leaky_relu_forward(float* data, const float* bias, int N, int C, int H, int W, float negative_slope,
                                        void* stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0) return fail(FN2_ERR_INVALID_ARG, "bias_leaky_relu: bad shape [%d,%d,%d,%d]", N, C, H, W);
  if (N == 0) return FN2_OK;
  if (!data) return fail(FN2_ERR_INVALID_ARG, "bias_leaky_relu: null blob");
  const long long planes = (long long)N * C, hw = (long long)H * W;
  if (planes > 65535 * 32ll || hw >= (1ll << 32)) return fail(FN2_ERR_UNSUPPORTED, "bias_leaky_relu: blob too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool vec = hw % 4 == 0 && (reinterpret_cast<uintptr_t>(data) & 15) == 0;
  const unsigned per = (unsigned)(vec ? hw / 4 : hw);
  unsigned bx = (per + 255) / 256;
  if (bx > 64) bx = 64;
  // gridDim.y is limited to 65535: fold planes in chunks
  for (long long p0 = 0; p0 < planes; p0 += 65535) {
    const unsigned py = (unsigned)((planes - p0) < 65535 ? (planes - p0) : 65535);
    float* base = data + (size_t)p0 * hw;
    const float* bb = bias;     // plane % C below needs the chunk to start on a multiple of C ...
    if (p0 % C != 0) return fail(FN2_ERR_UNSUPPORTED, "bias_leaky_relu: N*C > 65535 with C not dividing 65535");
    if (vec) hipLaunchKernelGGL(bias_leaky_relu<true>, dim3(bx, py), dim3(256), 0, st, base, bb, C, (unsigned)hw, negative_slope);
    else     hipLaunchKernelGGL(bias_leaky_relu<false>, dim3(bx, py), dim3(256), 0, st, base, bb, C, (unsigned)hw, negative_slope);
  }
  return check_launch("bias_leaky_relu_forward");
}
