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

// ---- backward ---------------------------------------------------------------------------------------------------------
// bottom_diff = top_diff * (top_data > 0 ? 1 : slope)   (ReLUBackward on the in-place blob, relu_layer.cu:33-43: the sign of
// the output is the sign of the input for slope > 0) and bias_diff[c] = sum over n, h, w of bottom_diff (backward_gpu_bias,
// base_conv_layer.cpp:389-393: a GEMV with a vector of ones per sample) in one pass: every block reduces its chunk of one
// (n, c) plane (wave shuffles + LDS), a second small kernel adds the partials of a channel in a fixed order.
constexpr int kBwdChunks = 8;      // blocks per (n, c) plane

// top_diff may be a channel slice of a wider blob (the gradient of a Concat arrives as one: concat_layer.cu:62-90 hands every bottom its
// range of top_diff): plane (n, c) of it starts at ((n * dctot + dc0 + c) * hw)
// MASK = false: the bias gradient alone (a Convolution without a fused ReLU: top_data / bottom_diff unused).
template <bool MASK, bool VEC4>
__global__ void __launch_bounds__(256) bias_leaky_relu_bwd(const float* __restrict__ top_data, const float* __restrict__ top_diff,
                                                           float* __restrict__ bottom_diff, float* __restrict__ partial,
                                                           unsigned hw, float slope, int C, int dctot, int dc0, int yctot, int yc0) {
  __shared__ float red[4];
  const unsigned plane = blockIdx.y;
  const size_t base = (size_t)plane * hw;
  const size_t dbase = ((size_t)(plane / (unsigned)C) * dctot + dc0 + plane % (unsigned)C) * hw;
  const size_t ybase = ((size_t)(plane / (unsigned)C) * yctot + yc0 + plane % (unsigned)C) * hw;       // top_data may be a channel slice too
  float acc = 0.f;
  if constexpr (VEC4) {
    // 16-byte loads and stores (hw % 4 == 0, 16-byte aligned blobs: every plane then starts on a 16-byte boundary).  The per-thread sum
    // adds the four lanes of a quad in order: another summation order than the scalar form, fixed for a given geometry
    const unsigned n4 = hw / 4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
      f32x4 g = reinterpret_cast<const f32x4*>(top_diff + dbase)[i];
      if constexpr (MASK) {
        const f32x4 y = reinterpret_cast<const f32x4*>(top_data + ybase)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] *= y[j] > 0.f ? 1.f : slope;
        reinterpret_cast<f32x4*>(bottom_diff + base)[i] = g;
      }
      acc += (g[0] + g[1]) + (g[2] + g[3]);
    }
  } else {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < hw; i += gridDim.x * 256u) {
      float g = top_diff[dbase + i];
      if constexpr (MASK) {
        g *= top_data[ybase + i] > 0.f ? 1.f : slope;
        bottom_diff[base + i] = g;
      }
      acc += g;
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)plane * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Small maps (round 6): ONE workgroup per channel walks all N planes of its channel and finishes the bias gradient itself -- no partials,
// no second launch (bias_diff_finalize was 15 launches of 5-12 us on the critical path of a FlowNetC training step; the layers from 1/16
// resolution down have at most a few thousand values per channel).  Order: thread t sums elements t, t + 256, ... of the (sample, pixel)
// sequence in order, then the fixed butterfly below.
template <bool MASK, bool VEC4>
__global__ void __launch_bounds__(256) bias_leaky_relu_bwd_channel(const float* __restrict__ top_data, const float* __restrict__ top_diff,
                                                                   float* __restrict__ bottom_diff, float* __restrict__ bias_diff,
                                                                   int N, unsigned hw, float slope, int C, int dctot, int dc0, int yctot, int yc0,
                                                                   int accumulate) {
  __shared__ float red[4];
  const unsigned c = blockIdx.x;
  float acc = 0.f;
  if constexpr (VEC4) {
    const unsigned q = hw / 4, total = (unsigned)N * q;
    for (unsigned e = threadIdx.x; e < total; e += 256u) {
      const unsigned n = e / q, i = e - n * q;
      f32x4 g = reinterpret_cast<const f32x4*>(top_diff + ((size_t)n * dctot + dc0 + c) * hw)[i];
      if constexpr (MASK) {
        const f32x4 y = reinterpret_cast<const f32x4*>(top_data + ((size_t)n * yctot + yc0 + c) * hw)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] *= y[j] > 0.f ? 1.f : slope;
        reinterpret_cast<f32x4*>(bottom_diff + ((size_t)n * C + c) * hw)[i] = g;
      }
      acc += (g[0] + g[1]) + (g[2] + g[3]);
    }
  } else {
    const unsigned total = (unsigned)N * hw;
    for (unsigned e = threadIdx.x; e < total; e += 256u) {
      const unsigned n = e / hw, i = e - n * hw;
      float g = top_diff[((size_t)n * dctot + dc0 + c) * hw + i];
      if constexpr (MASK) {
        g *= top_data[((size_t)n * yctot + yc0 + c) * hw + i] > 0.f ? 1.f : slope;
        bottom_diff[((size_t)n * C + c) * hw + i] = g;
      }
      acc += g;
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && bias_diff) {
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    bias_diff[c] = accumulate ? bias_diff[c] + tot : tot;
  }
}

// one wave per channel: the N * chunks partials of the channel spread over the lanes (lane l takes partials l, l + 64, ... in order),
// then a fixed-shape butterfly -- deterministic; a thread per channel walked N * chunks dependent loads (9.6 us per call, 15 calls per
// FlowNetC training step)
__global__ void __launch_bounds__(256) bias_diff_finalize(const float* __restrict__ partial, float* __restrict__ bias_diff,
                                                          int N, int C, int chunks, int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const int total = N * chunks;
  float acc = 0.f;
  for (int i = lane; i < total; i += 64) {
    const int n = i / chunks, k = i - n * chunks;
    acc += partial[((size_t)n * C + c) * chunks + k];
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) bias_diff[c] = accumulate ? bias_diff[c] + acc : acc;
}

// one workgroup per channel (bias_leaky_relu_bwd_channel) where that fills the chip and a channel is short: a function of the geometry only
static inline bool small_map(int N, int C, long long hw) { return C >= 128 && (long long)N * hw <= 20000; }

}  // namespace fn2

using namespace fn2;

FN2_API size_t fn2_bias_leaky_relu_backward_workspace_bytes(int N, int C, int H, int W) {
  (void)H; (void)W;
  if (N <= 0 || C <= 0) return 0;
  return sizeof(float) * (size_t)N * C * kBwdChunks;
}

FN2_API int fn2_bias_leaky_relu_backward_slices(const float* top_data, const float* top_diff, int diff_channels, int diff_c0,
                                                float* bottom_diff, float* bias_diff, int N, int C, int H, int W, float negative_slope,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  return fn2_bias_leaky_relu_backward_slices2(top_data, C, 0, top_diff, diff_channels, diff_c0, bottom_diff, bias_diff, N, C, H, W, negative_slope,
                                              workspace, workspace_bytes, stream);
}

FN2_API int fn2_bias_leaky_relu_backward_slices2(const float* top_data, int data_channels, int data_c0, const float* top_diff, int diff_channels,
                                                 int diff_c0, float* bottom_diff, float* bias_diff, int N, int C, int H, int W, float negative_slope,
                                                 void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0) return fail(FN2_ERR_INVALID_ARG, "bias_leaky_relu_backward: bad shape [%d,%d,%d,%d]", N, C, H, W);
  if (diff_c0 < 0 || diff_c0 + C > diff_channels) return fail(FN2_ERR_INVALID_ARG, "bias_leaky_relu_backward: top_diff slice outside its blob");
  if (data_c0 < 0 || data_c0 + C > data_channels) return fail(FN2_ERR_INVALID_ARG, "bias_leaky_relu_backward: top_data slice outside its blob");
  if (N == 0) return FN2_OK;
  if (!top_data || !top_diff || !bottom_diff) return fail(FN2_ERR_INVALID_ARG, "bias_leaky_relu_backward: null blob");
  const long long planes = (long long)N * C, hw = (long long)H * W;
  if (planes > 65535 || hw >= (1ll << 32)) return fail(FN2_ERR_UNSUPPORTED, "bias_leaky_relu_backward: blob too large");
  const size_t need = fn2_bias_leaky_relu_backward_workspace_bytes(N, C, H, W);
  if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "bias_leaky_relu_backward: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  const bool vec = hw % 4 == 0 && ((reinterpret_cast<uintptr_t>(top_data) | reinterpret_cast<uintptr_t>(top_diff) | reinterpret_cast<uintptr_t>(bottom_diff)) & 15) == 0;
  if (small_map(N, C, hw)) {
    if (vec) hipLaunchKernelGGL((bias_leaky_relu_bwd_channel<true, true>), dim3((unsigned)C), dim3(256), 0, st, top_data, top_diff, bottom_diff, bias_diff, N,
                                (unsigned)hw, negative_slope, C, diff_channels, diff_c0, data_channels, data_c0, 0);
    else hipLaunchKernelGGL((bias_leaky_relu_bwd_channel<true, false>), dim3((unsigned)C), dim3(256), 0, st, top_data, top_diff, bottom_diff, bias_diff, N,
                            (unsigned)hw, negative_slope, C, diff_channels, diff_c0, data_channels, data_c0, 0);
    return check_launch("bias_leaky_relu_backward");
  }
  if (vec) hipLaunchKernelGGL((bias_leaky_relu_bwd<true, true>), dim3(kBwdChunks, (unsigned)planes), dim3(256), 0, st, top_data, top_diff, bottom_diff, partial,
                              (unsigned)hw, negative_slope, C, diff_channels, diff_c0, data_channels, data_c0);
  else hipLaunchKernelGGL((bias_leaky_relu_bwd<true, false>), dim3(kBwdChunks, (unsigned)planes), dim3(256), 0, st, top_data, top_diff, bottom_diff, partial,
                          (unsigned)hw, negative_slope, C, diff_channels, diff_c0, data_channels, data_c0);
  if (bias_diff) hipLaunchKernelGGL(bias_diff_finalize, dim3((C + 3) / 4), dim3(256), 0, st, partial, bias_diff, N, C, kBwdChunks, 0);
  return check_launch("bias_leaky_relu_backward");
}

FN2_API int fn2_conv_backward_bias(const float* top_diff, int diff_channels, int diff_c0, float* bias_diff, int N, int C, int H, int W,
                                   int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0) return fail(FN2_ERR_INVALID_ARG, "conv_backward_bias: bad shape [%d,%d,%d,%d]", N, C, H, W);
  if (diff_c0 < 0 || diff_c0 + C > diff_channels) return fail(FN2_ERR_INVALID_ARG, "conv_backward_bias: top_diff slice outside its blob");
  if (N == 0) return FN2_OK;
  if (!top_diff || !bias_diff) return fail(FN2_ERR_INVALID_ARG, "conv_backward_bias: null blob");
  const long long planes = (long long)N * C, hw = (long long)H * W;
  if (planes > 65535 || hw >= (1ll << 32)) return fail(FN2_ERR_UNSUPPORTED, "conv_backward_bias: blob too large");
  const size_t need = fn2_bias_leaky_relu_backward_workspace_bytes(N, C, H, W);
  if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "conv_backward_bias: workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  const bool vec = hw % 4 == 0 && (reinterpret_cast<uintptr_t>(top_diff) & 15) == 0;
  if (small_map(N, C, hw)) {
    if (vec) hipLaunchKernelGGL((bias_leaky_relu_bwd_channel<false, true>), dim3((unsigned)C), dim3(256), 0, st, nullptr, top_diff, nullptr, bias_diff, N,
                                (unsigned)hw, 1.0f, C, diff_channels, diff_c0, C, 0, accumulate);
    else hipLaunchKernelGGL((bias_leaky_relu_bwd_channel<false, false>), dim3((unsigned)C), dim3(256), 0, st, nullptr, top_diff, nullptr, bias_diff, N,
                            (unsigned)hw, 1.0f, C, diff_channels, diff_c0, C, 0, accumulate);
    return check_launch("conv_backward_bias");
  }
  if (vec) hipLaunchKernelGGL((bias_leaky_relu_bwd<false, true>), dim3(kBwdChunks, (unsigned)planes), dim3(256), 0, st, nullptr, top_diff, nullptr, partial,
                              (unsigned)hw, 1.0f, C, diff_channels, diff_c0, C, 0);
  else hipLaunchKernelGGL((bias_leaky_relu_bwd<false, false>), dim3(kBwdChunks, (unsigned)planes), dim3(256), 0, st, nullptr, top_diff, nullptr, partial,
                          (unsigned)hw, 1.0f, C, diff_channels, diff_c0, C, 0);
  hipLaunchKernelGGL(bias_diff_finalize, dim3((C + 3) / 4), dim3(256), 0, st, partial, bias_diff, N, C, kBwdChunks, accumulate);
  return check_launch("conv_backward_bias");
}

FN2_API int fn2_bias_leaky_relu_backward(const float* top_data, const float* top_diff, float* bottom_diff, float* bias_diff,
                                         int N, int C, int H, int W, float negative_slope, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  return fn2_bias_leaky_relu_backward_slices(top_data, top_diff, C, 0, bottom_diff, bias_diff, N, C, H, W, negative_slope, workspace,
                                             workspace_bytes, stream);
}

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

// ---- deploy head: top[n, top_c0 + c] = bottom[n, c] * scale + shift[c], TWO roundings (product, then sum: no fma contraction) -- the
// bits of Eltwise{coeff: 1/255} (eltwise_layer.cu:46-52: 0 + coeff * x) followed by the mean subtraction of the deploy-time
// DataAugmentation layer (data_augmentation_layer.cu:592-621: x - mean[c]) when the Resample between them is the identity, in one pass,
// written into a channel slice of the blob the first convolution reads (FlowNetS: [img0 | img1] along the channel axis).
namespace fn2 {
template <bool VEC4>
__global__ void __launch_bounds__(256) scale_shift(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ shift,
                                                   int C, unsigned hw, int out_ctot, int out_c0, float scale) {
  const unsigned plane = blockIdx.y;                       // n * C + c
  const unsigned n = plane / (unsigned)C, c = plane % (unsigned)C;
  const float sh = shift ? shift[c] : 0.f;
  const float* p = in + (size_t)plane * hw;
  float* q = out + ((size_t)n * out_ctot + out_c0 + c) * hw;
  if constexpr (VEC4) {                 // 16-byte loads / stores (hw % 4 == 0, 16-byte aligned blobs): 9.5 -> see profiles/r06_rocprof_summary.md
    const unsigned n4 = hw / 4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
      f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float prod = v[j] * scale;
        asm volatile("" : "+v"(prod));
        v[j] = prod + sh;
      }
      reinterpret_cast<f32x4*>(q)[i] = v;
    }
  } else {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
      float prod = p[i] * scale;
      asm volatile("" : "+v"(prod));       // the product is rounded on its own: hipcc contracts a * b + c into an fma otherwise (also through __fmul_rn)
      q[i] = prod + sh;
    }
  }
}
}  // namespace fn2

FN2_API int fn2_scale_shift_forward(const float* bottom, float* top, const float* shift, int N, int C, int H, int W,
                                    int top_channels, int top_c0, float scale, void* stream) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return fn2::fail(FN2_ERR_INVALID_ARG, "scale_shift: bad shape [%d,%d,%d,%d]", N, C, H, W);
  if (top_c0 < 0 || top_c0 + C > top_channels) return fn2::fail(FN2_ERR_INVALID_ARG, "scale_shift: channel slice outside the blob");
  if (N == 0) return FN2_OK;
  if (!bottom || !top) return fn2::fail(FN2_ERR_INVALID_ARG, "scale_shift: NULL blob pointer");
  if ((long long)N * C > 65535) return fn2::fail(FN2_ERR_UNSUPPORTED, "scale_shift: too many planes");
  const unsigned hw = (unsigned)H * (unsigned)W;
  const bool vec = hw % 4 == 0 && ((reinterpret_cast<uintptr_t>(bottom) | reinterpret_cast<uintptr_t>(top)) & 15) == 0;
  if (vec) hipLaunchKernelGGL(fn2::scale_shift<true>, dim3(fn2::blocks_for(hw / 4, 256, 256), (unsigned)(N * C)), dim3(256), 0, fn2::as_stream(stream), bottom, top,
                              shift, C, hw, top_channels, top_c0, scale);
  else hipLaunchKernelGGL(fn2::scale_shift<false>, dim3(fn2::blocks_for(hw, 256, 64), (unsigned)(N * C)), dim3(256), 0, fn2::as_stream(stream), bottom, top, shift,
                          C, hw, top_channels, top_c0, scale);
  return fn2::check_launch("scale_shift_forward");
}
