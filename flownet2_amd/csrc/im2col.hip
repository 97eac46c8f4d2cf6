// im2col / col2im for gfx950: the data-movement halves of Caffe's GEMM convolution, batched over the mini-batch.
//
// Reference: im2col_gpu / col2im_gpu (src/caffe/util/im2col.cu:8-72, 246-318), called once per SAMPLE by
// BaseConvolutionLayer::forward_gpu_gemm (conv: im2col + SGEMM, base_conv_layer.cpp:325-341) and
// backward_gpu_gemm (deconv forward: SGEMM + col2im, base_conv_layer.cpp:352-368, deconv_layer.cu:8-23).
// Here one launch covers the whole batch, so that the GEMM in between is one batched library call; the layers
// that take this route are the ones where the library's direct convolutions are weak on this chip (3x3 stride-2 and
// small-map layers 35-68 TFLOP/s, 4x4/2 deconvs 45-54 TFLOP/s, against 85-125 TFLOP/s for the plain fp32 GEMM of
// the same shape).  col2im carries the deconvolution's bias term and the following in-place leaky ReLU, i.e.
// forward_gpu_bias (base_conv_layer.cpp:343-348) + ReLUForward (relu_layer.cu:8-14), in the same pass.
// Both kernels are HBM-bound streams: bytes = col matrix + image, once.
#include "fn2_common.hpp"

namespace fn2 {

struct ColArgs {
  int C, H, W;        // image blob (per sample)
  int Hc, Wc;         // column grid
  int k, pad, stride;
  float slope;
  int relu;
  int im_ctot, im_c0; // col2im: the image blob may be a channel slice [c0, c0 + C) of a blob with ctot channels
};

// grid: (ceil(Hc*Wc / 256), N * C).  Thread = one column position of one channel: k*k loads from a k x k window
// (neighbouring threads share them through L1/L2), k*k stores each coalesced along the column index.
template <int K>
__global__ void __launch_bounds__(256) im2col_kernel(const float* __restrict__ im, float* __restrict__ col, ColArgs a) {
  const int kk = K > 0 ? K : a.k;
  const unsigned hwc = (unsigned)a.Hc * a.Wc;
  const unsigned plane = blockIdx.y;                       // n * C + c
  const float* src = im + (size_t)plane * a.H * a.W;
  float* dst = col + (size_t)plane * kk * kk * hwc;
  for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < hwc; p += gridDim.x * blockDim.x) {
    const int yc = p / a.Wc, xc = p % a.Wc;
    const int y0 = yc * a.stride - a.pad, x0 = xc * a.stride - a.pad;
    for (int i = 0; i < kk; ++i) {       // K > 0: constant trip counts, fully unrolled by the compiler
      const int y = y0 + i;
      for (int j = 0; j < kk; ++j) {
        const int x = x0 + j;
        const bool in = y >= 0 && y < a.H && x >= 0 && x < a.W;
        dst[(size_t)(i * kk + j) * hwc + p] = in ? src[(size_t)y * a.W + x] : 0.f;
      }
    }
  }
}

// grid: (ceil(H*W / 256), N * C).  Thread = one image element: adds the column entries that map onto it, rows of the
// column grid ascending, then columns ascending (the reference kernel's order), then bias and the optional leaky ReLU.
__global__ void __launch_bounds__(256) col2im_bias_relu_kernel(const float* __restrict__ col, const float* __restrict__ bias,
                                                               float* __restrict__ im, ColArgs a) {
  const unsigned hw = (unsigned)a.H * a.W, hwc = (unsigned)a.Hc * a.Wc;
  const unsigned plane = blockIdx.y;                       // n * C + c
  const float* src = col + (size_t)plane * a.k * a.k * hwc;
  const float b = bias ? bias[plane % (unsigned)a.C] : 0.f;
  for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
    const int y = p / a.W + a.pad, x = p % a.W + a.pad;
    const int yc_lo = y < a.k ? 0 : (y - a.k) / a.stride + 1, yc_hi = min(y / a.stride + 1, a.Hc);
    const int xc_lo = x < a.k ? 0 : (x - a.k) / a.stride + 1, xc_hi = min(x / a.stride + 1, a.Wc);
    float v = 0.f;
    for (int yc = yc_lo; yc < yc_hi; ++yc) {
      const int i = y - yc * a.stride;
      for (int xc = xc_lo; xc < xc_hi; ++xc) {
        const int j = x - xc * a.stride;
        v += src[(size_t)(i * a.k + j) * hwc + (unsigned)yc * a.Wc + xc];
      }
    }
    v += b;
    im[((size_t)(plane / (unsigned)a.C) * a.im_ctot + a.im_c0 + plane % (unsigned)a.C) * hw + p] = (a.relu && v <= 0.f) ? v * a.slope : v;
  }
}

static int col_geometry(const char* who, int N, int C, int H, int W, int k, int pad, int stride, ColArgs* a) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || pad < 0 || stride <= 0)
    return fail(FN2_ERR_INVALID_ARG, "%s: bad arguments (N %d C %d H %d W %d kernel %d pad %d stride %d)", who, N, C, H, W, k, pad, stride);
  if (H + 2 * pad < k || W + 2 * pad < k) return fail(FN2_ERR_INVALID_ARG, "%s: kernel %d larger than the padded image", who, k);
  a->C = C; a->H = H; a->W = W; a->k = k; a->pad = pad; a->stride = stride;
  a->Hc = (H + 2 * pad - k) / stride + 1;
  a->Wc = (W + 2 * pad - k) / stride + 1;
  a->slope = 0.f; a->relu = 0; a->im_ctot = C; a->im_c0 = 0;
  if ((long long)N * C > 0x7fffffffll / 256 || (long long)H * W >= (1ll << 31) || (long long)k * k * a->Hc * a->Wc >= (1ll << 31))
    return fail(FN2_ERR_UNSUPPORTED, "%s: blob too large", who);
  return FN2_OK;
}

}  // namespace fn2

using namespace fn2;

FN2_API int fn2_im2col_forward(const float* im, float* col, int N, int C, int H, int W, int kernel, int pad, int stride, void* stream) {
  ColArgs a;
  int rc = col_geometry("im2col", N, C, H, W, kernel, pad, stride, &a);
  if (rc) return rc;
  if (N == 0) return FN2_OK;
  if (!im || !col) return fail(FN2_ERR_INVALID_ARG, "im2col: null blob");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned hwc = (unsigned)a.Hc * a.Wc;
  unsigned bx = (hwc + 255) / 256; if (bx > 64) bx = 64;
  const long long planes = (long long)N * C;
  for (long long p0 = 0; p0 < planes; p0 += 65535) {
    const unsigned py = (unsigned)((planes - p0) < 65535 ? (planes - p0) : 65535);
    const float* s = im + (size_t)p0 * H * W;
    float* d = col + (size_t)p0 * kernel * kernel * hwc;
    if (kernel == 3) hipLaunchKernelGGL(im2col_kernel<3>, dim3(bx, py), dim3(256), 0, st, s, d, a);
    else             hipLaunchKernelGGL(im2col_kernel<0>, dim3(bx, py), dim3(256), 0, st, s, d, a);
  }
  return check_launch("im2col_forward");
}

static int col2im_launch(const float* col, const float* bias, float* im, int N, int C, int H, int W, int kernel, int pad, int stride,
                         int apply_relu, float negative_slope, int im_ctot, int im_c0, void* stream) {
  ColArgs a;
  int rc = col_geometry("col2im_bias_relu", N, C, H, W, kernel, pad, stride, &a);
  if (rc) return rc;
  if (im_c0 < 0 || im_c0 + C > im_ctot) return fail(FN2_ERR_INVALID_ARG, "col2im_bias_relu: channel slice outside the blob");
  if (N == 0) return FN2_OK;
  if (!im || !col) return fail(FN2_ERR_INVALID_ARG, "col2im_bias_relu: null blob");
  a.relu = apply_relu ? 1 : 0; a.slope = negative_slope; a.im_ctot = im_ctot; a.im_c0 = im_c0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned hw = (unsigned)H * W, hwc = (unsigned)a.Hc * a.Wc;
  unsigned bx = (hw + 255) / 256; if (bx > 64) bx = 64;
  const long long planes = (long long)N * C;
  // gridDim.y <= 65535: fold the planes in chunks of whole samples (the kernel derives (n, c) from its plane index)
  const long long step = planes > 65535 ? ((65535 / C) > 0 ? (long long)(65535 / C) * C : 0) : planes;
  if (step == 0) return fail(FN2_ERR_UNSUPPORTED, "col2im_bias_relu: more than 65535 channels");
  for (long long p0 = 0; p0 < planes; p0 += step) {
    const unsigned py = (unsigned)((planes - p0) < step ? (planes - p0) : step);
    hipLaunchKernelGGL(col2im_bias_relu_kernel, dim3(bx, py), dim3(256), 0, st, col + (size_t)p0 * kernel * kernel * hwc, bias,
                       im + (size_t)(p0 / C) * im_ctot * hw, a);
  }
  return check_launch("col2im_bias_relu_forward");
}

FN2_API int fn2_col2im_bias_relu_forward(const float* col, const float* bias, float* im, int N, int C, int H, int W,
                                         int kernel, int pad, int stride, int apply_relu, float negative_slope, void* stream) {
  return col2im_launch(col, bias, im, N, C, H, W, kernel, pad, stride, apply_relu, negative_slope, C, 0, stream);
}

FN2_API int fn2_col2im_bias_relu_forward_into(const float* col, const float* bias, float* top, int N, int C, int H, int W,
                                              int kernel, int pad, int stride, int apply_relu, float negative_slope,
                                              int top_channels, int top_c0, void* stream) {
  return col2im_launch(col, bias, top, N, C, H, W, kernel, pad, stride, apply_relu, negative_slope, top_channels, top_c0, stream);
}
