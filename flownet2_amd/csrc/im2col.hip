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

// Deconvolution{4, 2, 1} (every deconv* layer of the FlowNet decoders; H, W even: Hc = H / 2, Wc = W / 2): a thread owns one COLUMN-GRID position
// (yc0, xc0) and writes the 2 x 2 image pixels (2 yc0 + {0, 1}, 2 xc0 + {0, 1}).  Image row 2 yc0 collects kernel rows i = 3 (from grid row
// yc0 - 1) and i = 1 (yc0), row 2 yc0 + 1 rows i = 2 (yc0) and i = 0 (yc0 + 1); columns likewise -- every one of the 16 (i, j) planes is read
// exactly once per thread at a FIXED offset from (yc0, xc0): 16 fully coalesced loads in flight and two 8-byte stores per thread instead of four
// short-lived threads with 4 loads each (round 6: 26.3 -> see profiles/r06_layer_microbench.md).  Same sums in the same order as the kernel
// above (grid rows ascending, then columns; a position outside the grid contributes + 0.0f, which leaves a running sum that started at + 0.0f
// unchanged bit for bit): bit-identical.
__global__ void __launch_bounds__(256) col2im_k4s2p1_bias_relu_kernel(const float* __restrict__ col, const float* __restrict__ bias,
                                                                      float* __restrict__ im, ColArgs a) {
  const unsigned hw = (unsigned)a.H * a.W, hwc = (unsigned)a.Hc * a.Wc;
  const unsigned plane = blockIdx.y;                       // n * C + c
  const float* src = col + (size_t)plane * 16 * hwc;
  const float b = bias ? bias[plane % (unsigned)a.C] : 0.f;
  float* dst = im + ((size_t)(plane / (unsigned)a.C) * a.im_ctot + a.im_c0 + plane % (unsigned)a.C) * hw;
  for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < hwc; p += gridDim.x * blockDim.x) {
    const int yc0 = p / a.Wc, xc0 = p - yc0 * a.Wc;
    // t[i][j]: plane (i, j) at grid position (yc0 + dy(i), xc0 + dx(j)),  d(3) = -1, d(1) = d(2) = 0, d(0) = +1
    float t[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yc = yc0 + (i == 3 ? -1 : i == 0 ? 1 : 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xc = xc0 + (j == 3 ? -1 : j == 0 ? 1 : 0);
        const bool ok = yc >= 0 && yc < a.Hc && xc >= 0 && xc < a.Wc;
        t[i][j] = ok ? src[(size_t)(i * 4 + j) * hwc + (unsigned)yc * a.Wc + xc] : 0.f;
      }
    }
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      const int i0 = py ? 2 : 3, i1 = py ? 0 : 1;          // kernel rows of the lower / higher grid row
      float o[2];
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int j0 = px ? 2 : 3, j1 = px ? 0 : 1;
        float v = 0.f;
        v += t[i0][j0]; v += t[i0][j1]; v += t[i1][j0]; v += t[i1][j1];
        v += b;
        o[px] = (a.relu && v <= 0.f) ? v * a.slope : v;
      }
      *reinterpret_cast<float2*>(dst + (size_t)(2 * yc0 + py) * a.W + 2 * xc0) = make_float2(o[0], o[1]);
    }
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
  const bool quad = kernel == 4 && stride == 2 && pad == 1 && H % 2 == 0 && W % 2 == 0 && (reinterpret_cast<uintptr_t>(im) & 7) == 0;
  unsigned bq = (hwc + 255) / 256; if (bq > 64) bq = 64;
  for (long long p0 = 0; p0 < planes; p0 += step) {
    const unsigned py = (unsigned)((planes - p0) < step ? (planes - p0) : step);
    if (quad) hipLaunchKernelGGL(col2im_k4s2p1_bias_relu_kernel, dim3(bq, py), dim3(256), 0, st, col + (size_t)p0 * kernel * kernel * hwc, bias,
                                 im + (size_t)(p0 / C) * im_ctot * hw, a);
    else hipLaunchKernelGGL(col2im_bias_relu_kernel, dim3(bx, py), dim3(256), 0, st, col + (size_t)p0 * kernel * kernel * hwc, bias,
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
