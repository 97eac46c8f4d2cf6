// Resample (NEAREST / LINEAR / CUBIC) for gfx950, forward only.
//
// Replaces ResampleLayer::Forward_gpu (reference: src/caffe/layers/resample_layer.cu:128-206).
// Arithmetic follows InterpolationKernel (:39-95) tap for tap -- including the swapped half-pixel
// offsets (x uses fy/2, y uses fx/2, :62-63) and the sum/wsum edge renormalisation (:93) -- but
// the row coefficient is hoisted out of the column loop and taps outside the image are skipped by
// clamping the loop bounds instead of testing every tap.
#include "fn2_common.hpp"

#include <cmath>

namespace fn2 {

__device__ __forceinline__ float bicubic_coeff(float x_) {   // :14-20
  const float x = fabsf(x_);
  if (x <= 1.0f) return x * x * (1.5f * x - 2.5f) + 1.0f;
  else if (x < 2.0f) return x * (x * (-0.5f * x + 2.5f) - 4.0f) + 2.0f;
  else return 0.0f;
}
__device__ __forceinline__ float triangle_coeff(float x) {   // :28-33
  if (-1 <= x && x < 0) return x + 1;
  if (0 <= x && x <= 1) return 1 - x;
  return 0;
}

struct ResampleArgs {
  int NC, Hin, Win, Hout, Wout;
  float fx, fy, ax, ay;
  int rx, ry;
};

__global__ void __launch_bounds__(256) resample_nearest(const float* __restrict__ in, float* __restrict__ out, ResampleArgs a) {
  const long long total = (long long)a.NC * a.Hout * a.Wout;
  const int out_cs = a.Hout * a.Wout;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx / out_cs);
    const int x_out = (int)(idx % out_cs) % a.Wout, y_out = (int)(idx % out_cs) / a.Wout;
    const float x_in = x_out * a.fx + a.fy / 2.0f - 0.5f;   // :117
    const float y_in = y_out * a.fy + a.fx / 2.0f - 0.5f;   // :118
    int xr = (int)roundf(x_in), yr = (int)roundf(y_in);
    // The reference reads in_ptr[yr*W+xr] unclamped (:123); clamp instead of faulting.
    xr = min(max(xr, 0), a.Win - 1);
    yr = min(max(yr, 0), a.Hin - 1);
    out[idx] = in[(size_t)c * a.Hin * a.Win + (size_t)yr * a.Win + xr];
  }
}

template <bool CUBIC>
__global__ void __launch_bounds__(256) resample_interp(const float* __restrict__ in, float* __restrict__ out, ResampleArgs a) {
  const long long total = (long long)a.NC * a.Hout * a.Wout;
  const int out_cs = a.Hout * a.Wout;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx / out_cs);
    const int x_out = (int)(idx % out_cs) % a.Wout, y_out = (int)(idx % out_cs) / a.Wout;
    const float x_in = x_out * a.fx + a.fy / 2.0f - 0.5f;   // :62
    const float y_in = y_out * a.fy + a.fx / 2.0f - 0.5f;   // :63
    const int xr = (int)roundf(x_in), yr = (int)roundf(y_in);
    const float* src = in + (size_t)c * a.Hin * a.Win;
    float sum = 0.f, wsum = 0.f;
    const int y0 = max(yr - a.ry, 0), y1 = min(yr + a.ry, a.Hin - 1);
    const int x0 = max(xr - a.rx, 0), x1 = min(xr + a.rx, a.Win - 1);
    for (int y = y0; y <= y1; ++y) {
      const float ky = CUBIC ? bicubic_coeff(a.ay * (y_in - y)) : triangle_coeff(a.ay * (y_in - y));
      for (int x = x0; x <= x1; ++x) {
        const float dx = x_in - x;
        // :87/:89 -- the reference evaluates ((ax*k(ax*dx))*ay)*k(ay*dy); same association here,
        // with k(ay*dy) hoisted out of the x loop.
        const float w = a.ax * (CUBIC ? bicubic_coeff(a.ax * dx) : triangle_coeff(a.ax * dx)) * a.ay * ky;
        sum = fmaf(w, src[(size_t)y * a.Win + x], sum);
        wsum += w;
      }
    }
    out[idx] = (!wsum) ? 0.f : (sum / wsum);   // :93
  }
}

}  // namespace fn2

using namespace fn2;

FN2_API int fn2_resample_forward(const float* in, float* out, int N, int C, int Hin, int Win, int Hout, int Wout,
                                 int type, int antialias_param, void* stream) {
  if (N < 0 || C < 1 || Hin < 1 || Win < 1) return fail(FN2_ERR_INVALID_ARG, "resample: bad bottom shape");
  if (Hout < 1 || Wout < 1) return fail(FN2_ERR_INVALID_ARG, "ResampleLayer must have top_height > 0 and top_width > 0");
  if (type != FN2_RESAMPLE_NEAREST && type != FN2_RESAMPLE_LINEAR && type != FN2_RESAMPLE_CUBIC)
    return fail(FN2_ERR_UNSUPPORTED, "ResampleLayer: only CUBIC, LINEAR and NEAREST interpolation is supported for now");
  if (N == 0) return FN2_OK;
  if (!in || !out) return fail(FN2_ERR_INVALID_ARG, "resample: NULL blob pointer");
  ResampleArgs a;
  a.NC = N * C; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout;
  a.fx = (float)Win / (float)Wout;                              // :146
  a.fy = (float)Hin / (float)Hout;                              // :147
  const bool is_down = (a.fx > 1) || (a.fy > 1);                // :179
  const bool antialias = is_down && antialias_param;            // :180
  const int kernel_width = (type == FN2_RESAMPLE_CUBIC) ? 4 : 2;   // :182-185
  a.ax = 1.0f / (antialias ? a.fx : 1.0f);                      // :71
  a.ay = 1.0f / (antialias ? a.fy : 1.0f);                      // :72
  a.rx = (a.fx < 1.0f) ? 2 : (int)std::ceil((float)kernel_width / a.ax);   // :73
  a.ry = (a.fy < 1.0f) ? 2 : (int)std::ceil((float)kernel_width / a.ay);   // :74
  const long long total = (long long)a.NC * Hout * Wout;
  const unsigned blocks = blocks_for(total, 256);
  hipStream_t st = as_stream(stream);
  if (type == FN2_RESAMPLE_NEAREST) hipLaunchKernelGGL(resample_nearest, dim3(blocks), dim3(256), 0, st, in, out, a);
  else if (type == FN2_RESAMPLE_CUBIC) hipLaunchKernelGGL(resample_interp<true>, dim3(blocks), dim3(256), 0, st, in, out, a);
  else hipLaunchKernelGGL(resample_interp<false>, dim3(blocks), dim3(256), 0, st, in, out, a);
  return check_launch("resample_forward");
}
