// Flow-head layers of the FlowNet decoders for gfx950: the `predict_flow` convolutions (3x3, stride 1, pad 1,
// num_output = 2) and the `upsample_flow` deconvolutions (4x4, stride 2, pad 1, 2 -> 2 channels).
//
// In the reference these are stock Convolution / Deconvolution layers (conv_layer.cpp:8-40, deconv_layer.cpp:8-45:
// per-sample im2col + SGEMM with M = 2 output rows).  They carry < 0.5 % of the network's flops but a dense-GEMM
// library path spends 30-160 us on each (MIOpen fp32, profiles/r01_rocprof_summary.md: 0.54 ms of a 4.5 ms FlowNetC
// forward) because a 2-row GEMM cannot fill matrix tiles.  They are HBM-bound reductions over input channels:
// read the input once, 18 FMAs per loaded value.
//
//   conv3x3_c2_taps : out[y,x] = sum_t P_t[y + dy - 1, x + dx - 1] with the per-tap channel reductions
//                P_t[y',x'] = sum_c w[o,c,t] * in[c,y',x'] taken AT THE INPUT PIXEL: every input value is loaded exactly once
//                (fully coalesced, 64 consecutive pixels per wave) and feeds 18 FMAs (9 taps x 2 outputs) with wave-uniform
//                weights, instead of being loaded by nine neighbouring threads (the texture path, not HBM, bounded that form:
//                43 us for the [8,194,80,112] head).  Block = 64 pixels x G channel groups (one wave each), partials reduced
//                through LDS in a fixed order, then written as an 18-plane image P (optionally one per channel split);
//   conv3x3_c2_gather : the 9-tap shift-sum of P (+ bias, + the splits in a fixed order): deterministic.
//   deconv4x4s2_c2 : one thread per output pixel; each output touches 2x2 input taps per input channel.
#include "fn2_common.hpp"

namespace fn2 {

constexpr int kHeadPix = 64;

// out[n, o, y, x] = bias[o] + sum_c sum_{dy,dx} w[o, c, dy, dx] * in[n, c, y + dy - 1, x + dx - 1]      (o < 2)
// Pass 1: P[split][n][o * 9 + t][y'][x'] = sum over this split's channels of w[o, c, t] * in[n, c, y', x'].
// grid: (pixel blocks of 64, channel splits); block = 64 pixels x G waves, wave g takes every G-th chunk of 8 channels.
template <int G>
__global__ void __launch_bounds__(kHeadPix* G) conv3x3_c2_taps(const float* __restrict__ in, const float* __restrict__ w,
                                                                float* __restrict__ P, int N, int C, int H, int W, int nsplit) {
  __shared__ float red[G][18][kHeadPix];
  const int lane_pix = threadIdx.x % kHeadPix;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / kHeadPix);      // one wave per channel group
  const size_t plane = (size_t)H * W;
  const long long total = (long long)N * H * W;
  const long long p = (long long)blockIdx.x * kHeadPix + lane_pix;
  const bool live = p < total;
  const int n = live ? (int)(p / (long long)plane) : 0;
  const unsigned pix = live ? (unsigned)(p - (long long)n * plane) : 0u;
  const float* src = in + (size_t)n * C * plane + pix;
  float a[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) a[k] = 0.f;
  const int ngroups = G * nsplit;
  const int c_per = (C + ngroups - 1) / ngroups;
  const int c_lo = min(C, ((int)blockIdx.y * G + grp) * c_per), c_hi = min(C, c_lo + c_per);
  for (int c = c_lo; c < c_hi; c += 8) {          // 8 channel loads in flight per thread
    float v[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) v[cc] = (c + cc < c_hi) ? src[(size_t)(c + cc) * plane] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      if (c + cc < c_hi) {                         // wave-uniform
        const float* w0 = w + (size_t)(c + cc) * 9;                 // w[0, c, :, :]  (scalar loads)
        const float* w1 = w + ((size_t)C + c + cc) * 9;             // w[1, c, :, :]
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          a[t] = fmaf(w0[t], v[cc], a[t]);
          a[9 + t] = fmaf(w1[t], v[cc], a[9 + t]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 18; ++k) red[grp][k][lane_pix] = a[k];
  __syncthreads();
  if (live) {
    float* dst = P + ((size_t)blockIdx.y * N + n) * 18 * plane + pix;
    for (int k = grp; k < 18; k += G) {            // the G waves share the 18 planes; fixed summation order
      float sacc = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) sacc += red[g][k][lane_pix];
      dst[(size_t)k * plane] = sacc;
    }
  }
}

// Pass 2: out[n, o, y, x] = bias[o] + sum_split sum_t P[split][n][o*9 + t][y + dy - 1][x + dx - 1]   (zero outside the image)
template <int NSPLIT>       // compile-time: all 9 * NSPLIT loads are in flight together (one memory round trip, not NSPLIT)
__global__ void __launch_bounds__(256) conv3x3_c2_gather(const float* __restrict__ P, const float* __restrict__ bias,
                                                          float* __restrict__ out, int N, int H, int W) {
  const unsigned plane = (unsigned)H * W;
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= plane) return;
  const int y = pix / W, x = pix - y * W;
  const unsigned no = blockIdx.y;                  // n * 2 + o
  const unsigned n = no >> 1, o = no & 1;
  float v[NSPLIT][9];
#pragma unroll
  for (int sp = 0; sp < NSPLIT; ++sp) {
    const float* Pn = P + (((size_t)sp * N + n) * 18 + o * 9) * plane;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = y + dy - 1, xx = x + dx - 1;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        v[sp][dy * 3 + dx] = ok ? Pn[(size_t)(dy * 3 + dx) * plane + (unsigned)yy * W + xx] : 0.f;
      }
  }
  float sacc = bias ? bias[o] : 0.f;
#pragma unroll
  for (int sp = 0; sp < NSPLIT; ++sp)
#pragma unroll
    for (int t = 0; t < 9; ++t) sacc += v[sp][t];
  out[(size_t)no * plane + pix] = sacc;
}

// Caffe Deconvolution 4x4, stride 2, pad 1 (weight [Cin=2, Cout=2, 4, 4], base_conv_layer.cpp:125-139):
// out[n, o, Y, X] = bias[o] + sum_c sum_{ky,kx : (Y + 1 - ky) even, (X + 1 - kx) even} w[c, o, ky, kx] * in[n, c, (Y+1-ky)/2, (X+1-kx)/2]
__global__ void __launch_bounds__(256) deconv4x4s2_c2(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int N, int H, int W, int out_ctot, int out_c0) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long long total = (long long)N * Ho * Wo;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(idx % Wo), Y = (int)((idx / Wo) % Ho), n = (int)(idx / ((long long)Wo * Ho));
    float a0 = bias ? bias[0] : 0.f, a1 = bias ? bias[1] : 0.f;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
      const int ky = ((Y + 1) & 1) + 2 * ty;            // kernel rows with (Y + 1 - ky) even
      const int iy = (Y + 1 - ky) / 2;
      if (Y + 1 - ky < 0 || iy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int kx = ((X + 1) & 1) + 2 * tx;
        const int ix = (X + 1 - kx) / 2;
        if (X + 1 - kx < 0 || ix >= W) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float v = in[(((size_t)n * 2 + c) * H + iy) * W + ix];
          a0 = fmaf(w[((c * 2 + 0) * 4 + ky) * 4 + kx], v, a0);
          a1 = fmaf(w[((c * 2 + 1) * 4 + ky) * 4 + kx], v, a1);
        }
      }
    }
    out[(((size_t)n * out_ctot + out_c0 + 0) * Ho + Y) * Wo + X] = a0;
    out[(((size_t)n * out_ctot + out_c0 + 1) * Ho + Y) * Wo + X] = a1;
  }
}

}  // namespace fn2

using namespace fn2;

constexpr int kHeadG = 16;     // waves (channel groups) per block: 64 pixels x 16 groups, 72 KB of LDS for the reduction

static int head_splits(int N, int C, int H, int W) {
  const long long blocks = ((long long)order_batch(N) * H * W + kHeadPix - 1) / kHeadPix;     // the channel split fixes the summation order
  // power of two in {1, 2, 4, 8}: ~4096 waves in flight, but at least 8 channels per wave
  int nsplit = 1;
  while (nsplit < 8 && blocks * kHeadG * nsplit < 4096 && C >= 8 * kHeadG * nsplit * 2) nsplit *= 2;
  return nsplit;
}

// Scratch: the 18-plane per-tap image P, one per channel split.
FN2_API size_t fn2_predict_flow_conv_workspace_bytes(int N, int C, int H, int W) {
  if (N < 1 || C < 1 || H < 1 || W < 1) return 0;
  return sizeof(float) * (size_t)head_splits(N, C, H, W) * N * 18 * H * W;
}

// Convolution{kernel 3, stride 1, pad 1, num_output 2}: in [N,C,H,W], weight [2,C,3,3], bias [2] or NULL, out [N,2,H,W].
FN2_API int fn2_predict_flow_conv_forward(const float* in, const float* weight, const float* bias, float* out,
                                          int N, int C, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "predict_flow_conv: bad shape [%d,%d,%d,%d]", N, C, H, W);
  if (N == 0) return FN2_OK;
  if (!in || !weight || !out) return fail(FN2_ERR_INVALID_ARG, "predict_flow_conv: NULL blob pointer");
  if ((long long)H * W >= (1ll << 31) || 2ll * N > 65535) return fail(FN2_ERR_UNSUPPORTED, "predict_flow_conv: blob too large");
  const size_t need = fn2_predict_flow_conv_workspace_bytes(N, C, H, W);
  if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "predict_flow_conv: workspace too small (%zu < %zu)", workspace_bytes, need);
  const long long total = (long long)N * H * W;
  const unsigned blocks = (unsigned)((total + kHeadPix - 1) / kHeadPix);
  const int nsplit = head_splits(N, C, H, W);
  hipStream_t st = as_stream(stream);
  float* P = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(conv3x3_c2_taps<kHeadG>, dim3(blocks, nsplit), dim3(kHeadPix * kHeadG), 0, st, in, weight, P, N, C, H, W, nsplit);
  const dim3 g2(((unsigned)H * W + 255) / 256, 2 * N);
  switch (nsplit) {
    case 1: hipLaunchKernelGGL(conv3x3_c2_gather<1>, g2, dim3(256), 0, st, P, bias, out, N, H, W); break;
    case 2: hipLaunchKernelGGL(conv3x3_c2_gather<2>, g2, dim3(256), 0, st, P, bias, out, N, H, W); break;
    case 4: hipLaunchKernelGGL(conv3x3_c2_gather<4>, g2, dim3(256), 0, st, P, bias, out, N, H, W); break;
    default: hipLaunchKernelGGL(conv3x3_c2_gather<8>, g2, dim3(256), 0, st, P, bias, out, N, H, W); break;
  }
  return check_launch("predict_flow_conv_forward");
}

// Deconvolution{kernel 4, stride 2, pad 1, num_output 2} on a 2-channel flow: in [N,2,H,W], weight [2,2,4,4], out [N,2,2H,2W].
FN2_API int fn2_upsample_flow_deconv_forward(const float* in, const float* weight, const float* bias, float* out,
                                             int N, int H, int W, void* stream) {
  if (N < 0 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv: bad shape");
  if (N == 0) return FN2_OK;
  if (!in || !weight || !out) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv: NULL blob pointer");
  const long long total = (long long)N * 4 * H * W;
  hipLaunchKernelGGL(deconv4x4s2_c2, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, as_stream(stream), in, weight, bias, out, N, H, W, 2, 0);
  return check_launch("upsample_flow_deconv_forward");
}

FN2_API int fn2_upsample_flow_deconv_forward_into(const float* in, const float* weight, const float* bias, float* top,
                                                  int N, int H, int W, int top_channels, int top_c0, void* stream) {
  if (N < 0 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv: bad shape");
  if (top_c0 < 0 || top_c0 + 2 > top_channels) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv: channel slice outside the blob");
  if (N == 0) return FN2_OK;
  if (!in || !weight || !top) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv: NULL blob pointer");
  const long long total = (long long)N * 4 * H * W;
  hipLaunchKernelGGL(deconv4x4s2_c2, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, as_stream(stream), in, weight, bias, top, N, H, W, top_channels, top_c0);
  return check_launch("upsample_flow_deconv_forward_into");
}
