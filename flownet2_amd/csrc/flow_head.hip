// Flow-head layers of the FlowNet decoders for gfx950: the `predict_flow` convolutions (3x3, stride 1, pad 1,
// num_output = 2) and the `upsample_flow` deconvolutions (4x4, stride 2, pad 1, 2 -> 2 channels).
//
// In the reference these are stock Convolution / Deconvolution layers (conv_layer.cpp:8-40, deconv_layer.cpp:8-45:
// per-sample im2col + SGEMM with M = 2 output rows).  They carry < 0.5 % of the network's flops but a dense-GEMM
// library path spends 30-160 us on each (MIOpen fp32, profiles/r01_rocprof_summary.md: 0.54 ms of a 4.5 ms FlowNetC
// forward) because a 2-row GEMM cannot fill matrix tiles.  They are HBM-bound reductions over input channels:
// read the input once, 18 FMAs per loaded value.
//
//   conv3x3_c2 : block = 64 output pixels x G channel groups (one wave each, G = 4 or 16); each thread walks its share
//                of the input channels, 9 taps x 2 outputs in registers; weights are wave-uniform (scalar loads);
//                the G partial sums are reduced through LDS in a fixed order (deterministic).
//   deconv4x4s2_c2 : one thread per output pixel; each output touches 2x2 input taps per input channel.
#include "fn2_common.hpp"

namespace fn2 {

constexpr int kHeadPix = 64;

// out[n, o, y, x] = bias[o] + sum_c sum_{dy,dx} w[o, c, dy, dx] * in[n, c, y + dy - 1, x + dx - 1]      (o < 2)
// One wave = 64 consecutive output pixels x one group of input channels; G waves per block split the channels
// (G = 16 for the small maps of the coarse scales, where there are few pixels but ~1000 channels, G = 4 for the
// fine ones).  The channel loop is unrolled 4x so that 36 tap loads are in flight per thread.
template <int G>
__global__ void __launch_bounds__(kHeadPix* G) conv3x3_c2(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int N, int C, int H, int W, int nsplit) {
  // blockIdx.y = channel split: with nsplit > 1 the block writes its partial sum (no bias) to out + split * N*2*H*W
  // (a workspace) and head_reduce adds the splits and the bias in a fixed order.
  __shared__ float red[G][2][kHeadPix];
  const int lane_pix = threadIdx.x % kHeadPix;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / kHeadPix);      // one wave per channel group
  const long long total = (long long)N * H * W;
  const long long p = (long long)blockIdx.x * kHeadPix + lane_pix;
  const bool live = p < total;
  const int x = live ? (int)(p % W) : 0;
  const int y = live ? (int)((p / W) % H) : 0;
  const int n = live ? (int)(p / ((long long)W * H)) : 0;
  const size_t plane = (size_t)H * W;
  // tap validity and offsets (zero padding): invalid taps read offset 0 of the plane and are multiplied by 0
  int off[9];
  float msk[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = y + dy - 1, xx = x + dx - 1;
      const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
      off[dy * 3 + dx] = ok ? yy * W + xx : 0;
      msk[dy * 3 + dx] = ok ? 1.f : 0.f;
    }
  const float* src = in + (size_t)n * C * plane;
  float a0 = 0.f, a1 = 0.f;
  const int ngroups = G * nsplit;
  const int c_per = (C + ngroups - 1) / ngroups;
  const int c_lo = min(C, ((int)blockIdx.y * G + grp) * c_per), c_hi = min(C, c_lo + c_per);
  // 4 channels per step: all 36 tap loads are issued before the first FMA (left to itself hipcc serialises
  // load -> s_waitcnt vmcnt(0) -> fma per tap and the kernel runs at 1/10 of the memory rate)
  for (int c = c_lo; c < c_hi; c += 4) {
    float v[4][9];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int ci = min(c + cc, c_hi - 1);              // tail: re-read the last channel, weight it by 0
      const float* pc = src + (size_t)ci * plane;
#pragma unroll
      for (int t = 0; t < 9; ++t) v[cc][t] = pc[off[t]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int ci = min(c + cc, c_hi - 1);
      const float tail = (c + cc < c_hi) ? 1.f : 0.f;
      const float* w0 = w + (size_t)ci * 9;              // w[0, c, :, :]  (wave-uniform -> scalar loads)
      const float* w1 = w + ((size_t)C + ci) * 9;        // w[1, c, :, :]
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float vv = v[cc][t] * (msk[t] * tail);
        a0 = fmaf(w0[t], vv, a0);
        a1 = fmaf(w1[t], vv, a1);
      }
    }
  }
  red[grp][0][lane_pix] = a0;
  red[grp][1][lane_pix] = a1;
  __syncthreads();
  if (threadIdx.x < 2 * kHeadPix) {
    const int o = threadIdx.x / kHeadPix;
    if (live) {
      float s = (bias && nsplit == 1) ? bias[o] : 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) s += red[g][o][lane_pix];
      out[(size_t)blockIdx.y * ((size_t)N * 2 * plane) + ((size_t)n * 2 + o) * plane + (size_t)y * W + x] = s;
    }
  }
}

__global__ void __launch_bounds__(256) head_reduce(const float* __restrict__ partial, const float* __restrict__ bias,
                                                    float* __restrict__ out, int nsplit, long long per_split, int plane) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per_split; i += (long long)gridDim.x * blockDim.x) {
    float s = bias ? bias[(i / plane) & 1] : 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(size_t)k * per_split + i];
    out[i] = s;
  }
}

// Caffe Deconvolution 4x4, stride 2, pad 1 (weight [Cin=2, Cout=2, 4, 4], base_conv_layer.cpp:125-139):
// out[n, o, Y, X] = bias[o] + sum_c sum_{ky,kx : (Y + 1 - ky) even, (X + 1 - kx) even} w[c, o, ky, kx] * in[n, c, (Y+1-ky)/2, (X+1-kx)/2]
__global__ void __launch_bounds__(256) deconv4x4s2_c2(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int N, int H, int W) {
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
    out[(((size_t)n * 2 + 0) * Ho + Y) * Wo + X] = a0;
    out[(((size_t)n * 2 + 1) * Ho + Y) * Wo + X] = a1;
  }
}

}  // namespace fn2

using namespace fn2;

static int head_splits(int N, int C, int H, int W) {
  const long long blocks = ((long long)N * H * W + kHeadPix - 1) / kHeadPix;
  int nsplit = (int)((2048 + blocks * 4 - 1) / (blocks * 4));       // aim at >= 2048 waves in flight
  nsplit = nsplit < 1 ? 1 : nsplit;
  const int max_by_channels = (C + 15) / 16;                          // keep >= 4 channels per wave
  if (nsplit > max_by_channels) nsplit = max_by_channels < 1 ? 1 : max_by_channels;
  if (nsplit > 64) nsplit = 64;
  return nsplit;
}

// Scratch for the channel-split partial sums of small maps (0 when no split is used).
FN2_API size_t fn2_predict_flow_conv_workspace_bytes(int N, int C, int H, int W) {
  if (N < 1 || C < 1 || H < 1 || W < 1) return 0;
  const int ns = head_splits(N, C, H, W);
  return ns > 1 ? sizeof(float) * (size_t)ns * N * 2 * H * W : 0;
}

// Convolution{kernel 3, stride 1, pad 1, num_output 2}: in [N,C,H,W], weight [2,C,3,3], bias [2] or NULL, out [N,2,H,W].
FN2_API int fn2_predict_flow_conv_forward(const float* in, const float* weight, const float* bias, float* out,
                                          int N, int C, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "predict_flow_conv: bad shape [%d,%d,%d,%d]", N, C, H, W);
  if (N == 0) return FN2_OK;
  if (!in || !weight || !out) return fail(FN2_ERR_INVALID_ARG, "predict_flow_conv: NULL blob pointer");
  const long long total = (long long)N * H * W;
  const unsigned blocks = (unsigned)((total + kHeadPix - 1) / kHeadPix);
  const int nsplit = head_splits(N, C, H, W);
  hipStream_t st = as_stream(stream);
  if (nsplit == 1) {
    hipLaunchKernelGGL(conv3x3_c2<4>, dim3(blocks, 1), dim3(kHeadPix * 4), 0, st, in, weight, bias, out, N, C, H, W, 1);
    return check_launch("predict_flow_conv_forward");
  }
  const size_t need = fn2_predict_flow_conv_workspace_bytes(N, C, H, W);
  if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "predict_flow_conv: workspace too small (%zu < %zu)", workspace_bytes, need);
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(conv3x3_c2<4>, dim3(blocks, nsplit), dim3(kHeadPix * 4), 0, st, in, weight, nullptr, partial, N, C, H, W, nsplit);
  const long long per_split = (long long)N * 2 * H * W;
  hipLaunchKernelGGL(head_reduce, dim3(blocks_for(per_split, 256, 1024)), dim3(256), 0, st, partial, bias, out, nsplit, per_split, H * W);
  return check_launch("predict_flow_conv_forward");
}

// Deconvolution{kernel 4, stride 2, pad 1, num_output 2} on a 2-channel flow: in [N,2,H,W], weight [2,2,4,4], out [N,2,2H,2W].
FN2_API int fn2_upsample_flow_deconv_forward(const float* in, const float* weight, const float* bias, float* out,
                                             int N, int H, int W, void* stream) {
  if (N < 0 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv: bad shape");
  if (N == 0) return FN2_OK;
  if (!in || !weight || !out) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv: NULL blob pointer");
  const long long total = (long long)N * 4 * H * W;
  hipLaunchKernelGGL(deconv4x4s2_c2, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, as_stream(stream), in, weight, bias, out, N, H, W);
  return check_launch("upsample_flow_deconv_forward");
}
