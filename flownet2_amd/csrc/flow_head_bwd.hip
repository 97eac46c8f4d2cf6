// Backward of the two 2-channel flow heads of the FlowNet decoders (round 3):
//   predict_flow*  = Convolution{3, 1, 1} C -> 2     <- ConvolutionLayer::Backward_gpu, src/caffe/layers/conv_layer.cu:26-60
//   upsample_flow* = Deconvolution{4, 2, 1} 2 -> 2    <- DeconvolutionLayer::Backward_gpu, src/caffe/layers/deconv_layer.cu:27-58
// (weight_gpu_gemm / backward_gpu_gemm / backward_gpu_bias, base_conv_layer.cpp:352-393).  A 2-channel side cannot fill a 16-wide MFMA
// tile and the layers are a few hundred MFLOP: the library spent ~30 us per call plus NCHW<->NHWC transposes of the wide blob on each of
// the nine heads of a FlowNetC training step.  Here they are streaming VALU kernels over NCHW:
//   * pf_wgrad:  workgroup = (16 bottom channels, sample, band of 32 rows); the two top_diff planes of the band (+ halo) sit in LDS; a wave takes
//                4 channels: its lanes walk the band's bottom pixels (coalesced rows) and feed 18 accumulators; one butterfly per (channel, part).
//                pf_finalize adds the parts of a channel in part order (lanes stride over the parts, butterfly): deterministic.
//                The bias gradient (sum of top_diff) rides along in the workgroups of channel group 0.
//   * pf_dgrad:  thread = pixel: the 2 x 3 x 3 neighbourhood of top_diff in registers, then 18 fmas + one coalesced store per channel.
//   * uf_*:      the same for the 4x4 / stride-2 transposed convolution on 2 -> 2 channels (64 weight gradients, 2 bottom gradients).
// Summation orders are fixed functions of the geometry; the oracle twins accumulate in double and are compared at 1e-5 * scale.
#include "fn2_common.hpp"

namespace fn2 {
namespace fhb {

constexpr int kCG = 16;         // bottom channels per workgroup (4 per wave)
constexpr int kRBMax = 32;      // rows per band: 32, 16 or 8 (the largest that still gives the launch ~3 workgroups per CU)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// K wave sums at once (round 6): a reduce-scatter butterfly.  At the step with lane mask m every lane hands HALF of the values it still
// holds to its partner and keeps the sums of the other half: K / 2 + K / 4 + ... + 1 = K - 1 shuffles and then log2(64 / K) plain butterfly
// steps on the one value that is left -- K - 1 + log2(64 / K) shuffles instead of the 6 K of K separate butterflies (64 sums: 63 instead
// of 384; the kernels below are chains of cross-lane operations and nothing else, 70-110 us each beside a matrix kernel).
// Afterwards v[0] of lane l is the wave-wide sum of the caller's v[l / (64 / K)] (every lane of a group of 64 / K holds it).
// Order: a fixed tree over the lanes (another one than K separate butterflies, hence other low bits), the same on every run.
template <int K, int CNT, int M>
struct ReduceScatterStep {          // (a template recursion, not a loop: every index and lane mask is a constant, the values stay in registers)
  static __device__ __forceinline__ void run(float (&v)[K], bool (&hi)[6]) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      const float keep = hi[M] ? v[i + CNT] : v[i];
      const float send = hi[M] ? v[i] : v[i + CNT];
      v[i] = keep + __shfl_xor(send, 1 << M, 64);
    }
    ReduceScatterStep<K, CNT / 2, M - 1>::run(v, hi);
  }
};
template <int K, int M>
struct ReduceScatterStep<K, 0, M> {
  static __device__ __forceinline__ void run(float (&v)[K], bool (&)[6]) {
#pragma unroll
    for (int b = M; b >= 0; --b) v[0] += __shfl_xor(v[0], 1 << b, 64);
  }
};
template <int K>
struct ReduceScatterStep<K, 0, -1> {
  static __device__ __forceinline__ void run(float (&)[K], bool (&)[6]) {}
};

template <int K>
__device__ __forceinline__ void wave_reduce_scatter(float (&v)[K], int lane) {
  static_assert(K >= 2 && K <= 64 && (K & (K - 1)) == 0, "a power of two of values");
  bool hi[6];
#pragma unroll
  for (int b = 0; b < 6; ++b) hi[b] = ((lane >> b) & 1) != 0;
  ReduceScatterStep<K, K / 2, 5>::run(v, hi);
}

// partial[(c * parts + part) * 18 + co * 9 + ky * 3 + kx];  bpart[part * 2 + co]
// Workgroup = (16 bottom channels, sample, band of up to 32 rows): the band of the two top_diff planes (+ halo) is staged once; every WAVE
// then takes 4 of the channels on its own -- lanes stride over the band's pixels, 18 accumulators, one butterfly per (channel, part) -- so
// there is no workgroup barrier per channel (the first version reduced 8-row bands through LDS per channel and spent its time there:
// 53-120 us per head against 10-20 now).
__global__ void __launch_bounds__(256) pf_wgrad(const float* __restrict__ bottom, int bctot, int bc0, const float* __restrict__ top_diff,
                                                float* __restrict__ partial, float* __restrict__ bpart, int C, int H, int W, int bands, int rb) {
  extern __shared__ float g[];                       // [2][rb + 2][W + 2], zero halo
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = blockIdx.y, n = part / bands, band = part - n * bands, parts = gridDim.y;
  const int y0 = band * rb, rows = min(rb, H - y0);
  const int Wp = W + 2, plane = (rb + 2) * Wp;
  // staging, eight elements per thread at a time: all eight loads are issued (addresses clamped into the blob) before the first value is
  // used -- a conditional load in front of its LDS store is a memory round trip of its own, and a band is 30 of them per thread; the index
  // decode uses reciprocal multiplies (two integer divisions per element otherwise)
  {
    const unsigned inv_plane = 0xffffffffu / (unsigned)plane + 1u, inv_wp = 0xffffffffu / (unsigned)Wp + 1u;      // exact for indices < 2^32 / divisor
    const float* td = top_diff + (size_t)(n * 2) * H * W;
    for (int i0 = tid; i0 < 2 * plane; i0 += 256 * 8) {
      float v[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const unsigned i = (unsigned)min(i0 + 256 * u, 2 * plane - 1);
        const unsigned co = __umulhi(i, inv_plane), r = i - co * (unsigned)plane, yy = __umulhi(r, inv_wp), xx = r - yy * (unsigned)Wp;
        const int y = y0 + (int)yy - 1, x = (int)xx - 1;
        ok[u] = y >= 0 && y < H && x >= 0 && x < W;
        v[u] = td[((size_t)co * H + min(max(y, 0), H - 1)) * W + min(max(x, 0), W - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 256 * u < 2 * plane) g[i0 + 256 * u] = ok[u] ? v[u] : 0.f;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0) {                             // bias gradient of this part: the interior of the band
    float b0 = 0.f, b1 = 0.f;
    for (int i = tid; i < rows * W; i += 256) {
      const int yy = i / W, xx = i - yy * W;
      b0 += g[(yy + 1) * Wp + xx + 1];
      b1 += g[plane + (yy + 1) * Wp + xx + 1];
    }
    b0 = wave_sum(b0); b1 = wave_sum(b1);
    if (lane == 0) { red[0][wave] = b0; red[1][wave] = b1; }
    __syncthreads();
    if (tid < 2) bpart[part * 2 + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
  }
  const int npix = rows * W;
  const int step_rows = 64 / W, step_cols = 64 - step_rows * W;
  for (int cc = 0; cc < kCG / 4; ++cc) {
    const int c = blockIdx.x * kCG + wave * (kCG / 4) + cc;
    if (c >= C) break;                               // wave-uniform; no barrier below
    const float* src = bottom + ((size_t)n * bctot + bc0 + c) * H * W + (size_t)y0 * W;
    float acc[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) acc[j] = 0.f;
    // (the pixel's row / column advance by 64 / W rows and 64 % W columns per step: the integer division per pixel and channel was a third
    // of this kernel's 32 M vector instructions per training step)
    int yy = lane / W, xx = lane - yy * W;
    for (int i = lane; i < npix; i += 64) {
      const float v = src[i];
      // dw[co][c][ky][kx] += bottom[c][y + ky - 1][x + kx - 1] * top_diff[co][y][x]; with p = (y + ky - 1, x + kx - 1) the bottom pixel:
      // top_diff at p - (ky - 1, kx - 1)  ->  LDS row (yy + 1) - (ky - 1), column (xx + 1) - (kx - 1)
      const float* gp = g + (yy + 2) * Wp + xx + 2;
      yy += step_rows; xx += step_cols;
      if (xx >= W) { xx -= W; ++yy; }
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            acc[co * 9 + ky * 3 + kx] = fmaf(v, gp[co * plane - ky * Wp - kx], acc[co * 9 + ky * 3 + kx]);
    }
    // lane j < 18 ends up holding sum j: the first 16 through one reduce-scatter (lane l then holds sum l / 4), the last two on their own
    float first[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) first[j] = acc[j];
    wave_reduce_scatter<16>(first, lane);
    float mine = __shfl(first[0], (lane & 15) * 4, 64);
    const float s16 = wave_sum(acc[16]), s17 = wave_sum(acc[17]);
    if (lane == 16) mine = s16;
    if (lane == 17) mine = s17;
    if (lane < 18) partial[((size_t)c * parts + part) * 18 + lane] = mine;
  }
}

// one wave per bottom channel: dw[co][c][tap] = sum over parts (lane l takes parts l, l + 64, ... in order; butterfly); block 0 also the bias
__global__ void __launch_bounds__(64) pf_finalize(const float* __restrict__ partial, const float* __restrict__ bpart, float* __restrict__ wdiff,
                                                  float* __restrict__ bdiff, int C, int parts, int accumulate) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float acc[18];
#pragma unroll
  for (int j = 0; j < 18; ++j) acc[j] = 0.f;
  for (int p = lane; p < parts; p += 64) {
    const float* q = partial + ((size_t)c * parts + p) * 18;
#pragma unroll
    for (int j = 0; j < 18; ++j) acc[j] += q[j];
  }
  {
    float first[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) first[j] = acc[j];
    wave_reduce_scatter<16>(first, lane);            // lane l: sum l / 4
    const float s16 = wave_sum(acc[16]), s17 = wave_sum(acc[17]);
    const int j = (lane & 3) == 0 ? lane >> 2 : lane == 1 ? 16 : lane == 2 ? 17 : -1;
    const float s = lane == 1 ? s16 : lane == 2 ? s17 : first[0];
    if (j >= 0 && wdiff) {
      float* d = wdiff + ((size_t)(j / 9) * C + c) * 9 + (j % 9);
      *d = accumulate ? *d + s : s;
    }
  }
  if (c == 0 && bdiff) {
    float b0 = 0.f, b1 = 0.f;
    for (int p = lane; p < parts; p += 64) { b0 += bpart[2 * p]; b1 += bpart[2 * p + 1]; }
    b0 = wave_sum(b0); b1 = wave_sum(b1);
    if (lane == 0) { bdiff[0] = accumulate ? bdiff[0] + b0 : b0; bdiff[1] = accumulate ? bdiff[1] + b1 : b1; }
  }
}

// bottom_diff[n][c][y][x] = sum_{co,ky,kx} top_diff[n][co][y - ky + 1][x - kx + 1] * w[co][c][ky][kx]
__global__ void __launch_bounds__(256) pf_dgrad(const float* __restrict__ top_diff, const float* __restrict__ weight, float* __restrict__ bottom_diff,
                                                int N, int C, int H, int W, int cpb) {
  const unsigned hw = (unsigned)H * W, pix = blockIdx.x * 256u + threadIdx.x;
  const int n = blockIdx.z;
  if (pix >= hw) return;
  const int y = pix / W, x = pix - y * W;
  float gv[18];
#pragma unroll
  for (int co = 0; co < 2; ++co)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y - ky + 1, xx = x - kx + 1;
        gv[co * 9 + ky * 3 + kx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? top_diff[((size_t)(n * 2 + co) * H + yy) * W + xx] : 0.f;
      }
  const int c0 = blockIdx.y * cpb, c1 = min(C, c0 + cpb);
  for (int c = c0; c < c1; ++c) {                    // the 18 weights of a channel are wave-uniform: scalar loads
    float s = 0.f;
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) s = fmaf(gv[co * 9 + t], weight[((size_t)co * C + c) * 9 + t], s);
    bottom_diff[((size_t)n * C + c) * hw + pix] = s;
  }
}

// ---- upsample_flow: top[n][co][Y][X] = sum_{ci,ky,kx: Y = 2y - 1 + ky, X = 2x - 1 + kx} bottom[n][ci][y][x] * w[ci][co][ky][kx] ----
// thread = bottom pixel: its 2 x 4 x 4 top_diff neighbourhood; 64 weight-gradient accumulators, 2 bias, 2 bottom gradients.
// partial[part * 66 + (ci * 2 + co) * 16 + ky * 4 + kx], [part * 66 + 64 + co] = bias part (top_diff summed over the block's OWN top pixels)
__global__ void __launch_bounds__(256) uf_backward(const float* __restrict__ bottom, const float* __restrict__ weight, const float* __restrict__ top_diff,
                                                   float* __restrict__ bottom_diff, float* __restrict__ partial, int H, int W, int want_w) {
  __shared__ float red[66][4];
  __shared__ float wl[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y, part = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned hw = (unsigned)H * W, pix = blockIdx.x * 256u + tid;
  const int Ho = 2 * H, Wo = 2 * W;
  if (tid < 64) wl[tid] = weight[tid];
  __syncthreads();
  float acc[64], bacc[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 64; ++j) acc[j] = 0.f;
  if (pix < hw) {
    const int y = pix / W, x = pix - y * W;
    float gv[2][16];
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const int Y = 2 * y - 1 + ky, X = 2 * x - 1 + kx;
          gv[co][ky * 4 + kx] = (Y >= 0 && Y < Ho && X >= 0 && X < Wo) ? top_diff[((size_t)(n * 2 + co) * Ho + Y) * Wo + X] : 0.f;
        }
    const float b0 = bottom[(size_t)(n * 2) * hw + pix], b1 = bottom[(size_t)(n * 2 + 1) * hw + pix];
    if (bottom_diff) {
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          d0 = fmaf(gv[co][t], wl[(0 * 2 + co) * 16 + t], d0);
          d1 = fmaf(gv[co][t], wl[(1 * 2 + co) * 16 + t], d1);
        }
      bottom_diff[(size_t)(n * 2) * hw + pix] = d0;
      bottom_diff[(size_t)(n * 2 + 1) * hw + pix] = d1;
    }
    if (want_w) {
#pragma unroll
      for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          acc[(0 * 2 + co) * 16 + t] = b0 * gv[co][t];
          acc[(1 * 2 + co) * 16 + t] = b1 * gv[co][t];
        }
      // bias: every top pixel belongs to exactly one bottom pixel's 2 x 2 block (Y in {2y, 2y + 1} <-> ky in {1, 2})
#pragma unroll
      for (int co = 0; co < 2; ++co) bacc[co] = (gv[co][1 * 4 + 1] + gv[co][1 * 4 + 2]) + (gv[co][2 * 4 + 1] + gv[co][2 * 4 + 2]);
    }
  }
  if (!want_w) return;
  {
    const float b0 = wave_sum(bacc[0]), b1 = wave_sum(bacc[1]);
    wave_reduce_scatter<64>(acc, lane);              // lane l: the wave's sum of weight-gradient term l
    red[lane][wave] = acc[0];
    if (lane == 0) { red[64][wave] = b0; red[65][wave] = b1; }
  }
  __syncthreads();
  if (tid < 66) partial[(size_t)part * 66 + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
}

__global__ void __launch_bounds__(64) uf_finalize(const float* __restrict__ partial, float* __restrict__ wdiff, float* __restrict__ bdiff, int parts,
                                                  int accumulate) {
  const int j = blockIdx.x, lane = threadIdx.x;      // one wave per output (64 weight gradients + 2 bias gradients)
  float s = 0.f;
  for (int p = lane; p < parts; p += 64) s += partial[(size_t)p * 66 + j];
  s = wave_sum(s);
  if (lane != 0) return;
  float* d = j < 64 ? (wdiff ? wdiff + j : nullptr) : (bdiff ? bdiff + (j - 64) : nullptr);
  if (d) *d = accumulate ? *d + s : s;
}

}  // namespace fhb
}  // namespace fn2

using namespace fn2;

// rows per band: a function of the geometry only (the summation order depends on it).  Only heights whose [2][rb + 2][W + 2] band fits
// the 60 KB of LDS a workgroup may take are candidates (wide maps get shorter bands: 1024x512 at 1/4 resolution, W = 256, C = 194 would
// prefer rb = 32 = 70 KB and takes 16); 0 = not even one row fits (W > 2558).
constexpr size_t kPfLdsMax = 60 * 1024;
static size_t pf_lds(int rb, int W) { return sizeof(float) * 2 * (size_t)(rb + 2) * (size_t)(W + 2); }
static int pf_rb(int N, int C, int H, int W) {
  int best = 0;
  long long best_cost = -1;
  for (int rb = fhb::kRBMax; rb >= 1; rb /= 2) {       // rounds of ~2 workgroups per CU x rows a workgroup walks (+ halo)
    if (pf_lds(rb, W) > kPfLdsMax) continue;
    if (rb < 8 && best > 0) break;                      // bands shorter than 8 rows only where nothing taller fits
    const long long blocks = (long long)N * ((H + rb - 1) / rb) * ((C + fhb::kCG - 1) / fhb::kCG);
    const long long cost = ((blocks + 511) / 512) * (long long)((rb < H ? rb : H) + 2);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rb; }
  }
  return best;
}
static int pf_bands(int N, int C, int H, int W) { const int rb = pf_rb(N, C, H, W); return rb > 0 ? (H + rb - 1) / rb : 0; }

FN2_API int fn2_predict_flow_conv_backward_supported(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (long long)H * W >= (1ll << 31) || N > 65535) return 0;
  const int bands = pf_bands(N, C, H, W);
  return bands > 0 && (long long)N * bands <= 65535;
}

FN2_API size_t fn2_predict_flow_conv_backward_workspace_bytes(int N, int C, int H, int W) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  const size_t parts = (size_t)N * pf_bands(N, C, H, W);
  return sizeof(float) * (parts * 18 * (size_t)C + parts * 2);
}

FN2_API int fn2_predict_flow_conv_backward(const float* bottom, int bottom_channels, int bottom_c0, const float* weight, const float* top_diff,
                                           float* bottom_diff, float* weight_diff, float* bias_diff, int N, int C, int H, int W,
                                           int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "predict_flow_conv_backward: bad shape [%d,%d,%d,%d]", N, C, H, W);
  if (bottom_c0 < 0 || bottom_c0 + C > bottom_channels) return fail(FN2_ERR_INVALID_ARG, "predict_flow_conv_backward: channel slice outside the blob");
  if (N == 0) return FN2_OK;
  if (!top_diff || ((weight_diff || bias_diff) && !bottom) || (bottom_diff && !weight)) return fail(FN2_ERR_INVALID_ARG, "predict_flow_conv_backward: null blob");
  if ((long long)H * W >= (1ll << 31) || N > 65535) return fail(FN2_ERR_UNSUPPORTED, "predict_flow_conv_backward: blob too large");
  hipStream_t st = as_stream(stream);
  if (weight_diff || bias_diff) {
    const size_t need = fn2_predict_flow_conv_backward_workspace_bytes(N, C, H, W);
    if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "predict_flow_conv_backward: workspace too small (%zu < %zu)", workspace_bytes, need);
    const int rb = pf_rb(N, C, H, W), bands = pf_bands(N, C, H, W), parts = N * bands;
    if (rb == 0) return fail(FN2_ERR_UNSUPPORTED, "predict_flow_conv_backward: rows of %d pixels are too wide for the LDS band", W);
    if (parts > 65535) return fail(FN2_ERR_UNSUPPORTED, "predict_flow_conv_backward: too many parts");
    float* partial = reinterpret_cast<float*>(workspace);
    float* bpart = partial + (size_t)parts * 18 * C;
    const size_t lds = pf_lds(rb, W);
    hipLaunchKernelGGL(fhb::pf_wgrad, dim3((unsigned)((C + fhb::kCG - 1) / fhb::kCG), (unsigned)parts), dim3(256), lds, st, bottom, bottom_channels, bottom_c0,
                       top_diff, partial, bpart, C, H, W, bands, rb);
    hipLaunchKernelGGL(fhb::pf_finalize, dim3((unsigned)C), dim3(64), 0, st, partial, bpart, weight_diff, bias_diff, C, parts, accumulate);
  }
  if (bottom_diff) {
    const unsigned bx = (unsigned)(((long long)H * W + 255) / 256);
    int cpb = 8;
    while (cpb < 64 && (long long)bx * ((C + cpb - 1) / cpb) * N >= 4096) cpb *= 2;
    hipLaunchKernelGGL(fhb::pf_dgrad, dim3(bx, (unsigned)((C + cpb - 1) / cpb), (unsigned)N), dim3(256), 0, st, top_diff, weight, bottom_diff, N, C, H, W, cpb);
  }
  return check_launch("predict_flow_conv_backward");
}

FN2_API size_t fn2_upsample_flow_deconv_backward_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return sizeof(float) * 66 * (size_t)N * (size_t)(((long long)H * W + 255) / 256);
}

FN2_API int fn2_upsample_flow_deconv_backward(const float* bottom, const float* weight, const float* top_diff, float* bottom_diff,
                                              float* weight_diff, float* bias_diff, int N, int H, int W, int accumulate,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv_backward: bad shape [%d,2,%d,%d]", N, H, W);
  if (N == 0) return FN2_OK;
  if (!bottom || !weight || !top_diff) return fail(FN2_ERR_INVALID_ARG, "upsample_flow_deconv_backward: null blob");
  if ((long long)H * W >= (1ll << 29) || N > 65535) return fail(FN2_ERR_UNSUPPORTED, "upsample_flow_deconv_backward: blob too large");
  const int want_w = (weight_diff || bias_diff) ? 1 : 0;
  const unsigned bx = (unsigned)(((long long)H * W + 255) / 256);
  float* partial = nullptr;
  if (want_w) {
    const size_t need = fn2_upsample_flow_deconv_backward_workspace_bytes(N, H, W);
    if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "upsample_flow_deconv_backward: workspace too small (%zu < %zu)", workspace_bytes, need);
    partial = reinterpret_cast<float*>(workspace);
  }
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(fhb::uf_backward, dim3(bx, (unsigned)N), dim3(256), 0, st, bottom, weight, top_diff, bottom_diff, partial, H, W, want_w);
  if (want_w) hipLaunchKernelGGL(fhb::uf_finalize, dim3(66), dim3(64), 0, st, partial, weight_diff, bias_diff, (int)(bx * (unsigned)N), accumulate);
  return check_launch("upsample_flow_deconv_backward");
}
