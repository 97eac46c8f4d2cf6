// ChannelNorm + Downsample for gfx950 (HBM-bound streaming kernels).
//
// ChannelNorm: ChannelNormLayer::Forward_gpu / Backward_gpu, reference
//   src/caffe/layers/channel_norm_layer.cu:16-90.
// Downsample : DownsampleLayer::Forward_gpu, reference src/caffe/layers/downsample_layer.cu:15-129.
#include "fn2_common.hpp"

#include <cmath>

namespace fn2 {

// grid: (pixel blocks, n) / (pixel blocks, n * C + c); 32-bit pixel index.
// bot / sub / top may be channel slices of wider blobs (sample stride = their blob's channel count).  SUB: the norm of bot - sub,
// i.e. the Eltwise{SUM, coeff 1, -1} in front of the layer in the FlowNet2 graphs folded in (the difference is rounded to fp32
// before it is squared, exactly as the Eltwise top would have been).
template <bool SUB>
__global__ void __launch_bounds__(256) channel_norm_fwd(const float* __restrict__ bot, const float* __restrict__ sub, float* __restrict__ top,
                                                        int N, int C, unsigned hw, int bctot, int bc0, int sctot, int sc0, int tctot, int tc0) {
  const unsigned s = blockIdx.x * 256u + threadIdx.x;
  if (s >= hw) return;
  for (unsigned n = blockIdx.y; n < (unsigned)N; n += gridDim.y) {
    const float* p = bot + ((size_t)n * bctot + bc0) * hw + s;
    const float* q = SUB ? sub + ((size_t)n * sctot + sc0) * hw + s : nullptr;
    float norm = 0.f;
    for (int c = 0; c < C; ++c) {
      float v = p[(size_t)c * hw];
      if constexpr (SUB) v = v - q[(size_t)c * hw];      // eltwise_layer.cu:  top = 1 * a + (-1) * b
      norm = fmaf(v, v, norm);          // :27-28
    }
    top[((size_t)n * tctot + tc0) * hw + s] = sqrtf(norm);   // :31-32
  }
}

__global__ void __launch_bounds__(256) channel_norm_bwd(const float* __restrict__ bot, const float* __restrict__ top,
                                                        const float* __restrict__ top_diff, float* __restrict__ bot_diff,
                                                        int N, int C, unsigned hw) {
  const unsigned s = blockIdx.x * 256u + threadIdx.x;
  if (s >= hw) return;
  for (unsigned pl = blockIdx.y; pl < (unsigned)N * C; pl += gridDim.y) {
    const unsigned n = pl / C;
    const size_t idx = (size_t)pl * hw + s;
    // :45 -- `top_data + 1e-9` promotes the quotient to double in the reference.
    bot_diff[idx] = (float)((double)(top_diff[(size_t)n * hw + s] * bot[idx]) / ((double)top[(size_t)n * hw + s] + 1e-9));
  }
}

struct DownArgs {
  int NC, Hin, Win, Hout, Wout, wradius, hradius;
  float widthScale, heightScale;
};

// grid: (output pixel blocks, n * C + c).  Tap bounds are clamped once (no test per tap), the row weight is hoisted and
// four taps of a row are in flight together: the reference's per-tap test serialises 121 load -> use round trips per
// output for the x4 reduction of a flow field.  Weights keep the reference's expression and order (:52).
// (bx, by, gy: the block's coordinates and the y extent of the grid it belongs to -- blockIdx / gridDim of its own launch, or its place in
// the job of a multi-scale launch, downsample_fwd_multi)
// (Column weights kept in LDS per thread, as the group kernels below keep their row weights, made this kernel SLOWER -- 20 -> 24 us at x4,
// 25 -> 34 us at x8: its time is the latency of 9-17 short rows per output, not the two divisions per tap.)
__device__ __forceinline__ void downsample_thread_body(const float* __restrict__ src, float* __restrict__ dst, const DownArgs& a, unsigned bx, unsigned by,
                                                       unsigned gy) {
  const unsigned hw_out = (unsigned)a.Hout * a.Wout;
  const unsigned pd = bx * 256u + threadIdx.x;
  if (pd >= hw_out) return;
  const int desty = pd / a.Wout, destx = pd - desty * a.Wout;
  const float botx = ((float)destx / (float)(a.Wout - 1)) * (float)(a.Win - 1);     // :27
  const float boty = ((float)desty / (float)(a.Hout - 1)) * (float)(a.Hin - 1);     // :28
  const int ibotx = (int)roundf(botx), iboty = (int)roundf(boty);                   // :30-31
  const int y0 = max(iboty - a.hradius, 0), y1 = min(iboty + a.hradius, a.Hin - 1);
  const int x0 = max(ibotx - a.wradius, 0), x1 = min(ibotx + a.wradius, a.Win - 1);
  for (unsigned cn = by; cn < (unsigned)a.NC; cn += gy) {
    const float* p = src + (size_t)cn * a.Hin * a.Win;
    float accum_value = 0.f, accum_weight = 0.f, accum_nan = 0.f;
    for (int ty = y0; ty <= y1; ++ty) {
      const float wy = fmaxf(0.0f, 1.0f - (fabsf((float)ty - boty) / a.heightScale));
      const float* row = p + (size_t)ty * a.Win;
      int tx = x0;
      for (; tx + 3 <= x1; tx += 4) {
        float sm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[j] = row[tx + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float sample = sm[j];
          float weight = fmaxf(0.0f, 1.0f - (fabsf((float)(tx + j) - botx) / a.widthScale)) * wy;   // :52
          if (sample != sample) { accum_nan += weight; sample = 0.f; weight = 0.f; }               // :53-57
          accum_value = fmaf(sample, weight, accum_value);
          accum_weight += weight;
        }
      }
      for (; tx <= x1; ++tx) {
        float sample = row[tx];
        float weight = fmaxf(0.0f, 1.0f - (fabsf((float)tx - botx) / a.widthScale)) * wy;
        if (sample != sample) { accum_nan += weight; sample = 0.f; weight = 0.f; }
        accum_value = fmaf(sample, weight, accum_value);
        accum_weight += weight;
      }
    }
    const size_t idx = (size_t)cn * hw_out + pd;
    if (accum_nan / accum_weight > 0.5f) dst[idx] = __builtin_bit_cast(float, 0x7fffffffu);   // :64-65
    else dst[idx] = accum_value / accum_weight;                                               // :67
  }
}

__global__ void __launch_bounds__(256) downsample_fwd(const float* __restrict__ src, float* __restrict__ dst, DownArgs a) {
  downsample_thread_body(src, dst, a, blockIdx.x, blockIdx.y, gridDim.y);
}

// Large reduction factors (the coarse scales of the multi-scale loss: 320x448 -> 5x7 is a 129 x 129 tap window per output).  The reference
// walks the window in one thread (:36-62); one thread per output left 280 threads with 16,641 dependent taps each -- 416 us per call, 14 % of a
// FlowNetC training step.  Here a GROUP of threads owns an output element: a wave (512 <= taps < 4096) or a whole workgroup (more).
// Round 6 layout: thread -> (column slot cx, row slot rs) of the window, CW = the power of two >= the window's width (at most the group).
// The weight of a tap is separable -- wx(bx) * wy(by), :52 -- and each factor holds an IEEE division: a thread keeps ONE wx for its column,
// the wy of the window's rows are computed once per output into LDS, so a tap costs a multiply, the NaN test and two adds instead of two
// divisions (the flat tap index of rounds 3-5 spent ~40 VALU instructions per tap: 30 us for the 9.3 M taps of the coarsest scale).  Loads
// run along rows (coalesced), four rows per thread in flight.  Order: a thread sums its column top to bottom (rows rs, rs + RS, ...), then a
// fixed butterfly over the lanes and the waves in wave order -- deterministic; another order than the reference's single thread (3e-6).
constexpr int kDownMaxRows = 2 * 512 + 1;       // rows of a window the LDS table holds (a reduction factor up to 512)

template <int G>        // threads per output element: 64 (a wave; four outputs per workgroup) or 256
__device__ __forceinline__ void downsample_group_body(const float* __restrict__ src, float* __restrict__ dst, const DownArgs& a, unsigned bx, unsigned gx,
                                                      float* wyt, float (*part)[4]) {
  constexpr int PER = 256 / G;                    // outputs a workgroup works on at a time
  const unsigned hw_out = (unsigned)a.Hout * a.Wout;
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned gt = G == 64 ? lane : tid;       // thread within its group
  float* wy = wyt + (G == 64 ? wave * kDownMaxRows : 0);
  const unsigned long long total = (unsigned long long)a.NC * hw_out;
  for (unsigned long long o = (unsigned long long)bx * PER + (G == 64 ? wave : 0); o < total; o += (unsigned long long)gx * PER) {
    const unsigned cn = (unsigned)(o / hw_out), pd = (unsigned)(o - (unsigned long long)cn * hw_out);
    const int desty = pd / a.Wout, destx = pd - desty * a.Wout;
    const float botx = ((float)destx / (float)(a.Wout - 1)) * (float)(a.Win - 1);     // :27
    const float boty = ((float)desty / (float)(a.Hout - 1)) * (float)(a.Hin - 1);     // :28
    const int ibotx = (int)roundf(botx), iboty = (int)roundf(boty);                   // :30-31
    const int y0 = max(iboty - a.hradius, 0), y1 = min(iboty + a.hradius, a.Hin - 1);
    const int x0 = max(ibotx - a.wradius, 0), x1 = min(ibotx + a.wradius, a.Win - 1);
    const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
    int cw = 1;
    while (cw < nx && cw < G) cw <<= 1;            // column slots (uniform over the group)
    const int rs_n = G / cw, cx = (int)gt & (cw - 1), rs = (int)gt / cw;
    if constexpr (G == 256) __syncthreads();       // (the previous output's table is still being read)
    for (int r = (int)gt; r < ny; r += G) wy[r] = fmaxf(0.0f, 1.0f - (fabsf((float)(y0 + r) - boty) / a.heightScale));
    if constexpr (G == 256) __syncthreads();
    const float* p = src + (size_t)cn * a.Hin * a.Win + (size_t)y0 * a.Win;
    float accum_value = 0.f, accum_weight = 0.f, accum_nan = 0.f;
    for (int c = cx; c < nx; c += cw) {            // (one pass unless the window is wider than the group)
      const int tbx = x0 + c;
      const float wx = fmaxf(0.0f, 1.0f - (fabsf((float)tbx - botx) / a.widthScale));
      const float* col = p + tbx;
      // four rows per thread in flight (sixteen, with the tail folded into a predicated round, ran 2.5x SLOWER: 19.6 -> 50.5 us at x64)
      int r = rs;
      for (; r + 3 * rs_n < ny; r += 4 * rs_n) {
        float sm[4], wr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { sm[j] = col[(size_t)(r + j * rs_n) * a.Win]; wr[j] = wy[r + j * rs_n]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float sample = sm[j], weight = wx * wr[j];                                             // :52
          if (sample != sample) { accum_nan += weight; sample = 0.f; weight = 0.f; }           // :53-57
          accum_value = fmaf(sample, weight, accum_value);
          accum_weight += weight;
        }
      }
      for (; r < ny; r += rs_n) {
        float sample = col[(size_t)r * a.Win], weight = wx * wy[r];
        if (sample != sample) { accum_nan += weight; sample = 0.f; weight = 0.f; }
        accum_value = fmaf(sample, weight, accum_value);
        accum_weight += weight;
      }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      accum_value += __shfl_xor(accum_value, m, 64);
      accum_weight += __shfl_xor(accum_weight, m, 64);
      accum_nan += __shfl_xor(accum_nan, m, 64);
    }
    if constexpr (G == 64) {
      if (lane == 0) {
        if (accum_nan / accum_weight > 0.5f) dst[o] = __builtin_bit_cast(float, 0x7fffffffu);   // :64-65
        else dst[o] = accum_value / accum_weight;                                               // :67
      }
    } else {
      if (lane == 0) { part[0][wave] = accum_value; part[1][wave] = accum_weight; part[2][wave] = accum_nan; }
      __syncthreads();
      if (tid == 0) {
        const float v = ((part[0][0] + part[0][1]) + part[0][2]) + part[0][3];
        const float w = ((part[1][0] + part[1][1]) + part[1][2]) + part[1][3];
        const float nn = ((part[2][0] + part[2][1]) + part[2][2]) + part[2][3];
        if (nn / w > 0.5f) dst[o] = __builtin_bit_cast(float, 0x7fffffffu);   // :64-65
        else dst[o] = v / w;                                                   // :67
      }
    }
  }
}

__global__ void __launch_bounds__(256) downsample_fwd_wave(const float* __restrict__ src, float* __restrict__ dst, DownArgs a) {
  __shared__ float wyt[4 * kDownMaxRows];
  downsample_group_body<64>(src, dst, a, blockIdx.x, gridDim.x, wyt, nullptr);
}

__global__ void __launch_bounds__(256) downsample_fwd_block(const float* __restrict__ src, float* __restrict__ dst, DownArgs a) {
  __shared__ float wyt[4 * kDownMaxRows];
  __shared__ float part[3][4];
  downsample_group_body<256>(src, dst, a, blockIdx.x, gridDim.x, wyt, part);
}

// Several top sizes of ONE bottom in one launch (round 6): the ground-truth pyramid of the multi-scale loss is five Downsample layers on the
// same blob (320x448 -> 80x112 ... 5x7), each a latency-bound launch of 18-37 us that fills a fraction of the chip; as one grid they run side by
// side.  A job = one top size with the decomposition its own launch would have had (thread / wave / workgroup per output element, same tap order:
// the result has the bits of fn2_downsample_forward); the coarsest (longest-running) jobs come first in the grid.
constexpr int kDownMaxJobs = 8;
struct DownMulti {
  DownArgs a[kDownMaxJobs];
  float* dst[kDownMaxJobs];
  unsigned first[kDownMaxJobs + 1];       // first block of job j; first[count] = the grid
  unsigned gx[kDownMaxJobs], gy[kDownMaxJobs];
  int mode[kDownMaxJobs];                 // 0 thread, 1 wave, 2 workgroup per output element
  int count;
};

__global__ void __launch_bounds__(256) downsample_fwd_multi(const float* __restrict__ src, DownMulti m) {
  __shared__ float part[3][4];
  __shared__ float wyt[4 * kDownMaxRows];
  int j = 0;
  while (j + 1 < m.count && blockIdx.x >= m.first[j + 1]) ++j;
  const unsigned b = blockIdx.x - m.first[j];
  // (the job index is uniform over the workgroup; a switch over constant indices keeps the descriptors in SGPRs)
#pragma unroll
  for (int k = 0; k < kDownMaxJobs; ++k) {
    if (k != j) continue;
    if (m.mode[k] == 2) downsample_group_body<256>(src, m.dst[k], m.a[k], b, m.gx[k], wyt, part);
    else if (m.mode[k] == 1) downsample_group_body<64>(src, m.dst[k], m.a[k], b, m.gx[k], wyt, nullptr);
    else downsample_thread_body(src, m.dst[k], m.a[k], b % m.gx[k], b / m.gx[k], m.gy[k]);
  }
}

}  // namespace fn2

using namespace fn2;

FN2_API int fn2_channel_norm_forward_slices(const float* bottom, int bottom_channels, int bottom_c0,
                                            const float* minus, int minus_channels, int minus_c0,
                                            float* top, int top_channels, int top_c0, int N, int C, int H, int W, void* stream) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "channel_norm: bad shape");
  if (bottom_c0 < 0 || bottom_c0 + C > bottom_channels || top_c0 < 0 || top_c0 + 1 > top_channels ||
      (minus && (minus_c0 < 0 || minus_c0 + C > minus_channels)))
    return fail(FN2_ERR_INVALID_ARG, "channel_norm: channel slice outside its blob");
  if (N == 0) return FN2_OK;
  if (!bottom || !top) return fail(FN2_ERR_INVALID_ARG, "channel_norm: NULL blob pointer");
  if ((long long)H * W >= (1ll << 31)) return fail(FN2_ERR_UNSUPPORTED, "channel_norm: plane too large");
  const unsigned hw = (unsigned)H * W;
  const dim3 grid((hw + 255) / 256, (unsigned)(N < 65535 ? N : 65535));
  if (minus) hipLaunchKernelGGL((channel_norm_fwd<true>), grid, dim3(256), 0, as_stream(stream), bottom, minus, top, N, C, hw,
                                bottom_channels, bottom_c0, minus_channels, minus_c0, top_channels, top_c0);
  else hipLaunchKernelGGL((channel_norm_fwd<false>), grid, dim3(256), 0, as_stream(stream), bottom, minus, top, N, C, hw,
                          bottom_channels, bottom_c0, 0, 0, top_channels, top_c0);
  return check_launch("channel_norm_forward");
}

FN2_API int fn2_channel_norm_forward(const float* bottom, float* top, int N, int C, int H, int W, void* stream) {
  return fn2_channel_norm_forward_slices(bottom, C, 0, nullptr, 0, 0, top, 1, 0, N, C, H, W, stream);
}

FN2_API int fn2_channel_norm_backward(const float* bottom, const float* top, const float* top_diff, float* bottom_diff,
                                      int N, int C, int H, int W, void* stream) {
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "channel_norm: bad shape");
  if (N == 0) return FN2_OK;
  if (!bottom || !top || !top_diff || !bottom_diff) return fail(FN2_ERR_INVALID_ARG, "channel_norm: NULL blob pointer");
  if ((long long)H * W >= (1ll << 31)) return fail(FN2_ERR_UNSUPPORTED, "channel_norm: plane too large");
  const unsigned hw = (unsigned)H * W;
  const long long planes = (long long)N * C;
  hipLaunchKernelGGL(channel_norm_bwd, dim3((hw + 255) / 256, (unsigned)(planes < 65535 ? planes : 65535)), dim3(256), 0, as_stream(stream),
                     bottom, top, top_diff, bottom_diff, N, C, hw);
  return check_launch("channel_norm_backward");
}

// geometry + decomposition of one Downsample: 0 thread, 1 wave, 2 workgroup per output element (by the tap count); -1: plane too large
static int g_down_wave_taps = 256;       // windows from this many taps up get a wave per output element
#define kDownWaveTaps g_down_wave_taps
static int down_plan(int N, int C, int Hin, int Win, int Hout, int Wout, DownArgs* out, unsigned* gx, unsigned* gy) {
  DownArgs a;
  a.NC = N * C; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout;
  a.widthScale = (float)(Win - 1) / (float)(Wout - 1);     // :104
  a.heightScale = (float)(Hin - 1) / (float)(Hout - 1);    // :105
  a.wradius = (int)std::ceil(a.widthScale);                // :107
  a.hradius = (int)std::ceil(a.heightScale);               // :108
  if ((long long)Hout * Wout >= (1ll << 31)) return -1;
  *out = a;
  const long long taps = (long long)(2 * a.wradius + 1) * (2 * a.hradius + 1);
  const long long outs = (long long)a.NC * Hout * Wout;
  *gy = 1;
  const bool table_fits = 2 * a.hradius + 1 <= kDownMaxRows;      // (the LDS table of row weights; a taller window runs on the thread kernel)
  if (taps >= 4096 && table_fits) { *gx = (unsigned)(outs < 65536 ? outs : 65536); return 2; }
  if (taps >= kDownWaveTaps && table_fits) { const long long blocks = (outs + 3) / 4; *gx = (unsigned)(blocks < 65536 ? blocks : 65536); return 1; }
  *gx = ((unsigned)Hout * Wout + 255) / 256;
  *gy = (unsigned)(a.NC < 65535 ? a.NC : 65535);
  return 0;
}

FN2_API int fn2_downsample_forward(const float* bottom, float* top, int N, int C, int Hin, int Win, int Hout, int Wout,
                                   void* stream) {
  if (N < 0 || C < 1 || Hin < 1 || Win < 1) return fail(FN2_ERR_INVALID_ARG, "downsample: bad bottom shape");
  if (Hout < 1 || Wout < 1) return fail(FN2_ERR_INVALID_ARG, "DownsampleLayer must have top_height > 0 and top_width > 0");
  if (N == 0) return FN2_OK;
  if (!bottom || !top) return fail(FN2_ERR_INVALID_ARG, "downsample: NULL blob pointer");
  hipStream_t st = as_stream(stream);
  if (Hin == Hout && Win == Wout) {   // downsample_layer.cpp:53-56 shares the blob; we copy
    if (bottom != top &&
        hipMemcpyAsync(top, bottom, sizeof(float) * (size_t)N * C * Hin * Win, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(FN2_ERR_LAUNCH, "downsample: hipMemcpyAsync failed");
    return FN2_OK;
  }
  // a 1-pixel-high or -wide top makes the reference divide by zero (:104-105: scale = inf, radius = (int)ceil(inf) is undefined)
  if (Hout < 2 || Wout < 2) return fail(FN2_ERR_INVALID_ARG, "downsample: top_height and top_width must be at least 2 when the size changes (downsample_layer.cu:104-105 divides by size - 1)");
  DownArgs a;
  unsigned gx, gy;
  const int mode = down_plan(N, C, Hin, Win, Hout, Wout, &a, &gx, &gy);
  if (mode < 0) return fail(FN2_ERR_UNSUPPORTED, "downsample: plane too large");
  if (mode == 2) hipLaunchKernelGGL(downsample_fwd_block, dim3(gx), dim3(256), 0, st, bottom, top, a);
  else if (mode == 1) hipLaunchKernelGGL(downsample_fwd_wave, dim3(gx), dim3(256), 0, st, bottom, top, a);
  else hipLaunchKernelGGL(downsample_fwd, dim3(gx, gy), dim3(256), 0, st, bottom, top, a);
  return check_launch("downsample_forward");
}

FN2_API int fn2_downsample_forward_multi(const float* bottom, float* const* tops, const int* top_heights, const int* top_widths, int count,
                                         int N, int C, int Hin, int Win, void* stream) {
  if (count < 1 || count > kDownMaxJobs) return fail(FN2_ERR_INVALID_ARG, "downsample_multi: 1 to %d top sizes per launch (got %d)", kDownMaxJobs, count);
  if (N < 0 || C < 1 || Hin < 1 || Win < 1) return fail(FN2_ERR_INVALID_ARG, "downsample: bad bottom shape");
  if (!tops || !top_heights || !top_widths) return fail(FN2_ERR_INVALID_ARG, "downsample_multi: NULL size / pointer table");
  if (N == 0) return FN2_OK;
  if (!bottom) return fail(FN2_ERR_INVALID_ARG, "downsample: NULL blob pointer");
  DownMulti m;
  int order[kDownMaxJobs], modes[kDownMaxJobs];
  DownArgs as[kDownMaxJobs];
  unsigned gxs[kDownMaxJobs], gys[kDownMaxJobs];
  for (int j = 0; j < count; ++j) {
    if (top_heights[j] < 2 || top_widths[j] < 2 || (top_heights[j] == Hin && top_widths[j] == Win))
      return fail(FN2_ERR_INVALID_ARG, "downsample_multi: every top must be at least 2 x 2 and differ from the bottom's size (top %d: %d x %d)", j, top_heights[j],
                  top_widths[j]);
    if (!tops[j]) return fail(FN2_ERR_INVALID_ARG, "downsample: NULL blob pointer");
    modes[j] = down_plan(N, C, Hin, Win, top_heights[j], top_widths[j], &as[j], &gxs[j], &gys[j]);
    if (modes[j] < 0) return fail(FN2_ERR_UNSUPPORTED, "downsample: plane too large");
    order[j] = j;
  }
  // longest-running decompositions first (stable): workgroup per output, wave per output, thread per output
  for (int i = 1; i < count; ++i)
    for (int k = i; k > 0 && modes[order[k]] > modes[order[k - 1]]; --k) { const int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t; }
  unsigned long long blocks = 0;
  for (int i = 0; i < count; ++i) {
    const int j = order[i];
    m.a[i] = as[j]; m.dst[i] = tops[j]; m.mode[i] = modes[j]; m.gx[i] = gxs[j]; m.gy[i] = gys[j];
    m.first[i] = (unsigned)blocks;
    blocks += (unsigned long long)gxs[j] * gys[j];
    if (blocks >= (1ull << 31)) return fail(FN2_ERR_UNSUPPORTED, "downsample_multi: grid too large");
  }
  for (int i = count; i < kDownMaxJobs; ++i) { m.a[i] = m.a[0]; m.dst[i] = nullptr; m.mode[i] = 0; m.gx[i] = 1; m.gy[i] = 1; m.first[i] = (unsigned)blocks; }
  m.first[count] = (unsigned)blocks;
  m.count = count;
  hipLaunchKernelGGL(downsample_fwd_multi, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), bottom, m);
  return check_launch("downsample_forward_multi");
}
