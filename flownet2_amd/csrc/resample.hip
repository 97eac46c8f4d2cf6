// Resample (NEAREST / LINEAR / CUBIC) for gfx950, forward only.
//
// Replaces ResampleLayer::Forward_gpu (reference: src/caffe/layers/resample_layer.cu:128-206).
// Arithmetic follows InterpolationKernel (:39-95) tap for tap -- including the swapped half-pixel
// offsets (x uses fy/2, y uses fx/2, :62-63) and the sum/wsum edge renormalisation (:93) -- but
// the row coefficient is hoisted out of the column loop and taps outside the image are skipped by
// clamping the loop bounds instead of testing every tap.
#include "fn2_common.hpp"

#include <cmath>

// No implicit a * b + c fusion in this file: the three LINEAR kernels (per output pixel, lean, integer up-sampling) must round the
// coefficient products and their sums alike to stay bit-identical to each other; the fused steps are the explicit fmaf() calls.
#pragma clang fp contract(off)

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
  int ppt;          // planes per thread (blockIdx.y walks plane groups)
  // round 3: the layers AROUND a Resample in the FlowNet2 graphs folded in.  in_scale: the Eltwise{coeff} in front of it (every tap is
  // multiplied -- and rounded -- before it is weighted, as the Eltwise top would have been; 1.0f is exact).  The top may be a channel
  // slice [oc0, oc0 + C) of a blob with octot channels (the Concat behind it), and a second top out2 = out * out2_scale (the Eltwise
  // behind it, rounded from the rounded out) may go to a slice of another blob.
  int C, octot, oc0, o2ctot, o2c0;
  float in_scale, out2_scale;
};

__device__ __forceinline__ size_t top_plane(int plane, int C, int ctot, int c0) {     // plane = n * C + c of the logical [N, C] top
  const int n = plane / C;
  return (size_t)n * ctot + c0 + (plane - n * C);
}
// one rounded product.  Every use below feeds a store or an OPERAND of an explicit fmaf(): there is no a * b + c expression hipcc could
// contract it into (the bit-equality with the separate Eltwise pass is tested on every kernel of this file).
__device__ __forceinline__ float scaled(float v, float s) { return v * s; }

// Indexing of both kernels: blockIdx.x * 256 + tid = output pixel (32-bit), blockIdx.y = group of `ppt` (n, c) planes.
// Everything that depends on the pixel only -- source position, tap range, coefficients -- is computed once and reused
// for the planes of the group (the first version decoded a 64-bit flat index per element and re-evaluated the
// coefficient of every one of the (2r+1)^2 taps: 30 us for a 10 MB up-sampling).

template <bool EXTRA>
__global__ void __launch_bounds__(256) resample_nearest(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out2, ResampleArgs a) {
  const unsigned hw_out = (unsigned)a.Hout * a.Wout, hw_in = (unsigned)a.Hin * a.Win;
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= hw_out) return;
  const int y_out = p / a.Wout, x_out = p - y_out * a.Wout;
  const float x_in = x_out * a.fx + a.fy / 2.0f - 0.5f;   // :117
  const float y_in = y_out * a.fy + a.fx / 2.0f - 0.5f;   // :118
  int xr = (int)roundf(x_in), yr = (int)roundf(y_in);
  // The reference reads in_ptr[yr*W+xr] unclamped (:123); clamp instead of faulting.
  xr = min(max(xr, 0), a.Win - 1);
  yr = min(max(yr, 0), a.Hin - 1);
  const unsigned src = (unsigned)yr * a.Win + xr;
  for (int c = blockIdx.y * a.ppt; c < min(a.NC, (int)(blockIdx.y + 1) * a.ppt); ++c) {
    const float raw = in[(size_t)c * hw_in + src];
    const float v = EXTRA ? scaled(raw, a.in_scale) : raw;
    out[(EXTRA ? top_plane(c, a.C, a.octot, a.oc0) : (size_t)c) * hw_out + p] = v;
    if (EXTRA && out2) out2[top_plane(c, a.C, a.o2ctot, a.o2c0) * hw_out + p] = scaled(v, a.out2_scale);
  }
}

// FAST: tap radius <= 2 on both axes (every up-sampling and same-size call): the 5 + 5 coefficients live in registers.
template <bool CUBIC, bool FAST, bool EXTRA>     // EXTRA: input scaling / channel-slice tops / second top; false = the plain layer
__global__ void __launch_bounds__(256) resample_interp(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out2, ResampleArgs a) {
  const unsigned hw_out = (unsigned)a.Hout * a.Wout, hw_in = (unsigned)a.Hin * a.Win;
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= hw_out) return;
  const int y_out = p / a.Wout, x_out = p - y_out * a.Wout;
  const float x_in = x_out * a.fx + a.fy / 2.0f - 0.5f;   // :62
  const float y_in = y_out * a.fy + a.fx / 2.0f - 0.5f;   // :63
  const int xr = (int)roundf(x_in), yr = (int)roundf(y_in);
  const int c_lo = blockIdx.y * a.ppt, c_hi = min(a.NC, c_lo + a.ppt);
  if constexpr (FAST) {
    // :87/:89 -- the reference evaluates ((ax*k(ax*dx))*ay)*k(ay*dy); same association: px = (ax*k(ax*dx))*ay per column,
    // ky per row, taps in the reference's order (rows outer, columns inner).  A tap outside the image is skipped by the
    // reference; here it gets weight 0 and a 0 sample, which leaves sum and wsum bit-identical.
    float px[5], ky[5];
    unsigned xo[5], yo[5], mx = 0u, my = 0u;     // mx / my: which of the 5 columns / rows are taps inside the image
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int x = xr - 2 + i, y = yr - 2 + i;
      const bool okx = x >= 0 && x < a.Win && (i >= 2 - a.rx && i <= 2 + a.rx);
      const bool oky = y >= 0 && y < a.Hin && (i >= 2 - a.ry && i <= 2 + a.ry);
      const float kx = CUBIC ? bicubic_coeff(a.ax * (x_in - x)) : triangle_coeff(a.ax * (x_in - x));
      const float kyv = CUBIC ? bicubic_coeff(a.ay * (y_in - y)) : triangle_coeff(a.ay * (y_in - y));
      px[i] = okx ? a.ax * kx * a.ay : 0.f;
      ky[i] = oky ? kyv : 0.f;
      xo[i] = okx ? (unsigned)x : 0u;
      yo[i] = oky ? (unsigned)y * a.Win : 0u;
      mx |= (okx ? 1u : 0u) << i;
      my |= (oky ? 1u : 0u) << i;
    }
    float wsum = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int i = 0; i < 5; ++i) wsum += px[i] * ky[j];
    for (int c = c_lo; c < c_hi; ++c) {
      const float* src = in + (size_t)c * hw_in;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const float w = px[i] * ky[j];
          // in-image taps are read even when their coefficient is 0: a NaN there poisons the result in the reference too
          const float tap = ((mx >> i) & (my >> j) & 1u) ? src[yo[j] + xo[i]] : 0.f;      // (the load stays unconditional: clamped address)
          sum = fmaf(w, EXTRA ? scaled(tap, a.in_scale) : tap, sum);
        }
      const float v = (!wsum) ? 0.f : (sum / wsum);   // :93
      out[(EXTRA ? top_plane(c, a.C, a.octot, a.oc0) : (size_t)c) * hw_out + p] = v;
      if (EXTRA && out2) out2[top_plane(c, a.C, a.o2ctot, a.o2c0) * hw_out + p] = scaled(v, a.out2_scale);
    }
  } else {
    const int y0 = max(yr - a.ry, 0), y1 = min(yr + a.ry, a.Hin - 1);
    const int x0 = max(xr - a.rx, 0), x1 = min(xr + a.rx, a.Win - 1);
    for (int c = c_lo; c < c_hi; ++c) {
      const float* src = in + (size_t)c * hw_in;
      float sum = 0.f, wsum = 0.f;
      for (int y = y0; y <= y1; ++y) {
        const float kyv = CUBIC ? bicubic_coeff(a.ay * (y_in - y)) : triangle_coeff(a.ay * (y_in - y));
        for (int x = x0; x <= x1; ++x) {
          const float dx = x_in - x;
          const float w = a.ax * (CUBIC ? bicubic_coeff(a.ax * dx) : triangle_coeff(a.ax * dx)) * a.ay * kyv;
          const float tap = src[(size_t)y * a.Win + x];
          sum = fmaf(w, EXTRA ? scaled(tap, a.in_scale) : tap, sum);
          wsum += w;
        }
      }
      const float v = (!wsum) ? 0.f : (sum / wsum);   // :93
      out[(EXTRA ? top_plane(c, a.C, a.octot, a.oc0) : (size_t)c) * hw_out + p] = v;
      if (EXTRA && out2) out2[top_plane(c, a.C, a.o2ctot, a.o2c0) * hw_out + p] = scaled(v, a.out2_scale);
    }
  }
}

// LINEAR up-sampling by any factor, the identity included (fx, fy <= 1: unit tap scale ax == ay == 1, radius 2 on both axes).
// The triangle kernel has support 1, so of the 25 taps the reference visits (:75-92) at most 2 x 2 carry a non-zero coefficient: the
// columns / rows on either side of the source position.  The other in-image taps enter with coefficient +0: for FINITE samples
// fmaf(+0, tap, sum) == sum bit for bit (the running sum starts at +0 and can never become -0), and wsum likewise -- so they matter
// only when a sample is NaN / Inf (0 * NaN poisons the sum in the reference).  A workgroup therefore scans the input footprint of its
// 16 x 64 output tile once (coalesced; it also pulls the rows into the cache), and
//   * every sample finite (any real image or flow): 4 loads + 4 fmaf per output, in the reference's tap order;
//   * otherwise: the 25-tap loop of resample_interp, tap for tap.
// Same bits as resample_interp<false, true, EXTRA> in both cases (tests/test_gpu_parity.py: random sizes, NaN / Inf planted).
// The chip retires ~40 T lane-instructions/s against 8 TB/s: at 8 bytes per output a streaming kernel that wants a third of the HBM peak
// has ~100 instructions per output, so everything that depends on the pixel only (tap offsets, the 4 coefficient products, wsum) is
// computed once per thread and reused for its rows and planes; per output and plane there are 4 loads, 4 fmaf, the division, the store.
constexpr int kLeanTW = 64, kLeanTH = 32, kLeanRows = 8, kLeanPlanes = 2;     // output tile of a workgroup: 64 columns x 32 rows, thread (tx, ty) takes rows ty, ty + 4, ...
constexpr int kLeanFR = kLeanTH + 6, kLeanLW = 80;                              // footprint rows (tile + 2 + 3 + rounding), LDS row stride (>= 76 columns)
constexpr int kLeanScanVec = 3, kLeanScanScalar = 11;                           // footprint <= 38 rows x 76 (70) columns: 16-byte / 4-byte scan items per thread

// The exact rows of resample_linear_lean: resample_interp's 25-tap loop for output column x_out, rows y_first, y_first + 4, ... <= y_last.
// Not inlined: its loop-invariant coefficient set-up would otherwise be hoisted in front of the plane loop of the caller and executed by
// every thread of every tile (it was: 250 of the kernel's 400 set-up instructions), for a path real inputs never take.
template <bool EXTRA>
__device__ __attribute__((noinline)) void resample_exact_rows(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dst2,
                                                               const ResampleArgs& a, int x_out, bool live_x, int y_first, int y_last) {
  const float x_in = x_out * a.fx + a.fy / 2.0f - 0.5f;   // :62
  const int xr = (int)roundf(x_in);
  float px[5]; unsigned xo[5], mx = 0u;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int x = xr - 2 + i;
    const bool okx = x >= 0 && x < a.Win;
    px[i] = okx ? a.ax * triangle_coeff(a.ax * (x_in - x)) * a.ay : 0.f;
    xo[i] = okx ? (unsigned)x : 0u;
    mx |= (okx ? 1u : 0u) << i;
  }
  for (int y_out = y_first; y_out <= y_last; y_out += 4) {
    const float y_in = y_out * a.fy + a.fx / 2.0f - 0.5f;   // :63
    const int yr = (int)roundf(y_in);
    float ky[5]; unsigned yo[5], my = 0u;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int y = yr - 2 + j;
      const bool oky = y >= 0 && y < a.Hin;
      ky[j] = oky ? triangle_coeff(a.ay * (y_in - y)) : 0.f;
      yo[j] = oky ? (unsigned)y * a.Win : 0u;
      my |= (oky ? 1u : 0u) << j;
    }
    float sum = 0.f, ws = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float wt = px[i] * ky[j];
        const float tap = ((mx >> i) & (my >> j) & 1u) ? src[yo[j] + xo[i]] : 0.f;
        sum = fmaf(wt, EXTRA ? scaled(tap, a.in_scale) : tap, sum);
        ws += wt;
      }
    const float v = (!ws) ? 0.f : (sum / ws);   // :93
    if (live_x) {
      dst[(size_t)y_out * a.Wout + x_out] = v;
      if (EXTRA && dst2) dst2[(size_t)y_out * a.Wout + x_out] = scaled(v, a.out2_scale);
    }
  }
}

template <bool EXTRA>
__global__ void __launch_bounds__(256) resample_linear_lean(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out2, ResampleArgs a) {
  using f2 = __attribute__((ext_vector_type(2))) float;
  using f4 = __attribute__((ext_vector_type(4))) float;
  const unsigned hw_out = (unsigned)a.Hout * a.Wout, hw_in = (unsigned)a.Hin * a.Win;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int x0 = blockIdx.x * kLeanTW, y0 = blockIdx.y * kLeanTH;
  const bool live_x = x0 + tx < a.Wout;
  const int x_out = live_x ? x0 + tx : a.Wout - 1;     // idle columns compute the last one (valid addresses), store nothing
  const int xh_out = min(x0 + kLeanTW, a.Wout) - 1, yh_out = min(y0 + kLeanTH, a.Hout) - 1;
  // ---- footprint of the tile: the rows / columns its 5 x 5 windows can touch (window centre = round(source position); monotonic maps).
  // Computed first, from the tile corners alone, so that the loads below are in flight while the coefficient tables are built.
  int fx0 = max((int)roundf(x0 * a.fx + a.fy / 2.0f - 0.5f) - 2, 0), fx1 = min((int)roundf(xh_out * a.fx + a.fy / 2.0f - 0.5f) + 2, a.Win - 1);
  const int fy0 = max((int)roundf(y0 * a.fy + a.fx / 2.0f - 0.5f) - 2, 0), fy1 = min((int)roundf(yh_out * a.fy + a.fx / 2.0f - 0.5f) + 2, a.Hin - 1);
  const bool vec = (a.Win & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;       // 16-byte copies of whole aligned column quads
  if (vec) { fx0 &= ~3; fx1 |= 3; }
  const unsigned ncol = vec ? (unsigned)(fx1 - fx0 + 1) >> 2 : (unsigned)(fx1 - fx0 + 1);     // items per footprint row (<= 20 / <= 70)
  const unsigned nitem = ncol * (unsigned)(fy1 - fy0 + 1);
  const unsigned magic = 0xffffffffu / ncol + 1u;                                               // idx / ncol == umulhi(idx, magic) for idx < 2^16; ncol == 1 wraps to 0: handled at the uses
  const int c_lo = blockIdx.z * kLeanPlanes;
  // ---- the footprint of every plane of the group goes to LDS in one sweep of coalesced loads (each sample leaves the L2 once; the four
  // taps of an output are LDS reads); on the way 0 * sample is summed up: NaN exactly when a sample is NaN / Inf (the reference's own
  // poisoning term).  Static trip counts: all loads are in flight before the first is waited for.
  f4 v[kLeanScanVec][kLeanPlanes];
  if (vec) {
#pragma unroll
    for (int it = 0; it < kLeanScanVec; ++it) {
      const unsigned idx = threadIdx.x + 256u * it;
      const unsigned r = ncol == 1u ? idx : __umulhi(idx, magic), q = idx - r * ncol;
      const size_t goff = (size_t)(fy0 + (int)r) * a.Win + fx0 + 4 * q;
#pragma unroll
      for (int pl = 0; pl < kLeanPlanes; ++pl)
        v[it][pl] = idx < nitem ? *reinterpret_cast<const f4*>(in + (size_t)min(c_lo + pl, a.NC - 1) * hw_in + goff) : f4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- coefficient tables of the tile, one entry per thread: 64 columns, 32 rows (the expressions of resample_interp: same bits).
  // Entry = first tap c (taps c, c + 1: the only ones with |pos - tap| < 1) and the two coefficients (0 for a tap outside the image).
  __shared__ int s_c[kLeanTW + kLeanTH];
  __shared__ float s_k[kLeanTW + kLeanTH][2];
  __shared__ unsigned s_bad;                       // bit pl: plane pl of this workgroup has a NaN / Inf under the tile
  if (threadIdx.x == 255) s_bad = 0u;
  if (threadIdx.x < kLeanTW + kLeanTH) {
    const bool col = threadIdx.x < kLeanTW;
    const int o = col ? min(x0 + (int)threadIdx.x, a.Wout - 1) : min(y0 + (int)threadIdx.x - kLeanTW, a.Hout - 1);
    const float pos = col ? o * a.fx + a.fy / 2.0f - 0.5f      // :62
                          : o * a.fy + a.fx / 2.0f - 0.5f;     // :63
    const int r = (int)roundf(pos);
    const int c = pos >= (float)r ? r : r - 1;
    const int lim = col ? a.Win : a.Hin;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = c + i;
      const bool ok = t >= 0 && t < lim;
      const float k = col ? a.ax * triangle_coeff(a.ax * (pos - t)) * a.ay : triangle_coeff(a.ay * (pos - t));
      s_k[threadIdx.x][i] = ok ? k : 0.f;
    }
    s_c[threadIdx.x] = c;
  }
  __syncthreads();
  const int cx = s_c[tx];
  const float pxl[2] = {s_k[tx][0], s_k[tx][1]};
  __shared__ __attribute__((aligned(16))) float s_img[kLeanPlanes][kLeanFR * kLeanLW];
  f2 poison[kLeanPlanes];
#pragma unroll
  for (int pl = 0; pl < kLeanPlanes; ++pl) poison[pl] = f2{0.f, 0.f};
  if (vec) {
#pragma unroll
    for (int it = 0; it < kLeanScanVec; ++it) {
      const unsigned idx = threadIdx.x + 256u * it;
      const unsigned r = ncol == 1u ? idx : __umulhi(idx, magic), q = idx - r * ncol;
#pragma unroll
      for (int pl = 0; pl < kLeanPlanes; ++pl) {
        f4 t = v[it][pl];
        if constexpr (EXTRA) t *= a.in_scale;
        poison[pl] = __builtin_elementwise_fma(f2{0.f, 0.f}, f2{t[0], t[1]}, poison[pl]);
        poison[pl] = __builtin_elementwise_fma(f2{0.f, 0.f}, f2{t[2], t[3]}, poison[pl]);
        if (idx < nitem) *reinterpret_cast<f4*>(&s_img[pl][r * kLeanLW + 4 * q]) = t;
      }
    }
  } else {
#pragma unroll 4
    for (int it = 0; it < kLeanScanScalar; ++it) {
      const unsigned idx = threadIdx.x + 256u * it;
      const unsigned r = ncol == 1u ? idx : __umulhi(idx, magic), q = idx - r * ncol;
      const size_t goff = (size_t)(fy0 + (int)r) * a.Win + fx0 + q;
#pragma unroll
      for (int pl = 0; pl < kLeanPlanes; ++pl) {
        float t = idx < nitem ? in[(size_t)min(c_lo + pl, a.NC - 1) * hw_in + goff] : 0.f;
        if constexpr (EXTRA) t = scaled(t, a.in_scale);
        poison[pl][0] = fmaf(0.f, t, poison[pl][0]);
        if (idx < nitem) s_img[pl][r * kLeanLW + q] = t;
      }
    }
  }
  unsigned badmask = 0u;
#pragma unroll
  for (int pl = 0; pl < kLeanPlanes; ++pl) {
    const float p = poison[pl][0] + poison[pl][1];
    badmask |= (p != p ? 1u : 0u) << pl;
  }
  if (badmask) atomicOr(&s_bad, badmask);
  __syncthreads();
  const unsigned verdict = s_bad;
  // ---- rows: per row the four coefficient products in the reference's order and their sum, shared by the planes; a tap outside the
  // image has coefficient 0 and reads a (finite) sample of the footprint instead
  const int xl0 = min(max(cx, 0), a.Win - 1) - fx0, xl1 = min(max(cx + 1, 0), a.Win - 1) - fx0;
  char* dstb[kLeanPlanes]; char* dst2b[kLeanPlanes];
#pragma unroll
  for (int pl = 0; pl < kLeanPlanes; ++pl) {
    const int c = min(c_lo + pl, a.NC - 1);
    dstb[pl] = reinterpret_cast<char*>(out + (EXTRA ? top_plane(c, a.C, a.octot, a.oc0) : (size_t)c) * hw_out);
    dst2b[pl] = (EXTRA && out2) ? reinterpret_cast<char*>(out2 + top_plane(c, a.C, a.o2ctot, a.o2c0) * hw_out) : nullptr;
  }
  if (verdict) {      // a NaN / Inf somewhere under this tile: those planes take the 25 taps of the reference, zero coefficients included
#pragma unroll
    for (int pl = 0; pl < kLeanPlanes; ++pl)
      if (c_lo + pl < a.NC && ((verdict >> pl) & 1u))
        resample_exact_rows<EXTRA>(in + (size_t)(c_lo + pl) * hw_in, reinterpret_cast<float*>(dstb[pl]), reinterpret_cast<float*>(dst2b[pl]), a, x_out, live_x,
                                   y0 + ty, yh_out);
  }
#pragma unroll
  for (int k = 0; k < kLeanRows; ++k) {
    const int row = kLeanTW + ty + 4 * k;
    const int y_out = y0 + ty + 4 * k;
    const int cy = s_c[row];
    const f2 kyv = *reinterpret_cast<const f2*>(&s_k[row][0]);
    const int yl0 = (min(max(cy, 0), a.Hin - 1) - fy0) * kLeanLW, yl1 = (min(max(cy + 1, 0), a.Hin - 1) - fy0) * kLeanLW;
    const float w0 = pxl[0] * kyv[0], w1 = pxl[1] * kyv[0], w2 = pxl[0] * kyv[1], w3 = pxl[1] * kyv[1];
    const float wsum = (((0.f + w0) + w1) + w2) + w3;
    const unsigned boff = 4u * ((unsigned)y_out * (unsigned)a.Wout + (unsigned)x_out);       // < 2^31 * 4: checked by the launcher
    const bool live = live_x && y_out <= yh_out;
#pragma unroll
    for (int pl = 0; pl < kLeanPlanes; ++pl) {
      const float t0 = s_img[pl][yl0 + xl0], t1 = s_img[pl][yl0 + xl1], t2 = s_img[pl][yl1 + xl0], t3 = s_img[pl][yl1 + xl1];   // (already scaled)
      const float sum = fmaf(w3, t3, fmaf(w2, t2, fmaf(w1, t1, fmaf(w0, t0, 0.f))));
      const float v = (!wsum) ? 0.f : (sum / wsum);   // :93
      if (live && c_lo + pl < a.NC && !((verdict >> pl) & 1u)) {
        *reinterpret_cast<float*>(dstb[pl] + boff) = v;
        if (EXTRA && dst2b[pl]) *reinterpret_cast<float*>(dst2b[pl] + boff) = scaled(v, a.out2_scale);
      }
    }
  }
}

// LINEAR up-sampling by an exact integer factor F on both axes (the x4 flow up-sampling behind every FlowNet2 stage and the deploy
// head).  With fx = fy = 1 / F the F x F outputs of one INPUT pixel (i, j) share their tap window: x_in = j + (ph + 0.5) / F - 0.5
// rounds to j for every phase, so the reference's loop (:75-92) visits the same 5 x 5 input taps for all of them and only the
// coefficients differ.  One thread per input pixel: the 25 taps are read once; the 16 taps of the outer ring have coefficient 0 for
// every phase (triangle support 1), so they enter as ONE term  poison = sum 0 * tap  -- 0 for finite taps, NaN if the reference's
// 0 * NaN / 0 * Inf would have poisoned the sums (the sign of a zero sum aside, the result is the reference's) -- and each output is
// 9 fused multiply-adds in the reference's order plus the sum / wsum division (:93).  Stores are whole 16-byte (8-byte) rows.
template <int F, bool EXTRA>     // EXTRA: input scaling / channel-slice tops / second top (fn2_resample_forward_slices); false = the plain layer
__global__ void __launch_bounds__(256) resample_up_linear(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out2, ResampleArgs a) {
  const unsigned hw_out = (unsigned)a.Hout * a.Wout, hw_in = (unsigned)a.Hin * a.Win;
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= hw_in) return;
  const int i = p / a.Win, j = p - i * a.Win;
  // per-axis tap tables: columns j - 2 .. j + 2 (rows i - 2 .. i + 2), the in-image mask, and per phase the 3 inner coefficients
  unsigned xo[5], yo[5], mx = 0u, my = 0u;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int x = j - 2 + t, y = i - 2 + t;
    const bool okx = x >= 0 && x < a.Win, oky = y >= 0 && y < a.Hin;
    xo[t] = okx ? (unsigned)x : 0u;
    yo[t] = oky ? (unsigned)y * a.Win : 0u;
    mx |= (okx ? 1u : 0u) << t;
    my |= (oky ? 1u : 0u) << t;
  }
  float px[F][3], ky[F][3], wsum[F][F];
#pragma unroll
  for (int ph = 0; ph < F; ++ph) {
    const float x_in = (F * j + ph) * a.fx + a.fy / 2.0f - 0.5f;   // :62
    const float y_in = (F * i + ph) * a.fy + a.fx / 2.0f - 0.5f;   // :63
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int x = j - 1 + t, y = i - 1 + t;
      px[ph][t] = ((mx >> (t + 1)) & 1u) ? a.ax * triangle_coeff(a.ax * (x_in - x)) * a.ay : 0.f;
      ky[ph][t] = ((my >> (t + 1)) & 1u) ? triangle_coeff(a.ay * (y_in - y)) : 0.f;
    }
  }
#pragma unroll
  for (int py = 0; py < F; ++py)
#pragma unroll
    for (int qx = 0; qx < F; ++qx) {
      float w = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) w += px[qx][t] * ky[py][r];
      wsum[py][qx] = w;
    }
  using vec_t = __attribute__((ext_vector_type(F))) float;
  const int c_lo = blockIdx.y * a.ppt, c_hi = min(a.NC, c_lo + a.ppt);
  for (int c = c_lo; c < c_hi; ++c) {
    const float* src = in + (size_t)c * hw_in;
    float v[3][3], poison = 0.f;
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        float s = ((mx >> t) & (my >> r) & 1u) ? src[yo[r] + xo[t]] : 0.f;
        if constexpr (EXTRA) s = scaled(s, a.in_scale);
        if (r >= 1 && r <= 3 && t >= 1 && t <= 3) v[r - 1][t - 1] = s;
        else poison = fmaf(0.f, s, poison);
      }
    const size_t pix0 = (size_t)(F * i) * a.Wout + F * j;
    float* dst = out + (EXTRA ? top_plane(c, a.C, a.octot, a.oc0) : (size_t)c) * hw_out + pix0;
    float* dst2 = (EXTRA && out2) ? out2 + top_plane(c, a.C, a.o2ctot, a.o2c0) * hw_out + pix0 : nullptr;
#pragma unroll
    for (int py = 0; py < F; ++py) {
      vec_t o;
#pragma unroll
      for (int qx = 0; qx < F; ++qx) {
        float sum = poison;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int t = 0; t < 3; ++t) sum = fmaf(px[qx][t] * ky[py][r], v[r][t], sum);
        const float ws = wsum[py][qx];                            // :93 -- away from the border the coefficients (multiples of 1/64) sum to exactly 1: x / 1 == x
        o[qx] = (!ws) ? 0.f : (ws == 1.0f ? sum : sum / ws);
      }
      *reinterpret_cast<vec_t*>(dst + (size_t)py * a.Wout) = o;
      if (EXTRA && dst2) {
        vec_t o2;
#pragma unroll
        for (int qx = 0; qx < F; ++qx) o2[qx] = scaled(o[qx], a.out2_scale);
        *reinterpret_cast<vec_t*>(dst2 + (size_t)py * a.Wout) = o2;
      }
    }
  }
}

}  // namespace fn2

using namespace fn2;

static int g_resample_generic = 0;     // test hook: 1 = always the per-output-pixel kernels
FN2_API int fn2_debug_set_resample_generic(int on) { g_resample_generic = on; return FN2_OK; }

FN2_API int fn2_resample_forward_slices(const float* in, float in_scale, float* out, int top_channels, int top_c0,
                                        float* out2, int top2_channels, int top2_c0, float out2_scale,
                                        int N, int C, int Hin, int Win, int Hout, int Wout, int type, int antialias_param, void* stream) {
  if (N < 0 || C < 1 || Hin < 1 || Win < 1) return fail(FN2_ERR_INVALID_ARG, "resample: bad bottom shape");
  if (Hout < 1 || Wout < 1) return fail(FN2_ERR_INVALID_ARG, "ResampleLayer must have top_height > 0 and top_width > 0");
  if (type != FN2_RESAMPLE_NEAREST && type != FN2_RESAMPLE_LINEAR && type != FN2_RESAMPLE_CUBIC)
    return fail(FN2_ERR_UNSUPPORTED, "ResampleLayer: only CUBIC, LINEAR and NEAREST interpolation is supported for now");
  if (N == 0) return FN2_OK;
  if (!in || !out) return fail(FN2_ERR_INVALID_ARG, "resample: NULL blob pointer");
  if (top_c0 < 0 || top_c0 + C > top_channels || (out2 && (top2_c0 < 0 || top2_c0 + C > top2_channels)))
    return fail(FN2_ERR_INVALID_ARG, "resample: channel slice outside its blob");
  ResampleArgs a;
  a.C = C; a.octot = top_channels; a.oc0 = top_c0; a.o2ctot = out2 ? top2_channels : C; a.o2c0 = out2 ? top2_c0 : 0;
  a.in_scale = in_scale; a.out2_scale = out2_scale;
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
  if ((long long)Hout * Wout >= (1ll << 29) || (long long)Hin * Win >= (1ll << 29)) return fail(FN2_ERR_UNSUPPORTED, "resample: plane too large");      // 32-bit byte offsets inside a plane
  const unsigned bx = (unsigned)(((long long)Hout * Wout + 255) / 256);
  // planes per thread: amortise the per-pixel work, but keep >= ~2k workgroups in flight
  a.ppt = 1;
  while (a.ppt < 8 && (long long)bx * ((a.NC + 2 * a.ppt - 1) / (2 * a.ppt)) >= 2048) a.ppt *= 2;
  const unsigned by = (unsigned)((a.NC + a.ppt - 1) / a.ppt);
  if (by > 65535u) return fail(FN2_ERR_UNSUPPORTED, "resample: too many planes");
  const dim3 grid(bx, by);
  hipStream_t st = as_stream(stream);
  const bool fast = a.rx <= 2 && a.ry <= 2;
  // exact x2 / x4 LINEAR up-sampling: one thread per INPUT pixel (resample_up_linear)
  const int up = (Wout == 4 * Win && Hout == 4 * Hin) ? 4 : (Wout == 2 * Win && Hout == 2 * Hin) ? 2 : 0;
  if (type == FN2_RESAMPLE_LINEAR && up && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(out2)) & 15) == 0 && !g_resample_generic) {
    const unsigned bxi = (unsigned)(((long long)Hin * Win + 255) / 256);
    a.ppt = 1;
    while (a.ppt < 4 && (long long)bxi * ((a.NC + 2 * a.ppt - 1) / (2 * a.ppt)) >= 8192) a.ppt *= 2;     // >= 8k workgroups before planes share a thread: the kernel is latency-bound (25 loads, then 16 stores)
    const dim3 gi(bxi, (unsigned)((a.NC + a.ppt - 1) / a.ppt));
    if (gi.y > 65535u) return fail(FN2_ERR_UNSUPPORTED, "resample: too many planes");
    const bool extra = in_scale != 1.0f || out2 || top_channels != C;
    if (up == 4 && extra) hipLaunchKernelGGL((resample_up_linear<4, true>), gi, dim3(256), 0, st, in, out, out2, a);
    else if (up == 4) hipLaunchKernelGGL((resample_up_linear<4, false>), gi, dim3(256), 0, st, in, out, out2, a);
    else if (extra) hipLaunchKernelGGL((resample_up_linear<2, true>), gi, dim3(256), 0, st, in, out, out2, a);
    else hipLaunchKernelGGL((resample_up_linear<2, false>), gi, dim3(256), 0, st, in, out, out2, a);
    return check_launch("resample_forward");
  }
  const bool ex = in_scale != 1.0f || out2 || top_channels != C;
  if (type == FN2_RESAMPLE_LINEAR && fast && a.ax == 1.0f && a.ay == 1.0f && a.rx == 2 && a.ry == 2 && a.fx <= 1.0f && a.fy <= 1.0f && !g_resample_generic) {
    const unsigned gx = (unsigned)((Wout + kLeanTW - 1) / kLeanTW), gy = (unsigned)((Hout + kLeanTH - 1) / kLeanTH);
    const unsigned gz = (unsigned)((a.NC + kLeanPlanes - 1) / kLeanPlanes);
    if (gy > 65535u || gz > 65535u) return fail(FN2_ERR_UNSUPPORTED, "resample: too many tiles / planes");
    if (ex) hipLaunchKernelGGL((resample_linear_lean<true>), dim3(gx, gy, gz), dim3(256), 0, st, in, out, out2, a);
    else hipLaunchKernelGGL((resample_linear_lean<false>), dim3(gx, gy, gz), dim3(256), 0, st, in, out, out2, a);
    return check_launch("resample_forward");
  }
#define FN2_RS(K_) do { if (ex) hipLaunchKernelGGL((K_<true>), grid, dim3(256), 0, st, in, out, out2, a); \
                        else hipLaunchKernelGGL((K_<false>), grid, dim3(256), 0, st, in, out, out2, a); } while (0)
#define FN2_RSI(C_, F_) do { if (ex) hipLaunchKernelGGL((resample_interp<C_, F_, true>), grid, dim3(256), 0, st, in, out, out2, a); \
                             else hipLaunchKernelGGL((resample_interp<C_, F_, false>), grid, dim3(256), 0, st, in, out, out2, a); } while (0)
  if (type == FN2_RESAMPLE_NEAREST) FN2_RS(resample_nearest);
  else if (type == FN2_RESAMPLE_CUBIC) { if (fast) FN2_RSI(true, true); else FN2_RSI(true, false); }
  else { if (fast) FN2_RSI(false, true); else FN2_RSI(false, false); }
#undef FN2_RS
#undef FN2_RSI
  return check_launch("resample_forward");
}

FN2_API int fn2_resample_forward(const float* in, float* out, int N, int C, int Hin, int Win, int Hout, int Wout,
                                 int type, int antialias_param, void* stream) {
  return fn2_resample_forward_slices(in, 1.0f, out, C, 0, nullptr, 0, 0, 1.0f, N, C, Hin, Win, Hout, Wout, type, antialias_param, stream);
}
