// DataAugmentation: the image half of the training augmentation, for given coefficients.
//
// Reference: DataAugmentationLayer::Forward_gpu (src/caffe/layers/data_augmentation_layer.cu:320-637) runs SpatialAugmentation, then
// up to three in-place passes over the cropped batch (ChromaticEigenAugmentation, ColorContrastAugmentation, ApplyEffects) and the mean
// subtraction as N small GEMMs / AXPYs.  Every one of those is a per-pixel function of the three channel values, so one kernel does
// the whole chain: per output pixel C bilinear samples (4 reads each, neighbouring lanes read neighbouring addresses for the mild
// transforms the generator draws) and C coalesced writes -- HBM-bound, bytes = 4 * N * C * (H*W read at most once + crop_h*crop_w).
// The chromatic-eigen transform needs batch statistics of the SOURCE images first (ComputeChromaticEigenspace, :147-187): a reduction
// kernel (wave shuffles -> LDS -> one atomic per block; 256 blocks: the atomics on the 12 statistics serialise) in front.
#include "augmentation.hpp"
#include "philox.hpp"

#include <cfloat>
#include <cstring>

namespace fn2 {

// tChromaticCoeffs / tChromaticEigenCoeffs / tEffectCoeffs, include/caffe/layers/augmentation_layer_base.hpp:37-113
struct ItemCoeffs {
  TransMat m;
  float gamma, brightness, contrast, color[3];
  float pow_nomean[3], add_nomean[3], mult_nomean[3];
  float pow_withmean0, add_withmean0, mult_withmean0, pow_withmean1, add_withmean1, mult_withmean1;
  float lmult_pow, lmult_add, lmult_mult, col_angle;
  float shadow_nx, shadow_ny, shadow_distance, shadow_strength;
  float noise;                         // tEffectCoeffs::noise: sigma of the additive Gaussian noise of this sample (:578-587)
  int chromatic, eigen, effect;        // needsComputation() of the three groups (per sample; the kernels run on the whole batch, see below)
};

// tChromaticEigenSpace, augmentation_layer_base.hpp:115-127
struct EigenSpace {
  float mean_eig[3], mean_rgb[3], max_abs_eig[3], max_rgb[3], min_rgb[3], max_l, eigvec[9];
};

constexpr int kItemChunk = 16;
struct DataAugArgs {
  const float* bottom;
  const float* mean;
  float* top;
  const EigenSpace* eigen;
  int n0, n_chunk, C, H, W, ch, cw, mean_mode, spatial;
  long long src_count;
  float max_multiplier;
  unsigned seed_lo, seed_hi, stream_lo, stream_hi;     // Philox key (seed) and the high counter words (stream = iteration)
  ItemCoeffs item[kItemChunk];
};

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }   // :20-22

// ComputeChromaticEigenspace, :147-187: per pixel the projections on the eigenvectors; batch max |eig|, max / min rgb, and
// sum of rgb / width / height (divided by num on the host, :517-518).
struct EigVec { float v[9]; };

// grid: (pixel blocks, sample): 32-bit pixel indices, the eigenvectors in kernel arguments (round 6: the flat 64-bit index cost a 64-bit
// division per pixel and the nine coefficients were re-read from memory -- 32 us for a batch of 8 at 512x384, most of it integer division)
__global__ void __launch_bounds__(256) eigenspace_stats(const float* __restrict__ data, int N, int H, int W, float* __restrict__ partial, EigVec ev, int vec4) {
  const unsigned hw = (unsigned)H * (unsigned)W;
  const float* img = data + (size_t)blockIdx.y * 3 * hw;
  float sum[3] = {0, 0, 0}, mx_eig[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  auto pixel = [&](const float (&rgb)[3]) {
    for (int c = 0; c < 3; ++c) {
      const float eig = ev.v[3 * c] * rgb[0] + ev.v[3 * c + 1] * rgb[1] + ev.v[3 * c + 2] * rgb[2];
      mx_eig[c] = fmaxf(mx_eig[c], fabsf(eig));
      mx[c] = fmaxf(mx[c], rgb[c]);
      mn[c] = fminf(mn[c], rgb[c]);
      sum[c] += rgb[c] / W / H;                                                                        // :176
    }
  };
  if (vec4) {       // four pixels per thread and step, 16-byte loads (the scalar loop was a chain of memory round trips: 72 per thread)
    using f4 = __attribute__((ext_vector_type(4))) float;
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < hw / 4; q += gridDim.x * blockDim.x) {
      f4 v[3];
      for (int c = 0; c < 3; ++c) v[c] = reinterpret_cast<const f4*>(img + (size_t)c * hw)[q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float rgb[3] = {v[0][j], v[1][j], v[2][j]};
        pixel(rgb);
      }
    }
  } else {
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
      float rgb[3];
      for (int c = 0; c < 3; ++c) rgb[c] = img[(size_t)c * hw + p];
      pixel(rgb);
    }
  }
  __shared__ float sh[4][12];
  for (int c = 0; c < 3; ++c) {
    for (int off = 32; off > 0; off >>= 1) {
      sum[c] += __shfl_down(sum[c], off);
      mx_eig[c] = fmaxf(mx_eig[c], __shfl_down(mx_eig[c], off));
      mx[c] = fmaxf(mx[c], __shfl_down(mx[c], off));
      mn[c] = fminf(mn[c], __shfl_down(mn[c], off));
    }
  }
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  if (lane == 0)
    for (int c = 0; c < 3; ++c) { sh[wave][c] = sum[c]; sh[wave][3 + c] = mx_eig[c]; sh[wave][6 + c] = mx[c]; sh[wave][9 + c] = mn[c]; }
  __syncthreads();
  // one row of 12 partial results per workgroup (round 6): the 12 atomics per workgroup on the SAME 12 words were the kernel -- 256 workgroups
  // x 12 contended atomics = most of its 32 us, 1,024 workgroups 114 us -- and made the mean a sum in arrival order; eigenspace_finish
  // now reduces the rows in a fixed order
  if (threadIdx.x < 12) {
    const int j = threadIdx.x;
    float r = sh[0][j];
    for (int w = 1; w < 4; ++w) r = j < 3 ? r + sh[w][j] : j < 9 ? fmaxf(r, sh[w][j]) : fminf(r, sh[w][j]);
    partial[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 12 + j] = r;
  }
}

__global__ void eigenspace_init(EigenSpace* es, EigVec ev) {                                              // :491-500
  for (int c = 0; c < 3; ++c) { es->mean_eig[c] = 0; es->mean_rgb[c] = 0; es->max_abs_eig[c] = 0; es->max_rgb[c] = 0; es->min_rgb[c] = FLT_MAX; }
  es->max_l = 0;
  for (int i = 0; i < 9; ++i) es->eigvec[i] = ev.v[i];
}

// one workgroup: thread t folds rows t, t + 256, ... in order, then a fixed tree over the threads; thread 0 finishes the statistics
__global__ void __launch_bounds__(256) eigenspace_finish(EigenSpace* es, const float* __restrict__ partial, int rows, int num) {     // :517-534 (host code in the reference)
  __shared__ float red[256][12];
  const int t = threadIdx.x;
  float r[12];
  for (int j = 0; j < 12; ++j) r[j] = j < 3 ? 0.f : j < 9 ? 0.f : FLT_MAX;
  for (int row = t; row < rows; row += 256)
    for (int j = 0; j < 12; ++j) {
      const float v = partial[(size_t)row * 12 + j];
      r[j] = j < 3 ? r[j] + v : j < 9 ? fmaxf(r[j], v) : fminf(r[j], v);
    }
  for (int j = 0; j < 12; ++j) red[t][j] = r[j];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s)
      for (int j = 0; j < 12; ++j) red[t][j] = j < 3 ? red[t][j] + red[t + s][j] : j < 9 ? fmaxf(red[t][j], red[t + s][j]) : fminf(red[t][j], red[t + s][j]);
    __syncthreads();
  }
  if (t != 0) return;
  for (int c = 0; c < 3; ++c) { es->mean_rgb[c] = red[0][c]; es->max_abs_eig[c] = red[0][3 + c]; es->max_rgb[c] = red[0][6 + c]; es->min_rgb[c] = red[0][9 + c]; }
  for (int c = 0; c < 3; ++c) es->mean_rgb[c] = es->mean_rgb[c] / num;
  for (int c = 0; c < 3; ++c) {
    es->mean_eig[c] = es->eigvec[3 * c] * es->mean_rgb[0] + es->eigvec[3 * c + 1] * es->mean_rgb[1] + es->eigvec[3 * c + 2] * es->mean_rgb[2];
    if (es->max_abs_eig[c] > 1e-2f) es->mean_eig[c] = es->mean_eig[c] / es->max_abs_eig[c];
  }
  es->max_l = sqrtf(es->max_abs_eig[0] * es->max_abs_eig[0] + es->max_abs_eig[1] * es->max_abs_eig[1] + es->max_abs_eig[2] * es->max_abs_eig[2]);
}

// ChromaticEigenAugmentation, :192-291, on one pixel
__device__ __forceinline__ void eigen_pixel(float* rgb_io, const ItemCoeffs& k, const EigenSpace& e, float max_multiplier) {
  float rgb[3], eig[3];
  for (int c = 0; c < 3; ++c) rgb[c] = rgb_io[c] - e.mean_rgb[c];                                         // :209
  for (int c = 0; c < 3; ++c) {
    eig[c] = e.eigvec[3 * c] * rgb[0] + e.eigvec[3 * c + 1] * rgb[1] + e.eigvec[3 * c + 2] * rgb[2];     // :214
    if (e.max_abs_eig[c] > 1e-2f) {
      eig[c] = eig[c] / e.max_abs_eig[c];
      eig[c] = copysignf(powf(fabsf(eig[c]), k.pow_nomean[c]), eig[c]);                                   // :218-231
      eig[c] = eig[c] + k.add_nomean[c];
      eig[c] = eig[c] * k.mult_nomean[c];
    }
  }
  for (int c = 0; c < 3; ++c) eig[c] = eig[c] + e.mean_eig[c];                                            // :236-237
  if (e.max_abs_eig[0] > 1e-2f) {                                                                         // :240-244
    eig[0] = copysignf(powf(fabsf(eig[0]), k.pow_withmean0), eig[0]);
    eig[0] = eig[0] + k.add_withmean0;
    eig[0] = eig[0] * k.mult_withmean0;
  }
  const float s = sqrtf(eig[1] * eig[1] + eig[2] * eig[2]);                                               // :245
  float s1 = s, l = 0.f, l1 = 0.f;
  if (s > 1e-2f) {                                                                                        // :247-251
    s1 = powf(s1, k.pow_withmean1);
    s1 = fmaxf(s1 + k.add_withmean1, 0.f);
    s1 = s1 * k.mult_withmean1;
  }
  if (k.col_angle != 0) {                                                                                 // :252-259
    const float t1 = cosf(k.col_angle) * eig[1] - sinf(k.col_angle) * eig[2];
    const float t2 = sinf(k.col_angle) * eig[1] + cosf(k.col_angle) * eig[2];
    eig[1] = t1;
    eig[2] = t2;
  }
  for (int c = 0; c < 3; ++c)
    if (e.max_abs_eig[c] > 1e-2f) eig[c] = eig[c] * e.max_abs_eig[c];                                     // :260-263
  if (e.max_l > 1e-2f) {                                                                                  // :264-267
    l1 = sqrtf(eig[0] * eig[0] + eig[1] * eig[1] + eig[2] * eig[2]);
    l1 = l1 / e.max_l;
  }
  if (s > 1e-2f) {                                                                                        // :268-271
    eig[1] = eig[1] / s * s1;
    eig[2] = eig[2] / s * s1;
  }
  if (e.max_l > 1e-2f) {                                                                                  // :272-284
    l = sqrtf(eig[0] * eig[0] + eig[1] * eig[1] + eig[2] * eig[2]);
    l1 = powf(l1, k.lmult_pow);
    l1 = fmaxf(l1 + k.lmult_add, 0.f);
    l1 = l1 * k.lmult_mult;
    l1 = l1 * e.max_l;
    if (l > 1e-2f)
      for (int c = 0; c < 3; ++c) {
        eig[c] = eig[c] / l * l1;
        if (eig[c] > e.max_abs_eig[c]) eig[c] = e.max_abs_eig[c];
      }
  }
  for (int c = 0; c < 3; ++c) {                                                                           // :285-290
    float v = e.eigvec[c] * eig[0] + e.eigvec[3 + c] * eig[1] + e.eigvec[6 + c] * eig[2];
    v = v < max_multiplier ? v : max_multiplier;
    v = v > 0 ? v : 0;
    rgb_io[c] = v;
  }
}

// ColorContrastAugmentation, :72-116, on one pixel
__device__ __forceinline__ void chromatic_pixel(float* rgb, const ItemCoeffs& k, float max_multiplier) {
  float mean_in = 0, mean_out = 0;
  for (int c = 0; c < 3; ++c) {
    mean_in += rgb[c];
    rgb[c] *= k.color[c];
    mean_out += rgb[c];
  }
  const float brightness_coeff = mean_in / (mean_out + 0.01f);                                            // :97
  for (int c = 0; c < 3; ++c) {
    float v = clampf(rgb[c] * brightness_coeff, 0.f, 1.f);                                                // :101
    v = powf(v, k.gamma);                                                                                 // :104
    v = v + k.brightness;                                                                                 // :107
    v = 0.5f + (v - 0.5f) * k.contrast;                                                                   // :110
    rgb[c] = clampf(v, 0.f, max_multiplier);                                                              // :113
  }
}

template <bool NOISE>      // the noise effect has its own instantiation: the Philox rounds and the Box-Muller transform cost the plain path 26 % (registers)
__global__ void __launch_bounds__(256) data_aug_kernel(DataAugArgs a) {
  const long long per = (long long)a.ch * a.cw, total = per * a.n_chunk;
  for (long long index = blockIdx.x * (long long)blockDim.x + threadIdx.x; index < total; index += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(index % a.cw), y = (int)((index / a.cw) % a.ch), kk = (int)(index / per);
    const int n = a.n0 + kk;
    const ItemCoeffs& k = a.item[kk];
    float w00 = 1.f, w11 = 0.f, w01 = 0.f, w10 = 0.f;
    long long off = (long long)y * a.W + x;            // no cropping: the bottom is copied (:590)
    if (a.spatial) {                                   // SpatialAugmentation, :41-62
      float xpos = x * k.m.t0 + y * k.m.t2 + k.m.t4;
      float ypos = x * k.m.t1 + y * k.m.t3 + k.m.t5;
      xpos = clampf(xpos, 0.0f, (float)(a.W) - 1.05f);
      ypos = clampf(ypos, 0.0f, (float)(a.H) - 1.05f);
      const float tlx = floorf(xpos), tly = floorf(ypos);
      const float xdist = xpos - tlx, ydist = ypos - tly;
      off = (long long)tly * a.W + (long long)tlx;
      w00 = (1 - xdist) * (1 - ydist); w11 = xdist * ydist; w01 = (1 - xdist) * ydist; w10 = xdist * (1 - ydist);
    }
    float px[4];                                       // C <= 3 goes through the colour path; other channel counts are written directly
    for (int c0 = 0; c0 < a.C; c0 += 3) {
      const int nc = a.C - c0 < 3 ? a.C - c0 : 3;
      for (int c = 0; c < nc; ++c) {
        const long long base = ((long long)n * a.C + c0 + c) * a.H * a.W + off;
        if (a.spatial) {
          const long long last = a.src_count - 1;      // the reference clamps the three neighbours to src_count (one past the end), :53-56
          const float tl = a.bottom[base];
          const float tr = a.bottom[base + 1 <= last ? base + 1 : last];
          const float bl = a.bottom[base + a.W <= last ? base + a.W : last];
          const float br = a.bottom[base + 1 + a.W <= last ? base + 1 + a.W : last];
          px[c] = w00 * tl + w11 * br + w01 * bl + w10 * tr;                                              // :61-64
        } else {
          px[c] = a.bottom[base];
        }
      }
      if (a.C == 3) {
        if (k.eigen) eigen_pixel(px, k, *a.eigen, a.max_multiplier);
        if (k.chromatic) chromatic_pixel(px, k, a.max_multiplier);
      }
      // the noise effect (:578-587: caffe_gpu_rng_gaussian(count, 0, noise) added to the sample after ApplyEffects): i.i.d. N(0, noise^2)
      // per element from Philox4x32-10 -- counter (pixel, sample * 4 + channel triple, stream), key = seed -- and Box-Muller
      float z[3] = {0.f, 0.f, 0.f};
      if (NOISE && k.noise > 0.f) {
        const long long pixn = (long long)y * a.cw + x;
        const Philox4 r = philox4x32_10((unsigned)pixn, (unsigned)(pixn >> 32) ^ ((unsigned)n * 4u + (unsigned)(c0 / 3)), a.stream_lo, a.stream_hi, a.seed_lo, a.seed_hi);
        const float r0 = sqrtf(-2.0f * logf(philox_unit(r.v[0]))), t0 = 6.283185307179586f * philox_unit(r.v[1]);
        const float r1 = sqrtf(-2.0f * logf(philox_unit(r.v[2]))), t1 = 6.283185307179586f * philox_unit(r.v[3]);
        z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0); z[2] = r1 * cosf(t1);
      }
      for (int c = 0; c < nc; ++c) {
        float v = px[c];
        if (k.effect) {                                                                                   // ApplyEffects, :308-315
          if ((x - a.cw / 2) * k.shadow_nx + (y - a.ch / 2) * k.shadow_ny - k.shadow_distance > 0) v -= k.shadow_strength;
          v = clampf(v, 0.f, a.max_multiplier);
        }
        if (NOISE && k.noise > 0.f) v = v + k.noise * z[c];
        const long long pix = (long long)y * a.cw + x;
        if (a.mean_mode == FN2_MEAN_PER_CHANNEL) v = v - a.mean[c0 + c];                                  // :620-634
        else if (a.mean_mode == FN2_MEAN_PER_PIXEL) v = v - a.mean[(long long)(c0 + c) * per + pix];      // :613-616
        a.top[((long long)n * a.C + c0 + c) * per + pix] = v;
      }
    }
  }
}

}  // namespace fn2

using namespace fn2;

constexpr int kStatRowsMax = 1024;        // workgroups of the statistics pass (rows of 12 partial results behind the EigenSpace record)
FN2_API size_t fn2_data_augmentation_workspace_bytes(int) { return 256 + sizeof(float) * 12 * kStatRowsMax; }

FN2_API int fn2_data_augmentation_forward(const fn2_data_aug_params* p, const float* bottom, const float* coeffs, const float* mean,
                                          float* top, int N, int C, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
  if (!p) return fail(FN2_ERR_INVALID_ARG, "data_augmentation: params == NULL");
  if (N < 0 || C < 1 || H < 1 || W < 1) return fail(FN2_ERR_INVALID_ARG, "data_augmentation: bad bottom shape [%d,%d,%d,%d]", N, C, H, W);
  const bool do_cropping = p->crop_width > 0 && p->crop_height > 0;                                       // cpp:94
  const int cw = do_cropping ? p->crop_width : W, ch = do_cropping ? p->crop_height : H;
  if (W < cw) return fail(FN2_ERR_INVALID_ARG, "crop width greater than original");                       // cpp:103
  if (H < ch) return fail(FN2_ERR_INVALID_ARG, "crop height greater than original");                      // cpp:104
  if (p->mean_mode < FN2_MEAN_NONE || p->mean_mode > FN2_MEAN_PER_PIXEL) return fail(FN2_ERR_INVALID_ARG, "data_augmentation: unknown mean_mode %d", p->mean_mode);
  if (N == 0) return FN2_OK;
  if (!bottom || !top || (p->mean_mode != FN2_MEAN_NONE && !mean)) return fail(FN2_ERR_INVALID_ARG, "data_augmentation: NULL pointer");
  hipStream_t st = as_stream(stream);

  DataAugArgs a;
  std::memset(&a, 0, sizeof(a));
  a.bottom = bottom; a.mean = mean; a.top = top; a.C = C; a.H = H; a.W = W; a.ch = ch; a.cw = cw;
  a.mean_mode = p->mean_mode; a.spatial = do_cropping; a.src_count = (long long)N * C * H * W; a.max_multiplier = p->max_multiplier;
  a.eigen = static_cast<const EigenSpace*>(workspace);
  a.seed_lo = (unsigned)p->noise_seed; a.seed_hi = (unsigned)(p->noise_seed >> 32);
  a.stream_lo = (unsigned)p->noise_stream; a.stream_hi = (unsigned)(p->noise_stream >> 32);

  // pass 1 over the coefficients (:452-476): the reference launches each colour / effect kernel over the WHOLE batch as soon as ONE
  // sample needs it (has_chromatic_augmentation etc. are batch flags); samples with default coefficients go through it too, and it is
  // not the identity for them (brightness compensation mean_in / (mean_out + 0.01), clamps to [0,1] and [0,max_multiplier]).
  bool any_eigen = false, any_chromatic = false, any_effect = false;
  static const float zeros[A_COUNT] = {0};
  auto item_coeffs = [&](int n, ItemCoeffs* k) -> int {
    AugCoeff c;
    c.from_array(coeffs ? coeffs + (size_t)n * A_COUNT : zeros);
    c.clear_defaults();
    std::memset(k, 0, sizeof(*k));
    k->m.identity();
    k->m.from_coeff(c, cw, ch, W, H);
    k->gamma = c.v[A_GAMMA]; k->brightness = c.v[A_BRIGHTNESS]; k->contrast = c.v[A_CONTRAST];
    k->color[0] = c.v[A_COLOR1]; k->color[1] = c.v[A_COLOR2]; k->color[2] = c.v[A_COLOR3];
    k->chromatic = k->gamma != 1 || k->brightness != 0 || k->contrast != 1 || k->color[0] != 1 || k->color[1] != 1 || k->color[2] != 1;
    for (int i = 0; i < 3; ++i) {
      k->pow_nomean[i] = c.v[A_POW_NOMEAN0 + i]; k->add_nomean[i] = c.v[A_ADD_NOMEAN0 + i]; k->mult_nomean[i] = c.v[A_MULT_NOMEAN0 + i];
    }
    k->pow_withmean0 = c.v[A_POW_WITHMEAN0]; k->add_withmean0 = c.v[A_ADD_WITHMEAN0]; k->mult_withmean0 = c.v[A_MULT_WITHMEAN0];
    k->pow_withmean1 = c.v[A_POW_WITHMEAN1]; k->add_withmean1 = c.v[A_ADD_WITHMEAN1]; k->mult_withmean1 = c.v[A_MULT_WITHMEAN1];
    k->lmult_pow = c.v[A_LMULT_POW]; k->lmult_add = c.v[A_LMULT_ADD]; k->lmult_mult = c.v[A_LMULT_MULT]; k->col_angle = c.v[A_COL_ANGLE];
    k->eigen = false;
    for (int f = A_POW_NOMEAN0; f <= A_COL_ANGLE; ++f) k->eigen = k->eigen || c.v[f] != kAugDefault[f];   // hpp:86-94 (incl. the unused *_withmean2)
    k->shadow_nx = (float)std::cos((double)c.v[A_SHADOW_ANGLE]); k->shadow_ny = (float)std::sin((double)c.v[A_SHADOW_ANGLE]);   // hpp:110
    k->shadow_distance = c.v[A_SHADOW_DISTANCE]; k->shadow_strength = c.v[A_SHADOW_STRENGTH];
    k->noise = c.v[A_NOISE] > 0 ? c.v[A_NOISE] : 0.f;
    k->effect = (c.v[A_FOG_AMOUNT] != 0 && c.v[A_FOG_SIZE] != 0) || c.v[A_MOTION_BLUR_SIZE] > 0 || k->shadow_strength > 0 || k->noise > 0;   // hpp:111
    return FN2_OK;
  };
  if (do_cropping) {
    ItemCoeffs tmp;
    for (int n = 0; n < N; ++n) {
      int rc = item_coeffs(n, &tmp);
      if (rc) return rc;
      any_eigen = any_eigen || tmp.eigen;
      any_chromatic = any_chromatic || tmp.chromatic;
      any_effect = any_effect || tmp.effect;
    }
    if (any_eigen && C != 3) return fail(FN2_ERR_INVALID_ARG, "Chromatic-Eigen augmentations only work with 3-channel input");   // :489
    if (any_chromatic && C != 3) return fail(FN2_ERR_INVALID_ARG, "Chromatic augmentations only work with 3-channel input");      // :548
    if (any_effect && C != 3) return fail(FN2_ERR_INVALID_ARG, "Effect augmentations only work with 3-channel input");            // :556
  }
  if (any_eigen) {
    if (!p->has_chromatic_eigvec) return fail(FN2_ERR_INVALID_ARG, "You need to specify chromatic eigenvectors for Chromatic-Eigen augementation");   // :494
    if (!workspace || workspace_bytes < fn2_data_augmentation_workspace_bytes(N))
      return fail(FN2_ERR_WORKSPACE, "data_augmentation: workspace of %zu bytes needed", fn2_data_augmentation_workspace_bytes(N));
    if (N > kStatRowsMax) return fail(FN2_ERR_UNSUPPORTED, "data_augmentation: more than %d samples per call", kStatRowsMax);
    // :488-536 without the reference's two device <-> host round trips: initialise, reduce and finish on the device
    EigVec ev;
    for (int i = 0; i < 9; ++i) ev.v[i] = p->chromatic_eigvec[i];                                         // :496-497
    EigenSpace* es = static_cast<EigenSpace*>(workspace);
    hipLaunchKernelGGL(eigenspace_init, dim3(1), dim3(1), 0, st, es, ev);
    if ((long long)H * W >= (1ll << 31) || N > 65535) return fail(FN2_ERR_UNSUPPORTED, "data_augmentation: blob too large for the statistics pass");
    const int vec4 = ((long long)H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(bottom) & 15) == 0;
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    const unsigned per_sample = (unsigned)(kStatRowsMax / N > 0 ? kStatRowsMax / N : 1);
    const unsigned bx = blocks_for((long long)H * W / (vec4 ? 4 : 1), 256, per_sample);
    hipLaunchKernelGGL(eigenspace_stats, dim3(bx, (unsigned)N), dim3(256), 0, st, bottom, N, H, W, partial, ev, vec4);
    hipLaunchKernelGGL(eigenspace_finish, dim3(1), dim3(256), 0, st, es, partial, (int)(bx * (unsigned)N), N);
  }
  for (int n0 = 0; n0 < N; n0 += kItemChunk) {
    a.n0 = n0;
    a.n_chunk = N - n0 < kItemChunk ? N - n0 : kItemChunk;
    for (int k = 0; k < a.n_chunk; ++k) {
      if (do_cropping) {
        int rc = item_coeffs(n0 + k, &a.item[k]);
        if (rc) return rc;
        a.item[k].eigen = any_eigen; a.item[k].chromatic = any_chromatic; a.item[k].effect = any_effect;
      }
      else { std::memset(&a.item[k], 0, sizeof(ItemCoeffs)); a.item[k].m.identity(); }
    }
    bool noisy = false;
    for (int k = 0; k < a.n_chunk; ++k) noisy = noisy || a.item[k].noise > 0.f;
    if (noisy) hipLaunchKernelGGL((data_aug_kernel<true>), dim3(blocks_for((long long)a.n_chunk * ch * cw, 256)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((data_aug_kernel<false>), dim3(blocks_for((long long)a.n_chunk * ch * cw, 256)), dim3(256), 0, st, a);
  }
  return check_launch("data_augmentation_forward");
}
