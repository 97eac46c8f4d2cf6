// Transposed convolution with stride 2 on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chains), NCHW in and out:
//
//     out[n][co][Y][X] = act( bias[co] + sum_{ci, ky, kx : Y = 2 y - pad + ky, X = 2 x - pad + kx}  in[n][ci][y][x] * W[ci][co][ky][kx] )
//
// which is (a) the FORWARD of Deconvolution{kernel 4, stride 2, pad 1} -- DeconvolutionLayer::Forward_gpu, src/caffe/layers/
// deconv_layer.cu:8-26 (per sample backward_gpu_gemm = weight^T x bottom + col2im_gpu, base_conv_layer.cpp:375-393, then
// forward_gpu_bias), weight blob [Cin, Cout, 4, 4], followed by the in-place ReLU (relu_layer.cu:8-27) -- and (b) the DATA GRADIENT of
// a stride-2 Convolution -- ConvolutionLayer::Backward_gpu -> backward_gpu_gemm (conv_layer.cu:53-57, base_conv_layer.cpp:352-366:
// weight^T x top_diff + col2im): in = top_diff, W = the layer's weight [Cout_conv, Cin_conv, k, k] read as [in][out][k][k], out =
// bottom_diff; classes kernel / pad = 5 / 2 (conv2, conv3), 3 / 1 (conv4, conv5, conv6), 4 / 1.  No column matrix, no col2im.
//
// An output pixel of parity class (py, px) = (Y & 1, X & 1) at class position (i, j) = (Y >> 1, X >> 1) receives exactly the taps
// ky == (py + pad) (mod 2) from input row  i + ((py + pad) >> 1) - t,  t = 0, 1, ..  (kx / columns alike): the four classes are four
// stride-1 convolutions of the SAME input window with disjoint subsets of the taps.  GEMM view as in csrc/conv_mfma.hip -- M = a 4x4
// patch of class positions, N = 16 output channels, K = (channel quad, ky, kx), k = the 4 channels of the quad -- with one accumulator
// tile per (class, channel group, patch): the pixel operand of tap (ky, kx) is one ds_read_b32 at lane base + immediate from the staged
// window (natural [channel][row][column] order, 16-byte LDS-DMA, zero padding = out-of-range lanes), the weight operand streams global
// -> VGPR from the packed order of fn2_conv_mfma_pack_weights (of the [out][in][k][k] view of the blob), and every MFMA of a tap lands
// in the accumulators of that tap's class.  Epilogue: a lane holds 4 consecutive class columns of both x parities of a row: 8
// consecutive output pixels, bias + ReLU, two 16-byte stores; the top blob may be a channel slice of a Concat blob.
// Summation order per output element (restated by the oracle twin fn2_tconv_forward_cpu): channel quads ascending, within a quad the
// taps (ky, kx) of the element's class ascending, within a tap the 4 channels -- the same for every tile variant (same bits).
#include "fn2_common.hpp"
#include "autotune.hpp"

namespace fn2 {
namespace tc {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int up_mod(int v, int r, int m) { return v + ((r - v % m) + m) % m; }

struct Args {
  const float* in; const float* wp; const float* bias; float* out;
  int N, Cin, Hin, Win, in_ctot, in_c0;
  int Cout, Hout, Wout, out_ctot, out_c0;
  int nchunks, ksteps;
  int tx, ty, ng;
  unsigned total;
  float slope; int relu;
  // round 6, data-gradient use: the output is the gradient w.r.t. the ACTIVATED output `mask` of the layer in front ([N, mask_ctot, Hout, Wout],
  // channels from mask_c0 on): multiplied by that layer's leaky-ReLU derivative on the way out (ReLUBackward, relu_layer.cu:33-43, folded
  // into the kernel that produces its top_diff: one pass over the blob instead of three)
  const float* mask; int mask_ctot, mask_c0; float mask_slope;
};

// parity class algebra of one axis: taps k == (p + PAD) (mod 2); tap k reads input position  class position + D(p, k)
template <int KS, int PAD> struct Par {
  static constexpr int par_of(int k) { return (k + PAD) & 1; }                          // the class a tap belongs to: k == p + PAD (mod 2)
  static constexpr int d_of(int k) { return ((par_of(k) + PAD) >> 1) - (k - ((par_of(k) + PAD) & 1)) / 2; }
  static constexpr int dmin() { int m = 99; for (int k = 0; k < KS; ++k) m = d_of(k) < m ? d_of(k) : m; return m; }
  static constexpr int dmax() { int m = -99; for (int k = 0; k < KS; ++k) m = d_of(k) > m ? d_of(k) : m; return m; }
};

template <int KS_, int PAD_, int MW_, int NP_, int WM_, int WNX_, int WNY_, int CQ_>
struct Cfg {
  static constexpr int KS = KS_, PAD = PAD_, MW = MW_, NP = NP_, WM = WM_, WNX = WNX_, WNY = WNY_, CQ = CQ_;
  using P = Par<KS, PAD>;
  static constexpr int NW = WM * WNX * WNY, THREADS = 64 * NW;
  static constexpr int DMIN = P::dmin(), DMAX = P::dmax(), ND = DMAX - DMIN + 1;
  static constexpr int PADL = 4;                                     // window column 0 <-> input column j0 - PADL (16-byte aligned)
  static constexpr int TW = 4 * NP * WNX, TH = 4 * WNY;              // class positions of a workgroup tile (2 TW x 2 TH output pixels)
  static constexpr int WR = TH + ND - 1;                             // window rows: input rows i0 + DMIN .. i0 + TH - 1 + DMAX
  static constexpr int WC = TW + PADL + DMAX;                        // window columns: input columns j0 - PADL .. j0 + TW - 1 + DMAX
  static constexpr int RS = up_mod(cdiv(WC, 4) * 4, 4, 16);
  static constexpr int CS = up_mod(WR * RS, 16, 32);
  static constexpr int SLOTS_C = CS / 4;
  static constexpr int SLOTS = 4 * CQ * SLOTS_C;
  static constexpr int NRUN = cdiv(SLOTS, 64);
  static constexpr int RPW = cdiv(NRUN, NW);
  static constexpr int BUF = NRUN * 256;
  static constexpr int KSC = CQ * KS * KS;
  static constexpr int NBUFA = (KSC % 6 == 0) ? 6 : (KSC % 5 == 0) ? 5 : (KSC % 7 == 0) ? 7 : (KSC % 4 == 0) ? 4 : 3;
  static_assert(KSC % NBUFA == 0, "ring phase must repeat per chunk");
  static_assert(-DMIN <= PADL, "left margin");
  static_assert(2 * BUF * 4 <= 80 * 1024, "LDS (two workgroups per CU)");
  static_assert(NW == 4, "256 threads");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MW> struct WVec;
template <> struct WVec<1> { using T = float; };
template <> struct WVec<2> { using T = f32x2; };
template <> struct WVec<4> { using T = f32x4; };
template <int MW>
__device__ __forceinline__ float wget(const typename WVec<MW>::T& v, int j) {
  if constexpr (MW == 1) return v; else return v[j];
}

template <class K>
__device__ __forceinline__ void stage_chunk(__amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[K::RPW], unsigned dst, int wave, unsigned soff) {
#pragma unroll
  for (int i = 0; i < K::RPW; ++i) {
    const int r = i * K::NW + wave;
    if (r < K::NRUN)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)r), 16, voff[i], soff, 0, 0);
  }
}

template <class K>
__device__ __forceinline__ void tconv_body(const Args& a, int g, int bx, int by, int n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KS = K::KS, MW = K::MW, NP = K::NP;
  using P = typename K::P;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % K::WM, wnx = (wave / K::WM) % K::WNX, wny = wave / (K::WM * K::WNX);
  const int j0 = bx * K::TW, i0 = by * K::TH;                       // class position of the tile

  // ---- LDS-DMA plan: slot s = 64 (i NW + wave) + lane -> (channel of the chunk, window row, group of 4 window columns)
  const size_t plane = (size_t)a.Hin * a.Win;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + ((size_t)n * a.in_ctot + a.in_c0) * plane), 0, (unsigned)(4u * a.Cin * plane), 0x00020000);
  constexpr unsigned OOB = 0x7ffffff0u;
  unsigned voff[K::RPW];
#pragma unroll
  for (int i = 0; i < K::RPW; ++i) {
    const int s = (i * K::NW + wave) * 64 + lane;
    voff[i] = OOB;
    if (s < K::SLOTS) {
      const int c = s / K::SLOTS_C, rem = s % K::SLOTS_C;
      const int row = rem / (K::RS / 4), gq = rem % (K::RS / 4);
      const int yi = i0 + K::DMIN + row, xi = j0 - K::PADL + 4 * gq;
      if (row < K::WR && 4 * gq < K::WC && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win)
        voff[i] = 4u * (unsigned)(c * plane + (size_t)yi * a.Win + xi);
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const unsigned chunk_bytes = 4u * 4u * K::CQ * (unsigned)plane;
  auto stage = [&](int chunk, int buf) { stage_chunk<K>(rs, voff, lds_base + 4u * (unsigned)(buf * K::BUF), wave, (unsigned)chunk * chunk_bytes); };

  // ---- operands: lane (pixel p16 = lane & 15 -> (pi, pj) of the 4x4 patch, kq = lane >> 4)
  const int kq = lane >> 4, p16 = lane & 15, pi = p16 >> 2, pj = p16 & 3;
  const int bbase = kq * K::CS + (4 * wny + pi - K::DMIN) * K::RS + (4 * NP * wnx + pj) + K::PADL;   // + d_of(ky) * RS + d_of(kx)
  using WV = typename WVec<MW>::T;
  const int cg0 = (g * K::WM + wm) * MW;
  const float* wl = a.wp + ((size_t)(cg0 / 4) * a.ksteps * 64 + lane) * 4 + (cg0 % 4);
  auto wload = [&](int ks) -> WV { return *reinterpret_cast<const WV*>(wl + (size_t)ks * 256); };

  f32x4 acc[4][MW][NP];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < MW; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) acc[c][j][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  WV wreg[K::NBUFA];
  stage(0, 0);
#pragma unroll
  for (int i = 0; i < K::NBUFA - 1; ++i) wreg[i] = wload(i);

  for (int c = 0; c < a.nchunks; ++c) {
    const int buf = c & 1;
    // in flight: this chunk's window (issued a chunk ago) and, younger, the weight prefetch of the last NBUFA - 1 k-steps: loads
    // retire in order, so the window has landed once at most NBUFA - 1 loads remain
    wait_vmcnt<K::NBUFA - 1>();
    __builtin_amdgcn_s_barrier();
    if (c + 1 < a.nchunks) stage(c + 1, buf ^ 1);
    const float* win = smem + buf * K::BUF + bbase;
    const int ks0 = c * K::KSC;
    // operand reads one k-step ahead, ONE read behind the MW MFMAs of every patch (pinned: a block of NP reads in front of the MFMAs
    // leaves the matrix pipe a single queued instruction deep while it issues; csrc/conv_plane.hip, DESIGN 3.4 c)
    float b[2][NP];
    auto tap_off = [](int ks) { const int cq = ks / (KS * KS), ky = (ks / KS) % KS, kx = ks % KS; return cq * 4 * K::CS + P::d_of(ky) * K::RS + P::d_of(kx); };
#pragma unroll
    for (int p = 0; p < NP; ++p) b[0][p] = win[tap_off(0) + 4 * p];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < K::KSC; ++ks) {
      wreg[(ks + K::NBUFA - 1) % K::NBUFA] = wload(ks0 + ks + K::NBUFA - 1);      // the packed array carries spare k-steps
      const int ky = (ks / KS) % KS, kx = ks % KS;
      const int cls = 2 * P::par_of(ky) + P::par_of(kx);
      const WV w = wreg[ks % K::NBUFA];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int j = 0; j < MW; ++j) acc[cls][j][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ks & 1][p], wget<MW>(w, j), acc[cls][j][p], 0, 0, 0);
        if (ks + 1 < K::KSC) b[(ks + 1) & 1][p] = win[tap_off(ks + 1 < K::KSC ? ks + 1 : ks) + 4 * p];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: lane (patch row = lane >> 4, channel = lane & 15) holds 4 consecutive class columns of class row i: for both
  // row parities the 8 output pixels X = 2 jc .. 2 jc + 7 (x parities interleaved)
  const int i = i0 + 4 * wny + (lane >> 4);
#pragma unroll
  for (int py = 0; py < 2; ++py) {
    const int Y = 2 * i + py;
    if (Y < a.Hout) {
#pragma unroll
      for (int j = 0; j < MW; ++j) {
        const int co = 16 * (cg0 + j) + (lane & 15);
        const float bv = a.bias ? a.bias[co] : 0.f;
        float* orow = a.out + (((size_t)n * a.out_ctot + a.out_c0 + co) * a.Hout + Y) * a.Wout;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int X0 = 2 * (j0 + 4 * (NP * wnx + p));
          const f32x4 e = acc[2 * py][j][p], o = acc[2 * py + 1][j][p];
          float v[8] = {e[0], o[0], e[1], o[1], e[2], o[2], e[3], o[3]};
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float s = v[r] + bv;
            if (a.relu) s = s > 0.f ? s : s * a.slope;
            v[r] = s;
          }
          if (a.mask) {
            const float* mrow = a.mask + (((size_t)n * a.mask_ctot + a.mask_c0 + co) * a.Hout + Y) * a.Wout;
            float m[8];
            if (X0 + 7 < a.Wout) {
              const f32x4 m0 = *reinterpret_cast<const f32x4*>(mrow + X0), m1 = *reinterpret_cast<const f32x4*>(mrow + X0 + 4);
#pragma unroll
              for (int r = 0; r < 4; ++r) { m[r] = m0[r]; m[4 + r] = m1[r]; }
            } else {
#pragma unroll
              for (int r = 0; r < 8; ++r) m[r] = X0 + r < a.Wout ? mrow[X0 + r] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= m[r] > 0.f ? 1.f : a.mask_slope;      // the expression of bias_leaky_relu_bwd: the same bits
          }
          if (X0 + 7 < a.Wout) {
            *reinterpret_cast<f32x4*>(orow + X0) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(orow + X0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (X0 + r < a.Wout) orow[X0 + r] = v[r];
          }
        }
      }
    }
  }
}

template <class K>
__global__ void __launch_bounds__(256, 2)
tconv_mfma(Args a) {
  const unsigned per_xcd = (a.total + 7) / 8;
  unsigned t = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (blockIdx.x / 8 >= per_xcd || t >= a.total) return;
  const int g = t % a.ng; t /= a.ng;
  const int bx = t % a.tx; t /= a.tx;
  const int by = t % a.ty;
  const int n = t / a.ty;
  tconv_body<K>(a, g, bx, by, n);
}

constexpr int kSpare = 8;
constexpr int kChunkQuads = 2;
inline int ksteps_for(int Cin, int KS) { return cdiv(cdiv(Cin, 4), kChunkQuads) * kChunkQuads * KS * KS; }   // = fn2_conv_mfma_pack_weights' layout

template <class K>
static int launch(const Args& base, hipStream_t st) {
  Args a = base;
  const int Hc = cdiv(a.Hout, 2), Wc = cdiv(a.Wout, 2);
  a.tx = cdiv(Wc, K::TW); a.ty = cdiv(Hc, K::TH);
  a.ng = a.Cout / (16 * K::MW * K::WM);
  a.nchunks = cdiv(cdiv(a.Cin, 4), K::CQ);
  const long long tiles = (long long)a.N * a.tx * a.ty * a.ng;
  if (tiles > 0x3fffff00ll) return fail(FN2_ERR_UNSUPPORTED, "tconv: grid too large");
  a.total = (unsigned)tiles;
  constexpr size_t lds = sizeof(float) * 2 * K::BUF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_mfma<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((tconv_mfma<K>), dim3(8 * ((a.total + 7) / 8)), dim3(K::THREADS), lds, st, a);
  return check_launch("tconv_forward");
}

struct Variant {
  int ks, pad, mw, np, wm, wnx, wny;
  int (*fn)(const Args&, hipStream_t);
};

// (KS, PAD, MW, NP, WM, WNX, WNY, CQ)
#define FN2_TC_TILES(X, KS, PAD, CQ) \
  X(KS, PAD, 1, 7, 2, 2, 1, CQ) X(KS, PAD, 1, 7, 2, 1, 2, CQ) X(KS, PAD, 1, 7, 4, 1, 1, CQ) X(KS, PAD, 2, 3, 2, 2, 1, CQ) X(KS, PAD, 2, 3, 2, 1, 2, CQ) \
  X(KS, PAD, 1, 4, 2, 2, 1, CQ) X(KS, PAD, 1, 4, 4, 1, 1, CQ) X(KS, PAD, 2, 2, 2, 2, 1, CQ) X(KS, PAD, 1, 3, 2, 2, 1, CQ) X(KS, PAD, 1, 6, 2, 2, 1, CQ)
#define FN2_TC_LIST(X) FN2_TC_TILES(X, 4, 1, 2) FN2_TC_TILES(X, 5, 2, 1) FN2_TC_TILES(X, 3, 1, 2)
#define FN2_TC_ROW(KS, PAD, MW, NP, WM, WNX, WNY, CQ) {KS, PAD, MW, NP, WM, WNX, WNY, &launch<Cfg<KS, PAD, MW, NP, WM, WNX, WNY, CQ>>},
static const Variant kVariants[] = {FN2_TC_LIST(FN2_TC_ROW)};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int g_forced_variant = -1;

static bool variant_applies(const Variant& v, const Args& a, int KS, int pad) {
  return v.ks == KS && v.pad == pad && a.Cout % (16 * v.mw * v.wm) == 0;
}

// workgroups over 512 slots (two per CU) x accumulator tiles of a wave, mild penalty for small wave tiles
static double variant_cost(const Variant& v, const Args& a) {
  const int Hc = cdiv(a.Hout, 2), Wc = cdiv(a.Wout, 2);
  const long long wgs = (long long)a.N * cdiv(Wc, 4 * v.np * v.wnx) * cdiv(Hc, 4 * v.wny) * (a.Cout / (16 * v.mw * v.wm));
  const double rounds = (double)((wgs + 511) / 512);
  const double eff = 1.0 / (1.0 + 0.08 * (2.0 / v.mw - 1.0) + 0.03 * (7.0 / v.np - 1.0));
  return rounds * v.mw * v.np / eff;
}

static bool geometry_ok(int Cin, int Hin, int Win, int Cout, int Hout, int Wout, int kernel, int pad) {
  if (Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || Cout % 64 != 0 || Win % 4 != 0) return false;
  if (!((kernel == 4 && pad == 1) || (kernel == 5 && pad == 2) || (kernel == 3 && pad == 1))) return false;
  if ((long long)Cin * Hin * Win >= (1ll << 28)) return false;
  // the output may be up to one pixel larger than 2 (Hin - 1) + kernel - 2 pad (the data gradient of an odd-sized bottom)
  const int hmax = 2 * (Hin - 1) + kernel - 2 * pad + 1, wmax = 2 * (Win - 1) + kernel - 2 * pad + 1;
  return Hout >= 1 && Wout >= 1 && Hout <= hmax && Wout <= wmax;
}

}  // namespace tc
}  // namespace fn2

using namespace fn2;

FN2_API int fn2_tconv_supported(int Cin, int Hin, int Win, int Cout, int Hout, int Wout, int kernel, int pad) {
  return tc::geometry_ok(Cin, Hin, Win, Cout, Hout, Wout, kernel, pad) ? 1 : 0;
}

FN2_API int fn2_debug_set_tconv_variant(int v) { tc::g_forced_variant = v; return FN2_OK; }
FN2_API int fn2_tconv_num_variants(void) { return tc::kNumVariants; }

namespace fn2 {
int tconv_forward_masked(const float* bottom, const float* packed_weight, const float* bias, float* top,
                         int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                         int Cout, int Hout, int Wout, int top_channels, int top_c0, int kernel, int pad,
                         int relu, float negative_slope, const float* mask, int mask_channels, int mask_c0, float mask_slope, void* stream);
}

FN2_API int fn2_tconv_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                              int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                              int Cout, int Hout, int Wout, int top_channels, int top_c0, int kernel, int pad,
                              int relu, float negative_slope, void* stream) {
  return fn2::tconv_forward_masked(bottom, packed_weight, bias, top, N, Cin, Hin, Win, bottom_channels, bottom_c0, Cout, Hout, Wout, top_channels, top_c0,
                                   kernel, pad, relu, negative_slope, nullptr, 0, 0, 1.f, stream);
}

int fn2::tconv_forward_masked(const float* bottom, const float* packed_weight, const float* bias, float* top,
                              int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                              int Cout, int Hout, int Wout, int top_channels, int top_c0, int kernel, int pad,
                              int relu, float negative_slope, const float* mask, int mask_channels, int mask_c0, float mask_slope, void* stream) {
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "tconv: bad batch");
  if (N == 0) return FN2_OK;
  if (!bottom || !packed_weight || !top) return fail(FN2_ERR_INVALID_ARG, "tconv: null blob");
  if (!tc::geometry_ok(Cin, Hin, Win, Cout, Hout, Wout, kernel, pad))
    return fail(FN2_ERR_UNSUPPORTED, "tconv: unsupported geometry (Cin %d, %dx%d, Cout %d, out %dx%d, k %d p %d)", Cin, Hin, Win, Cout, Hout, Wout, kernel, pad);
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels)
    return fail(FN2_ERR_INVALID_ARG, "tconv: channel slice outside the blob");
  if (((reinterpret_cast<uintptr_t>(bottom) | reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(packed_weight)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "tconv: blobs must be 16-byte aligned");
  tc::Args a{};
  a.in = bottom; a.wp = packed_weight; a.bias = bias; a.out = top;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.in_ctot = bottom_channels; a.in_c0 = bottom_c0;
  a.Cout = Cout; a.Hout = Hout; a.Wout = Wout; a.out_ctot = top_channels; a.out_c0 = top_c0;
  a.ksteps = tc::ksteps_for(Cin, kernel) + tc::kSpare;
  a.slope = negative_slope; a.relu = relu;
  if (mask) {
    if (mask_c0 < 0 || mask_c0 + Cout > mask_channels) return fail(FN2_ERR_INVALID_ARG, "tconv: mask slice outside its blob");
    if ((reinterpret_cast<uintptr_t>(mask) & 15) != 0) return fail(FN2_ERR_UNSUPPORTED, "tconv: mask blob must be 16-byte aligned");
  }
  a.mask = mask; a.mask_ctot = mask_channels; a.mask_c0 = mask_c0; a.mask_slope = mask_slope;
  hipStream_t st = as_stream(stream);
  int best = -1;
  if (tc::g_forced_variant >= 0) {
    best = tc::g_forced_variant;
    if (best >= tc::kNumVariants || !tc::variant_applies(tc::kVariants[best], a, kernel, pad))
      return fail(FN2_ERR_UNSUPPORTED, "tconv: forced variant %d does not apply", best);
  } else {
    if (autotune_enabled(st)) {
      static TuneCache cache("tconv", tc::kNumVariants);
      const TuneKey key{N, Cin, Hin, Win, Cout, Hout, Wout, kernel * 16 + pad, bottom_channels == Cin, top_channels == Cout};
      auto usable = [&](int c) -> bool { return tc::variant_applies(tc::kVariants[c], a, kernel, pad); };
      best = autotune_pick(cache, key, tc::kNumVariants, st, [&](int c) -> int {
        return usable(c) ? tc::kVariants[c].fn(a, st) : FN2_ERR_UNSUPPORTED;
      }, usable);
    }
    if (best < 0) {
      double bc = 0;
      for (int i = 0; i < tc::kNumVariants; ++i) {
        if (!tc::variant_applies(tc::kVariants[i], a, kernel, pad)) continue;
        const double c = tc::variant_cost(tc::kVariants[i], a);
        if (best < 0 || c < bc) { best = i; bc = c; }
      }
    }
  }
  if (best < 0) return fail(FN2_ERR_UNSUPPORTED, "tconv: no kernel variant for this geometry");
  return tc::kVariants[best].fn(a, st);
}
