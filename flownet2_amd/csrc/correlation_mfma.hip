// Correlation forward fast path for gfx950: kernel_size 1, stride_1 1, MULTIPLY, pad == max_displacement
// (the FlowNetC instance: max_displacement 20, stride_2 2 -> 21x21 = 441 displacement channels).
//
// Replaces blob_rearrange_kernel2 + CorrelateData (reference: src/caffe/layers/correlation_layer.cu:23-114).
//
// Formulation.  With stride_2 = S2 the pixels split into S2*S2 parity classes; in class coordinates
// (y = S2*i + py, x = S2*j + px) the op is
//     top[(q,o), (i,j)] = 1/C * sum_c A[c,(i,j)] * B[c,(i+q, j+o)],      |q|,|o| <= R = max_disp / S2
// i.e. a 2-D *banded* matrix product between positions of the first map (M) and positions of the
// second (N), contracted over channels (K = C).  An M tile is a 4x4 patch of class positions, an N
// tile likewise; one M tile needs the (2R+4)/4 x (2R+4)/4 N tiles around it (6 x 6 for R = 10), of
// which 441/576 = 76.6 % of the products are inside the band -- against 33-47 % for a row-wise
// (1-D) banding.  Each tile product runs on v_mfma_f32_16x16x4_f32: exact fp32 (a k-ordered fma
// chain), 64 flop/clk/SIMD = the fp32 peak of the chip (157.3 TFLOP/s).
//
// Work decomposition.  One workgroup (8 waves) = (sample n, y-parity py, 4 class rows I, one N
// patch-row a, a 32-pixel x span): wave w owns the M tile (px = w % S2, patch w / S2) and the NB N
// tiles of patch-row a.  The 4 + 4 image rows a workgroup needs are staged through LDS in chunks of
// 16 channels (coalesced row reads straight from NCHW, zero fill outside the image = the
// reference's padding, de-interleaved by x parity so the MFMA operand reads are conflict-free),
// double-buffered with register staging (loads of chunk k+1 in flight under the MFMAs of chunk k,
// one barrier per chunk).  The accumulators are scattered into an LDS image of the output and
// written back as full 128-byte rows.  Two workgroups are resident per CU.
//
// HBM traffic: every workgroup reads each input row once per (I, a) pair it participates in; with
// the sample -> XCD mapping below those re-reads hit the 4 MiB L2, so HBM sees ~ the algorithmic
// 4*N*H*W*(2C + 441) bytes.
#include "correlation.hpp"

namespace fn2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kKC = 16;   // channels per LDS chunk (4 MFMA k-steps)

constexpr int up_4mod8(int v) { return v + ((4 - v % 8) + 8) % 8; }
constexpr int up_16mod32(int v) { return v + ((16 - v % 32) + 32) % 32; }
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int S2, int R>
struct Cfg {
  static constexpr int D = 2 * R + 1;                 // displacements per axis
  static constexpr int NB = (2 * R + 4 + 3) / 4;      // N tiles per axis around one M tile
  static constexpr int PJ = kWaves / S2;              // M patches along x per x-parity class
  static constexpr int SPANC = 4 * PJ;                // class columns per workgroup
  static constexpr int SPANPX = SPANC * S2;           // pixels per workgroup row (32)
  static constexpr int JW = SPANC - 4 + 4 * NB;       // class columns of the second map staged per row
  static constexpr int JWP = up_4mod8(JW);            // row stride == 4 (mod 8): 16 (i,j) lanes hit 16 banks
  static constexpr int AWP = up_4mod8(SPANC);
  static constexpr int BPX = JW * S2;                 // staged pixels per row of the second map
  static constexpr int AOFF = S2 * 4 * JWP;           // first-map region inside one channel slot
  static constexpr int KS = up_16mod32(AOFF + S2 * 4 * AWP);   // channel slot stride == 16 (mod 32)
  static constexpr int NBE = cdiv(kKC * 4 * BPX, kThreads);    // staged elements per thread
  static constexpr int NAE = cdiv(kKC * 4 * SPANPX, kThreads);
  static constexpr bool B_EXACT = (kKC * 4 * BPX) % kThreads == 0;
  static constexpr bool A_EXACT = (kKC * 4 * SPANPX) % kThreads == 0;
  static constexpr int XS = SPANPX + 1;               // output-image row stride in LDS
  static constexpr int OROWS = 16 * D;                // (mi, ni, o) rows of the output image
  static constexpr int LDS_FLOATS = cmax(2 * kKC * KS, OROWS * XS);
  static_assert(kWaves % S2 == 0, "waves must split evenly over x parities");
  static_assert(kThreads % SPANPX == 0, "store phase mapping");
};

struct MfmaArgs {
  int N, C, H, W;
  int NI, NSPAN;        // M patch rows per y-parity class, x spans
  int G, GP;            // logical workgroups, workgroups per XCD
};

template <int S2, int R>
__global__ void __launch_bounds__(kThreads, 4)
corr_fwd_mfma(const float* __restrict__ b0, const float* __restrict__ b1, float* __restrict__ top, MfmaArgs g) {
  using K = Cfg<S2, R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // ---- task decode; blocks b, b+8, b+16.. run on one XCD, give them one contiguous task range ----
  const int L = (int)(blockIdx.x % 8) * g.GP + (int)(blockIdx.x / 8);
  if (L >= g.G) return;
  int t = L;
  const int span = t % g.NSPAN; t /= g.NSPAN;
  const int I = t % g.NI; t /= g.NI;
  const int py = t % S2; t /= S2;
  const int a = t % K::NB; t /= K::NB;
  const int n = t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = wave % S2, Jw = wave / S2;
  const int i0 = 4 * I, jS = K::SPANC * span, jw = jS + 4 * Jw;
  const int Hc = (g.H - py + S2 - 1) / S2;      // class rows of this y parity
  const int Wc = (g.W - px + S2 - 1) / S2;      // class cols of this wave's x parity
  if (i0 >= Hc) return;                         // whole workgroup (uniform): no output rows

  const int kk = lane >> 4, ni = (lane & 15) >> 2, nj = lane & 3;
  const size_t plane = (size_t)g.H * g.W;
  const float* a_n = b0 + (size_t)n * g.C * plane;
  const float* b_n = b1 + (size_t)n * g.C * plane;

  // Which N tiles of this wave can be non-zero (x direction), and is any staged B row inside the image?
  unsigned bmask = 0;
  if (jw < Wc) {
#pragma unroll
    for (int b = 0; b < K::NB; ++b) {
      const int j2 = jw - R + 4 * b;
      if (j2 + 3 >= 0 && j2 < Wc) bmask |= 1u << b;
    }
  }
  bmask = __builtin_amdgcn_readfirstlane(bmask);
  const int i2_0 = i0 - R + 4 * a;
  const bool rows_live = (i2_0 + 3 >= 0) && (i2_0 < Hc);

  f32x4 acc[K::NB];
#pragma unroll
  for (int b = 0; b < K::NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (rows_live) {
    // ---- per-thread staging plan (constant over channel chunks) ----
    int gB[K::NBE], lB[K::NBE], gA[K::NAE], lA[K::NAE];
    unsigned vB = 0, vA = 0;
#pragma unroll
    for (int k = 0; k < K::NBE; ++k) {
      const int e = tid + kThreads * k;
      const int kc = e / (4 * K::BPX), rem = e % (4 * K::BPX);
      const int row = rem / K::BPX, col = rem % K::BPX;
      const int yb = S2 * (i2_0 + row) + py, xb = S2 * (jS - R) + col;
      const bool ok = (e < kKC * 4 * K::BPX) && yb >= 0 && yb < g.H && (i2_0 + row) >= 0 && xb >= 0 && xb < g.W;
      gB[k] = ok ? (kc * (int)plane + yb * g.W + xb) : 0;
      lB[k] = (e < kKC * 4 * K::BPX) ? (kc * K::KS + (col % S2) * (4 * K::JWP) + row * K::JWP + col / S2) : -1;
      vB |= ok ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < K::NAE; ++k) {
      const int e = tid + kThreads * k;
      const int kc = e / (4 * K::SPANPX), rem = e % (4 * K::SPANPX);
      const int row = rem / K::SPANPX, col = rem % K::SPANPX;
      const int ya = S2 * (i0 + row) + py, xa = S2 * jS + col;
      const bool ok = (e < kKC * 4 * K::SPANPX) && ya < g.H && xa < g.W;
      gA[k] = ok ? (kc * (int)plane + ya * g.W + xa) : 0;
      lA[k] = (e < kKC * 4 * K::SPANPX) ? (kc * K::KS + K::AOFF + (col % S2) * (4 * K::AWP) + row * K::AWP + col / S2) : -1;
      vA |= ok ? (1u << k) : 0u;
    }
    float sB[K::NBE], sA[K::NAE];

    auto load_chunk = [&](int chunk) {
      const float* pb = b_n + (size_t)chunk * kKC * plane;
      const float* pa = a_n + (size_t)chunk * kKC * plane;
#pragma unroll
      for (int k = 0; k < K::NBE; ++k) sB[k] = pb[gB[k]];      // invalid elements read offset 0 (always mapped) ...
#pragma unroll
      for (int k = 0; k < K::NAE; ++k) sA[k] = pa[gA[k]];
    };
    auto store_chunk = [&](float* buf) {
#pragma unroll
      for (int k = 0; k < K::NBE; ++k)
        if (K::B_EXACT || lB[k] >= 0) buf[lB[k]] = ((vB >> k) & 1u) ? sB[k] : 0.f;   // ... and become the zero padding here
#pragma unroll
      for (int k = 0; k < K::NAE; ++k)
        if (K::A_EXACT || lA[k] >= 0) buf[lA[k]] = ((vA >> k) & 1u) ? sA[k] : 0.f;
    };

    // operand addresses: lane (kk, ni, nj) of the 16x16x4 fragment reads channel kk of the k-step
    const int aAddr = kk * K::KS + K::AOFF + px * (4 * K::AWP) + ni * K::AWP + 4 * Jw + nj;
    const int bAddr = kk * K::KS + px * (4 * K::JWP) + ni * K::JWP + 4 * Jw + nj;
    auto compute = [&](const float* buf) {
#pragma unroll
      for (int ks = 0; ks < kKC / 4; ++ks) {
        const float av = buf[aAddr + 4 * ks * K::KS];
#pragma unroll
        for (int b = 0; b < K::NB; ++b) {
          if (bmask & (1u << b)) {
            const float bv = buf[bAddr + 4 * ks * K::KS + 4 * b];
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[b], 0, 0, 0);
          }
        }
      }
    };

    const int nchunks = g.C / kKC;
    float* buf0 = smem;
    float* buf1 = smem + kKC * K::KS;
    load_chunk(0);
    store_chunk(buf0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
      if (ch + 1 < nchunks) load_chunk(ch + 1);
      compute(buf0);
      if (ch + 1 < nchunks) store_chunk(buf1);
      __syncthreads();
      if (ch + 1 < nchunks) {
        if (ch + 2 < nchunks) load_chunk(ch + 2);
        compute(buf1);
        if (ch + 2 < nchunks) store_chunk(buf0);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: accumulators -> LDS image [mi][ni][o][x] -> coalesced rows of top ----
  const float sumelems = (float)g.C;      // kernel_size^2 * channels, correlation_layer.cu:108
  const int mi = lane >> 4;               // C/D layout of 16x16 MFMA: row = 4*(lane>>4) + reg, col = lane & 15
#pragma unroll
  for (int b = 0; b < K::NB; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {         // reg r <-> mj
      const int oo = 4 * b + nj - r;      // o + R
      if (oo >= 0 && oo < K::D)
        smem[((mi * 4 + ni) * K::D + oo) * K::XS + S2 * (4 * Jw + r) + px] = acc[b][r] / sumelems;
    }
  }
  __syncthreads();
  const int xl = tid % K::SPANPX;
  const int x = S2 * jS + xl;
  const size_t top_n = (size_t)n * K::D * K::D;
  if (x < g.W) {
    for (int rowid = tid / K::SPANPX; rowid < K::OROWS; rowid += kThreads / K::SPANPX) {
      const int rmi = rowid / (4 * K::D), rni = (rowid / K::D) % 4, oo = rowid % K::D;
      const int qq = 4 * a + rni - rmi;   // q + R
      const int y = S2 * (i0 + rmi) + py;
      if (qq >= 0 && qq < K::D && y < g.H)
        top[((top_n + (size_t)qq * K::D + oo) * g.H + y) * g.W + x] = smem[rowid * K::XS + xl];
    }
  }
}

template <int S2, int R>
static int launch(const CorrGeom& cg, const float* b0, const float* b1, float* top, hipStream_t st) {
  using K = Cfg<S2, R>;
  MfmaArgs g;
  g.N = cg.N; g.C = cg.C; g.H = cg.H; g.W = cg.W;
  const int Hc = (cg.H + S2 - 1) / S2, Wc = (cg.W + S2 - 1) / S2;
  g.NI = (Hc + 3) / 4;
  g.NSPAN = (Wc + K::SPANC - 1) / K::SPANC;
  const long long G = (long long)cg.N * K::NB * S2 * g.NI * g.NSPAN;
  if (G > (1ll << 30)) return fail(FN2_ERR_UNSUPPORTED, "correlation: problem too large for the MFMA path");
  g.G = (int)G;
  g.GP = (g.G + 7) / 8;
  const size_t lds = sizeof(float) * K::LDS_FLOATS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_mfma<S2, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((corr_fwd_mfma<S2, R>), dim3(8 * g.GP), dim3(kThreads), lds, st, b0, b1, top, g);
  return check_launch("correlation_forward (mfma)");
}

bool corr_fwd_mfma_supported(const CorrGeom& g) {
  if (g.K != 1 || g.s1 != 1 || g.type != FN2_CORR_MULTIPLY || g.pad != g.md) return false;
  if (g.C % kKC != 0) return false;
  if ((long long)g.C * g.H * g.W >= (1ll << 30)) return false;      // 32-bit staging offsets
  if (g.s2 == 2 && g.ngr == 10) return true;     // FlowNetC / FlowNet2: max_displacement 20, stride_2 2
  if (g.s2 == 1 && g.ngr == 4) return true;      // 9x9 cost volumes (max_displacement 4, stride_2 1)
  if (g.s2 == 2 && g.ngr == 4) return true;
  if (g.s2 == 1 && g.ngr == 8) return true;
  return false;
}

int corr_fwd_mfma_launch(const CorrGeom& g, const float* b0, const float* b1, float* top, hipStream_t st) {
  if (g.s2 == 2 && g.ngr == 10) return launch<2, 10>(g, b0, b1, top, st);
  if (g.s2 == 1 && g.ngr == 4) return launch<1, 4>(g, b0, b1, top, st);
  if (g.s2 == 2 && g.ngr == 4) return launch<2, 4>(g, b0, b1, top, st);
  if (g.s2 == 1 && g.ngr == 8) return launch<1, 8>(g, b0, b1, top, st);
  return fail(FN2_ERR_UNSUPPORTED, "correlation: no MFMA instantiation for stride_2 %d, radius %d", g.s2, g.ngr);
}

}  // namespace fn2
