// Correlation backward fast path for gfx950 (kernel_size 1, stride_1 1, MULTIPLY, pad == max_displacement).
//
// Replaces CorrelateDataBackward0 / CorrelateDataBackward1 (reference: src/caffe/layers/correlation_layer.cu:117-249,
// launched per sample at :546-572): one thread per bottom element looping over the 441 displacements with strided
// top-diff reads.  Here both gradients are the same 2-D banded GEMM as the forward (correlation_mfma.hip), with the
// roles of channels and displacements exchanged:
//
//   WHICH 0:  d0[c, m] = 1/C * sum_k  G[m,k] * b1[c,k]        G[m,k] = topdiff[(k - m), m]   (displacement k-m, taken at m)
//   WHICH 1:  d1[c, k] = 1/C * sum_m  G[m,k] * b0[c,m]        same G, contracted over the other index
//
// with m, k 4x4 patches of class positions (x/y parity classes of stride_2, as in the forward).  One MFMA
// (v_mfma_f32_16x16x4_f32, exact fp32) multiplies a [16 output positions x 4 contraction positions] slab of G with a
// [4 positions x 16 channels] slab of the other feature map.
//
// Workgroup (8 waves) = (sample, y parity, 4 class rows of OUTPUT positions, 32-pixel x span, 64-channel quarter):
// wave = one 4x4 output patch x 64 channels (16 accumulator VGPRs).  For each of the <= 6 patch-rows `a` of the
// contraction side the wave first gathers its slab of G (24 VGPRs: 6 patches x 4 rows, zero outside the band / image,
// through a raw buffer descriptor) and then walks the 4 channel chunks: the 4 image rows of the other map are staged
// through LDS exactly as in the forward (coalesced NCHW row reads, zero fill = padding, [row][parity][col][16 ch] with
// the same quad swizzle, double buffered) and each chunk costs 24 ds_read_b32 + 24 MFMAs per wave.
// The 64 x 4 x 32 output block goes through LDS and leaves as 128-byte rows.  No atomics: deterministic.
#include "correlation.hpp"

namespace fn2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace bwd {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kKC = 16;       // channels per staged chunk = one MFMA N tile
constexpr int kCQ = 64;       // channels per workgroup
constexpr int kNCH = kCQ / kKC;

constexpr int up_mod(int v, int r, int m) { return v + ((r - v % m) + m) % m; }
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int S2, int R>
struct Cfg {
  static constexpr int D = 2 * R + 1;
  static constexpr int NB = (2 * R + 4 + 3) / 4;      // contraction patches per axis around one output patch
  static constexpr int PJ = kWaves / S2;
  static constexpr int SPANC = 4 * PJ;
  static constexpr int SPANPX = SPANC * S2;
  static constexpr int JW = SPANC - 4 + 4 * NB;       // staged class columns per row
  static constexpr int BPX = JW * S2;                 // staged pixels per row
  static constexpr int BPL = JW * kKC;                // floats per (row, parity) plane
  static constexpr int BRS = up_mod(S2 * BPL, 8, 16); // row stride (same image as the forward's B region)
  static constexpr int CHUNK = 4 * BRS;
  static constexpr int SWAVES = cdiv(4 * BPX, 64);    // waves that stage
  static constexpr int XS = SPANPX + 1;
  static constexpr int OUT_FLOATS = kCQ * 4 * XS;
  static constexpr int LDS_FLOATS = cmax(2 * CHUNK, OUT_FLOATS);
  static_assert(SWAVES <= kWaves, "staging does not fit the workgroup");
  static_assert(kThreads % SPANPX == 0, "store phase mapping");
};

struct Args {
  int N, C, H, W;
  int NI, NSPAN, NCQ;   // output patch rows per y parity, x spans, 64-channel quarters
  int G, GP;
};

template <int S2, int R, int WHICH>
__global__ void __launch_bounds__(kThreads, 4)
corr_bwd_mfma(const float* __restrict__ other, const float* __restrict__ top_diff, float* __restrict__ out, Args g) {
  using K = Cfg<S2, R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int L = (int)(blockIdx.x % 8) * g.GP + (int)(blockIdx.x / 8);
  if (L >= g.G) return;
  int t = L;
  const int cq = t % g.NCQ; t /= g.NCQ;
  const int span = t % g.NSPAN; t /= g.NSPAN;
  const int I = t % g.NI; t /= g.NI;
  const int py = t % S2; t /= S2;
  const int n = t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = wave % S2, Jw = wave / S2;
  const int i0 = 4 * I, jS = K::SPANC * span, jw = jS + 4 * Jw;
  const int Hc = (g.H - py + S2 - 1) / S2;
  if (i0 >= Hc) return;

  const int plane = g.H * g.W;
  const float* src_n = other + ((size_t)n * g.C + (size_t)cq * kCQ) * plane;     // the 64 channels of this workgroup
  const float* g_n = top_diff + (size_t)n * K::D * K::D * plane;
  const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_n), 0, 4u * K::D * K::D * (unsigned)plane, 0x00020000);
  constexpr unsigned OOB = 0x7ffffff0u;

  // lane roles inside the 16x16x4 fragments
  const int kk = lane >> 4;                     // contraction column of the k-step (both operands)
  const int pi = (lane & 15) >> 2, pj = lane & 3;   // A operand: output position (pi, pj) of the wave's patch
  const int ch = lane & 15;                     // B operand: channel inside the 16-channel chunk

  // ---- staging plan: thread = one pixel of the 4 staged rows, 16 channels (as the forward's B region) ----
  const bool stager_wave = wave < K::SWAVES;
  unsigned voff = OOB;
  int laddr = -1, wsw = 0;
  int srow = 0, scol = 0;
  if (stager_wave) {
    srow = tid / K::BPX;
    scol = tid % K::BPX;
    if (srow < 4) {
      laddr = srow * K::BRS + (scol % S2) * K::BPL + (scol / S2) * kKC;
      wsw = (((scol / S2) >> 1) & 3) ^ (((scol % S2) << 1) & 3);
    }
  }
  const unsigned plane_bytes = 4u * (unsigned)plane;
  const unsigned chunk_bytes = 4u * kKC * (unsigned)plane;
  float sv[kKC];

  // operand read address of the staged map: position (row ks, column 4Jw + 4b + kk), channel ch
  const int rsw = ((2 * Jw + (kk >> 1)) & 3) ^ ((px << 1) & 3);
  const int xAddr0 = px * K::BPL + (4 * Jw + kk) * kKC + 4 * ((ch >> 2) ^ rsw) + (ch & 3);
  const int xAddr1 = px * K::BPL + (4 * Jw + kk) * kKC + 4 * ((ch >> 2) ^ rsw ^ 2) + (ch & 3);

  // two partial accumulators per channel chunk (even / odd contraction patches): consecutive MFMAs never hit the
  // same accumulator (a dependent 16x16x4 pair stalls 8 cycles)
  f32x4 acc[kNCH][2];
#pragma unroll
  for (int c = 0; c < kNCH; ++c) acc[c][0] = acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  float* buf0 = smem;
  float* buf1 = smem + K::CHUNK;

  for (int a = 0; a < K::NB; ++a) {
    // contraction-side patch row: class rows r0 .. r0+3
    const int r0 = i0 - R + 4 * a;
    if (r0 + 3 < 0 || r0 >= Hc) continue;            // all four rows outside the image: contributes nothing (uniform)

    // ---- gather this wave's slab of G: Gv[b][ks] for contraction position (r0 + ks, jw - R + 4b + kk) ----
    float Gv[K::NB][4];
#pragma unroll
    for (int b = 0; b < K::NB; ++b) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // displacement from map-0 position to map-1 position, and the map-0 position where top_diff is taken
        int q, o, yy, xx;
        if (WHICH == 0) {           // output = map-0 position (i0+pi, jw+pj); contraction = map-1 position
          q = (r0 + ks) - (i0 + pi);
          o = (jw - R + 4 * b + kk) - (jw + pj);
          yy = S2 * (i0 + pi) + py;
          xx = S2 * (jw + pj) + px;
        } else {                    // output = map-1 position (i0+pi, jw+pj); contraction = map-0 position
          q = (i0 + pi) - (r0 + ks);
          o = (jw + pj) - (jw - R + 4 * b + kk);
          yy = S2 * (r0 + ks) + py;
          xx = S2 * (jw - R + 4 * b + kk) + px;
        }
        const bool ok = q >= -R && q <= R && o >= -R && o <= R && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        const unsigned off = ok ? 4u * (unsigned)(((q + R) * K::D + (o + R)) * plane + yy * g.W + xx) : OOB;
        Gv[b][ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rs, off, 0, 0));
      }
    }

    // ---- staging offsets of the 4 rows r0 .. r0+3 of the other map ----
    voff = OOB;
    if (stager_wave && srow < 4) {
      const int ir = r0 + srow, yb = S2 * ir + py, xb = S2 * (jS - R) + scol;
      if (ir >= 0 && yb < g.H && xb >= 0 && xb < g.W) voff = 4u * (unsigned)(yb * g.W + xb);
    }
    auto load_chunk = [&](int c16) {
      if (laddr >= 0) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_n + (size_t)c16 * kKC * plane), 0, chunk_bytes, 0x00020000);
#pragma unroll
        for (int kc = 0; kc < kKC; ++kc)
          sv[kc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, kc * plane_bytes, 0));
      }
    };
    auto store_chunk = [&](float* buf) {
      if (laddr >= 0) {
#pragma unroll
        for (int q4 = 0; q4 < kKC / 4; ++q4)
          *reinterpret_cast<f32x4*>(buf + laddr + 4 * (q4 ^ wsw)) = f32x4{sv[4 * q4], sv[4 * q4 + 1], sv[4 * q4 + 2], sv[4 * q4 + 3]};
      }
    };
    auto compute = [&](const float* buf, f32x4 (&accc)[2]) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int b = 0; b < K::NB; ++b) {
          const float xv = buf[((b & 1) ? xAddr1 : xAddr0) + ks * K::BRS + 4 * b * kKC];
          accc[b & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gv[b][ks], xv, accc[b & 1], 0, 0, 0);
        }
      }
    };

    // ---- 4 channel chunks, double buffered (chunk c16 in buffer c16 & 1) ----
    __syncthreads();                  // everybody is done with both buffers of the previous patch row
    load_chunk(0);
    store_chunk(buf0);
    load_chunk(1);
    __syncthreads();
    compute(buf0, acc[0]);
    store_chunk(buf1);
    load_chunk(2);
    __syncthreads();
    compute(buf1, acc[1]);
    store_chunk(buf0);
    load_chunk(3);
    __syncthreads();
    compute(buf0, acc[2]);
    store_chunk(buf1);
    __syncthreads();
    compute(buf1, acc[3]);
  }

  // ---- epilogue: acc[c16][r] = value for output position (pi = lane>>4, pj = r), channel c16*16 + (lane & 15) ----
  __syncthreads();
  const float sumelems = (float)g.C;
  const bool pow2 = (g.C & (g.C - 1)) == 0;
  const float rcp = 1.0f / sumelems;
  const int opi = lane >> 4;
#pragma unroll
  for (int c = 0; c < kNCH; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        const float v = acc[c][0][r] + acc[c][1][r];
        smem[((c * kKC + ch) * 4 + opi) * K::XS + S2 * (4 * Jw + r) + px] = pow2 ? v * rcp : v / sumelems;
      }
  }
  __syncthreads();
  const int xl = tid % K::SPANPX;
  const int x = S2 * jS + xl;
  if (x < g.W) {
    float* out_n = out + ((size_t)n * g.C + (size_t)cq * kCQ) * plane;
    for (int rowid = tid / K::SPANPX; rowid < kCQ * 4; rowid += kThreads / K::SPANPX) {
      const int c = rowid >> 2, rpi = rowid & 3;
      const int y = S2 * (i0 + rpi) + py;
      if (y < g.H) out_n[(size_t)c * plane + (size_t)y * g.W + x] = smem[rowid * K::XS + xl];
    }
  }
}

template <int S2, int R, int WHICH>
static int launch(const CorrGeom& cg, const float* other, const float* top_diff, float* out, hipStream_t st) {
  using K = Cfg<S2, R>;
  Args g;
  g.N = cg.N; g.C = cg.C; g.H = cg.H; g.W = cg.W;
  const int Hc = (cg.H + S2 - 1) / S2, Wc = (cg.W + S2 - 1) / S2;
  g.NI = (Hc + 3) / 4;
  g.NSPAN = (Wc + K::SPANC - 1) / K::SPANC;
  g.NCQ = cg.C / kCQ;
  const long long G = (long long)cg.N * S2 * g.NI * g.NSPAN * g.NCQ;
  if (G > (1ll << 30)) return fail(FN2_ERR_UNSUPPORTED, "correlation backward: problem too large for the MFMA path");
  g.G = (int)G;
  g.GP = (g.G + 7) / 8;
  const size_t lds = sizeof(float) * K::LDS_FLOATS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_bwd_mfma<S2, R, WHICH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((corr_bwd_mfma<S2, R, WHICH>), dim3(8 * g.GP), dim3(kThreads), lds, st, other, top_diff, out, g);
  return check_launch("correlation_backward (mfma)");
}

}  // namespace bwd

bool corr_bwd_mfma_supported(const CorrGeom& g) {
  if (g.K != 1 || g.s1 != 1 || g.type != FN2_CORR_MULTIPLY || g.pad != g.md) return false;
  if (g.C % bwd::kCQ != 0) return false;
  if ((long long)g.topC * g.H * g.W >= (1ll << 28) || (long long)g.C * g.H * g.W >= (1ll << 28)) return false;
  return (g.s2 == 2 && g.ngr == 10) || (g.s2 == 1 && g.ngr == 4);
}

// which = 0: bottom0 diff (other = bottom1);  which = 1: bottom1 diff (other = bottom0)
int corr_bwd_mfma_launch(const CorrGeom& g, int which, const float* other, const float* top_diff, float* out, hipStream_t st) {
  if (g.s2 == 2 && g.ngr == 10) return which == 0 ? bwd::launch<2, 10, 0>(g, other, top_diff, out, st) : bwd::launch<2, 10, 1>(g, other, top_diff, out, st);
  if (g.s2 == 1 && g.ngr == 4) return which == 0 ? bwd::launch<1, 4, 0>(g, other, top_diff, out, st) : bwd::launch<1, 4, 1>(g, other, top_diff, out, st);
  return fail(FN2_ERR_UNSUPPORTED, "correlation backward: no MFMA instantiation for stride_2 %d, radius %d", g.s2, g.ngr);
}

}  // namespace fn2
