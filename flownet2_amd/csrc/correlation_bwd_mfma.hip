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
using f32x2 = __attribute__((ext_vector_type(2))) float;

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

// ---------------------------------------------------------------------------------------------------------------------
// Second generation (stride_2 2, radius 10, W % 4 == 0, 16-byte aligned blobs): the same decomposition and the same
// arithmetic, with the staging of the forward kernels.  The 4 rows x 72 pixels x 16 channels of the other map arrive by
// 16-byte LDS-DMA in their natural [channel][row][column] order (zero padding = out-of-range for the buffer descriptor),
// two buffers, ONE barrier per chunk (the first generation spends five per patch row on a register round trip: 16 loads +
// 4 ds_write_b128 per stager lane and chunk); the slab of G for the next patch row is gathered while the last chunk of the
// current one is multiplied.  B operand = one ds_read_b32 at lane base + immediate (2-way bank conflicts: the channel stride
// is a multiple of 4 dwords).
// ---------------------------------------------------------------------------------------------------------------------
namespace dma {
constexpr int R = 10, S2 = 2;
using K = Cfg<S2, R>;
constexpr int RS = K::BPX;                    // 72 dwords per staged row
constexpr int SLOTS_R = RS / 4;               // 18 slots
constexpr int SLOTS_C = 4 * SLOTS_R + 1;      // 73: one pad slot per channel -> channel stride 292 dwords == 4 (mod 32)
constexpr int CS = 4 * SLOTS_C;
constexpr int SLOTS = kKC * SLOTS_C;          // 1168 per chunk
constexpr int NRUN = cdiv(SLOTS, 64);         // 19
constexpr int RPW = cdiv(NRUN, kWaves);       // 3
constexpr int BUF = NRUN * 256;
constexpr int LDS_FLOATS = cmax(2 * BUF, K::OUT_FLOATS);
using lds_ptr_t = __attribute__((address_space(3))) void*;

__device__ __forceinline__ void stage(__amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[RPW], unsigned dst, int wave, unsigned soff) {
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = i * kWaves + wave;
    if (r < NRUN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)r), 16, voff[i], soff, 0, 0);
  }
}

template <int WHICH>
__device__ __forceinline__ void gather_g(float (&Gv)[K::NB][4], __amdgpu_buffer_rsrc_t g_rs, const Args& g, int plane, int r0, int i0, int jw, int pi, int pj, int kk,
                                         int py, int px) {
  constexpr unsigned OOB = 0x7ffffff0u;
#pragma unroll
  for (int b = 0; b < K::NB; ++b) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      int q, o, yy, xx;
      if (WHICH == 0) {
        q = (r0 + ks) - (i0 + pi);
        o = (jw - R + 4 * b + kk) - (jw + pj);
        yy = S2 * (i0 + pi) + py;
        xx = S2 * (jw + pj) + px;
      } else {
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
}

template <int WHICH>
__global__ void __launch_bounds__(kThreads, 4)
corr_bwd_dma(const float* __restrict__ other, const float* __restrict__ top_diff, float* __restrict__ out, Args g) {
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
  const float* src_n = other + ((size_t)n * g.C + (size_t)cq * kCQ) * plane;
  const __amdgpu_buffer_rsrc_t s_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_n), 0, 4u * kCQ * (unsigned)plane, 0x00020000);
  const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(top_diff + (size_t)n * K::D * K::D * plane), 0, 4u * K::D * K::D * (unsigned)plane, 0x00020000);
  constexpr unsigned OOB = 0x7ffffff0u;

  const int kk = lane >> 4, pi = (lane & 15) >> 2, pj = lane & 3, ch = lane & 15;

  // live patch rows of the contraction side: r0 = i0 - R + 4a, a in [alo, ahi]
  int alo = 0, ahi = K::NB - 1;
  while (alo < K::NB && i0 - R + 4 * alo + 3 < 0) ++alo;
  while (ahi >= 0 && i0 - R + 4 * ahi >= Hc) --ahi;

  // ---- DMA plan: slot -> (channel, row, 4 columns); column / channel part once, row part per patch row
  int srow[RPW];
  unsigned vxc[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int s = (i * kWaves + wave) * 64 + lane;
    srow[i] = -1; vxc[i] = 0;
    if (s < SLOTS) {
      const int c = s / SLOTS_C, rem = s % SLOTS_C;
      if (rem < 4 * SLOTS_R) {
        const int row = rem / SLOTS_R, gq = rem % SLOTS_R;
        const int xb = S2 * (jS - R) + 4 * gq;
        if (xb >= 0 && xb < g.W) { srow[i] = row; vxc[i] = 4u * (unsigned)(c * plane + xb); }
      }
    }
  }
  auto row_offsets = [&](int a, unsigned (&voff)[RPW]) {
    const int r0 = i0 - R + 4 * a;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      voff[i] = OOB;
      if (srow[i] >= 0) {
        const int ir = r0 + srow[i], yb = S2 * ir + py;
        if (ir >= 0 && yb < g.H) voff[i] = vxc[i] + 4u * (unsigned)(yb * g.W);
      }
    }
  };
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const unsigned chunk_bytes = 4u * kKC * (unsigned)plane;

  f32x4 acc[kNCH][2];
#pragma unroll
  for (int c = 0; c < kNCH; ++c) acc[c][0] = acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int bAddr = ch * CS + S2 * (4 * Jw + kk) + px;
  if (alo <= ahi) {
    float Gc[K::NB][4], Gn[K::NB][4];
    unsigned voff[RPW];
    gather_g<WHICH>(Gc, g_rs, g, plane, i0 - R + 4 * alo, i0, jw, pi, pj, kk, py, px);
    row_offsets(alo, voff);
    stage(s_rs, voff, lds_base, wave, 0u);
    int buf = 0;
    for (int a = alo; a <= ahi; ++a) {
#pragma unroll
      for (int c16 = 0; c16 < kNCH; ++c16) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c16 + 1 < kNCH) stage(s_rs, voff, lds_base + 4u * (unsigned)((buf ^ 1) * BUF), wave, (unsigned)(c16 + 1) * chunk_bytes);
        else if (a < ahi) {
          row_offsets(a + 1, voff);
          stage(s_rs, voff, lds_base + 4u * (unsigned)((buf ^ 1) * BUF), wave, 0u);
          gather_g<WHICH>(Gn, g_rs, g, plane, i0 - R + 4 * (a + 1), i0, jw, pi, pj, kk, py, px);
        }
        const float* sb = smem + buf * BUF + bAddr;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int b = 0; b < K::NB; ++b)
            acc[c16][b & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gc[b][ks], sb[ks * RS + 2 * 4 * b], acc[c16][b & 1], 0, 0, 0);
        buf ^= 1;
      }
      if (a < ahi) {
#pragma unroll
        for (int b = 0; b < K::NB; ++b)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) Gc[b][ks] = Gn[b][ks];
      }
    }
  }

  // ---- epilogue (as the first generation): accumulators -> LDS -> 128-byte rows
  __syncthreads();
  const float sumelems = (float)g.C;
  const bool pow2 = (g.C & (g.C - 1)) == 0;
  const float rcp = 1.0f / sumelems;
  const int opi = lane >> 4;
#pragma unroll
  for (int c = 0; c < kNCH; ++c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
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
}  // namespace dma

// ---------------------------------------------------------------------------------------------------------------------
// Third generation: G through LDS.  Generations 1 and 2 gather every wave's slab of G = top_diff[(k - m), m] with 24 scattered
// dword loads per lane and patch row -- ~28 cache lines per wave instruction, 192 such instructions per workgroup and patch row:
// the texture path, not the matrix pipe, sets their run time (the LDS-DMA staging of generation 2 changed nothing: 194 -> 190 us).
// Here the chunk is ONE contraction row r (class row of the other map) and everything a workgroup needs for it arrives by
// coalesced 16-byte LDS-DMA:
//   * the row of the other map, 64 channels x 72 pixels, natural order (channel stride 76 dwords);
//   * the rows of top_diff that hold G for that contraction row.  WHICH 0 (d bottom0): displacement row q = r - (i0 + pi) for
//     each of the 4 output rows pi, all 21 o: 4 x 21 rows of the 32 output pixels of the workgroup (128 bytes each).
//     WHICH 1 (d bottom1): q = (i0 + pi) - r, taken at the CONTRACTION position: 4 x 21 rows of the 72 staged pixels.
//   The A operand of patch b is then one ds_read_b32 at lane base + immediate (+ a lane mask for displacements outside the band).
// Rows r outside the image are skipped altogether (the earlier generations multiply their zeros).
// Same products as before, summed row by row over the contraction rows; two accumulators per channel chunk.
// ---------------------------------------------------------------------------------------------------------------------
namespace g3 {
#ifndef FN2_G3_ABL
#define FN2_G3_ABL 0          // profiling builds (scripts/probes/corr_bwd_variants.sh; wrong results): 1 no G-slab DMA, 2 no other-map DMA, 4 no MFMAs
#endif
constexpr int kAbl = FN2_G3_ABL;
constexpr int R = 10, S2 = 2, D = 21;
using K = Cfg<S2, R>;
using lds_ptr_t = __attribute__((address_space(3))) void*;
constexpr int BSL = K::BPX / 4 + 1;             // 19 slots per channel of the staged row (one pad slot): stride 76 dwords
constexpr int BCS = 4 * BSL;
constexpr int BSLOTS = kCQ * BSL;               // 1216 = 19 whole runs
constexpr int BRUN = BSLOTS / 64;
static_assert(BSLOTS % 64 == 0, "the other-map region ends on a run boundary");
template <int WHICH> struct G {
  static constexpr int GW = WHICH == 0 ? K::SPANPX : K::BPX;     // pixels per G row: the 32 output pixels / the 72 staged pixels
  // Row stride of the G slab in LDS: one 16-byte slot of padding per row.  With dense rows (32 or 72 dwords) the A-operand read --
  // lane (pi, pj, kk) at row pi * 21 + (kk - pj) + ... -- put all 64 lanes on 4 banks (stride 32) or on multiples of 8 (stride 72):
  // 68 % of the LDS cycles of the kernel were bank conflicts (PMC) and the LDS, not the matrix pipe, set its pace.
  static constexpr int GS = GW + 4;
  static constexpr int GSL = GW / 4;                             // data slots per row
  static constexpr int GSLS = GS / 4;                            // slots per row incl. the padding slot
  static constexpr int GSLOTS = 4 * D * GSLS;
  static constexpr int SLOTS = BSLOTS + GSLOTS;
  static constexpr int NRUN = cdiv(SLOTS, 64);
  static constexpr int RPW = cdiv(NRUN, kWaves);
  static constexpr int BUF = NRUN * 256;
  static constexpr int GOFF = BSLOTS * 4;                         // dword offset of the G slab inside a buffer
  static constexpr int LDS_FLOATS = cmax(2 * BUF, K::OUT_FLOATS);
};

template <int WHICH>
__device__ __forceinline__ void stage(const float* bsrc, unsigned bbytes, const float* gsrc, unsigned gbytes,
                                      const unsigned (&vb)[G<WHICH>::RPW], const int (&gpi)[G<WHICH>::RPW], unsigned dst, int wave, int r, int i0) {
  using T = G<WHICH>;
  constexpr unsigned OOB = 0x7ffffff0u;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bsrc), 0, bbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, gbytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < T::RPW; ++i) {
    const int run = i * kWaves + wave;
    if (run < BRUN) {
      if constexpr (!(kAbl & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)run), 16, vb[i], 0, 0, 0);
    } else if (run < T::NRUN && !(kAbl & 1)) {
      const int q = WHICH == 0 ? r - i0 - gpi[i] : i0 + gpi[i] - r;          // displacement row of this slab row
      const unsigned v = (gpi[i] >= 0 && q >= -R && q <= R) ? vb[i] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)run), 16, v, 0, 0, 0);
    }
  }
}

template <int WHICH>
__global__ void __launch_bounds__(kThreads, WHICH == 0 ? 4 : 2)
corr_bwd_g3(const float* __restrict__ other, const float* __restrict__ top_diff, float* __restrict__ out, Args g) {
  using T = G<WHICH>;
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
  const int i0 = 4 * I, jS = K::SPANC * span;
  const int Hc = (g.H - py + S2 - 1) / S2;
  if (i0 >= Hc) return;
  const int plane = g.H * g.W;
  const float* src_n = other + ((size_t)n * g.C + (size_t)cq * kCQ) * plane;
  const float* td_n = top_diff + (size_t)n * D * D * plane;
  constexpr unsigned OOB = 0x7ffffff0u;

  const int kk = lane >> 4, pi = (lane & 15) >> 2, pj = lane & 3, ch = lane & 15;
  const int rlo = i0 - R < 0 ? 0 : i0 - R, rhi = i0 + 3 + R > Hc - 1 ? Hc - 1 : i0 + 3 + R;

  // ---- DMA plan.  Runs [0, BRUN): (channel, 4 pixels) of the other map's row; the rest: (pi, oo, 4 pixels) of the G slab.
  // Per-lane byte offsets are relative to per-chunk base pointers (the image row / the displacement row move with r).
  unsigned vb[T::RPW];
  int gpi[T::RPW];
#pragma unroll
  for (int i = 0; i < T::RPW; ++i) {
    const int s = (i * kWaves + wave) * 64 + lane;
    vb[i] = OOB; gpi[i] = -1;
    if (s < BSLOTS) {
      const int c = s / BSL, gq = s % BSL;
      const int xb = S2 * (jS - R) + 4 * gq;
      if (gq < K::BPX / 4 && xb >= 0 && xb < g.W) vb[i] = 4u * (unsigned)(c * plane + xb);
    } else if (s < T::SLOTS) {
      const int sg = s - BSLOTS;
      const int spi = sg / (D * T::GSLS), oo = (sg / T::GSLS) % D, gq = sg % T::GSLS;
      if (gq >= T::GSL) {
        // the padding slot of a row: stays out of range (written as zeros, never read)
      } else if (WHICH == 0) {
        // top_diff[(q + R) * 21 + oo][y = 2 (i0 + pi) + py][2 jS + 4 gq ..], q = r - i0 - pi:
        //   channel = (r - i0 + R - 3) * 21  [per-chunk base]  +  21 * (3 - pi) + oo  [here]
        const int y = S2 * (i0 + spi) + py, x = S2 * jS + 4 * gq;
        if (y < g.H && x < g.W) { vb[i] = 4u * (unsigned)((D * (3 - spi) + oo) * plane + y * g.W + x); gpi[i] = spi; }
      } else {
        // top_diff[(q + R) * 21 + oo][y = 2 r + py][2 (jS - R) + 4 gq ..], q = i0 + pi - r:
        //   channel = (i0 - r + R) * 21, row y  [per-chunk base]  +  21 * pi + oo  [here]
        const int x = S2 * (jS - R) + 4 * gq;
        if (x >= 0 && x < g.W) { vb[i] = 4u * (unsigned)((D * spi + oo) * plane + x); gpi[i] = spi; }
      }
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

  // ---- operand addresses (dwords inside a buffer)
  const int bAddr = ch * BCS + S2 * (4 * Jw + kk) + px;                     // + c16 * 16 * BCS + 8 b
  int aAddr, aStep;
  int amask = 0;                                                            // bit b: displacement column of patch b is inside the band
  if (WHICH == 0) {
    aAddr = T::GOFF + (pi * D + (kk - pj)) * T::GS + S2 * (4 * Jw + pj) + px;   // + b * 4 * GS   (oo = 4 b + kk - pj)
    aStep = 4 * T::GS;
#pragma unroll
    for (int b = 0; b < K::NB; ++b) { const int oo = 4 * b + kk - pj; if (oo >= 0 && oo < D) amask |= 1 << b; }
  } else {
    aAddr = T::GOFF + (pi * D + (pj + 2 * R - kk)) * T::GS + S2 * (4 * Jw + kk) + px;   // + b * (8 - 4 * GS)   (oo = pj + 2R - 4 b - kk)
    aStep = 8 - 4 * T::GS;
#pragma unroll
    for (int b = 0; b < K::NB; ++b) { const int oo = pj + 2 * R - 4 * b - kk; if (oo >= 0 && oo < D) amask |= 1 << b; }
  }

  f32x4 acc[kNCH][2];
#pragma unroll
  for (int c = 0; c < kNCH; ++c) acc[c][0] = acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned td_bytes = 4u * D * D * (unsigned)plane, src_bytes = 4u * kCQ * (unsigned)plane;
  // chunk r: base pointers (may point below the blob for displacement rows that do not exist: those lanes are masked in stage())
#define FN2_G3_STAGE(rr, bb)                                                                                                     \
  do {                                                                                                                           \
    const int yb_ = S2 * (rr) + py;                                                                                              \
    const long long goff_ = WHICH == 0 ? (long long)((rr) - i0 + R - 3) * D * plane                                              \
                                       : (long long)(i0 - (rr) + R) * D * plane + (long long)yb_ * g.W;                          \
    stage<WHICH>(src_n + (size_t)yb_ * g.W, src_bytes - 4u * (unsigned)(yb_ * g.W), td_n + goff_,                                \
                 (unsigned)((long long)td_bytes - 4 * goff_), vb, gpi, lds_base + 4u * (unsigned)((bb) * T::BUF), wave, (rr), i0); \
  } while (0)

  FN2_G3_STAGE(rlo, 0);
  int buf = 0;
  for (int r = rlo; r <= rhi; ++r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (r < rhi) FN2_G3_STAGE(r + 1, buf ^ 1);
    const float* sb = smem + buf * T::BUF;
    float Gv[K::NB];
#pragma unroll
    for (int b = 0; b < K::NB; ++b) {
      const float v = sb[aAddr + b * aStep];
      Gv[b] = (amask >> b) & 1 ? v : 0.f;
    }
    if constexpr (!(kAbl & 4)) {
#pragma unroll
      for (int c16 = 0; c16 < kNCH; ++c16)
#pragma unroll
        for (int b = 0; b < K::NB; ++b)
          acc[c16][b & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gv[b], sb[bAddr + c16 * kKC * BCS + 8 * b], acc[c16][b & 1], 0, 0, 0);
    } else {
#pragma unroll
      for (int b = 0; b < K::NB; ++b) acc[0][b & 1][0] += Gv[b];
    }
    buf ^= 1;
  }
#undef FN2_G3_STAGE

  // ---- epilogue (as the earlier generations): accumulators -> LDS -> 128-byte rows
  __syncthreads();
  const float sumelems = (float)g.C;
  const bool pow2 = (g.C & (g.C - 1)) == 0;
  const float rcp = 1.0f / sumelems;
  const int opi = lane >> 4;
#pragma unroll
  for (int c = 0; c < kNCH; ++c) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float v = acc[c][0][rr] + acc[c][1][rr];
      smem[((c * kKC + ch) * 4 + opi) * K::XS + S2 * (4 * Jw + rr) + px] = pow2 ? v * rcp : v / sumelems;
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
}  // namespace g3


// ---------------------------------------------------------------------------------------------------------------------
// Fourth generation (round 6).  Ablation builds of generation 3 (scripts/probes/corr_bwd_variants.sh, profiles/r06_corr_bwd_notes.md):
// without any staging the kernel takes 45 us per bottom, without the other map's rows 61-64 (they come from L2: free), without the G slab
// 51-55 -- the rows of top_diff are touched ONCE per workgroup quartet and come from HBM, and a one-row-ahead prefetch does not cover that
// latency; the bottom-1 slab (84 rows x 72 staged pixels, 44 % of them used) also kept that variant at one workgroup per CU.  Here:
//   * G is staged THREE rows deep (ring of 3 slabs), the other map's row two deep: a row's G has two row times to arrive, and the wait
//     in front of a row (`s_waitcnt vmcnt(n)`, n = this wave's G instructions of the youngest slab) does not cover it;
//   * bottom 1 stages, for displacement column oo, only the 32 contraction pixels the workgroup's 16 output columns pair with that
//     displacement -- a window that slides by 2 pixels per displacement, cut at 16-byte slots: 36 dwords per row (as bottom 0) instead of
//     76: the slab is 12 KB for both bottoms, 74 KB of LDS per workgroup = two workgroups per CU for both;
//   * the operand reads of MFMA k + LA are issued behind MFMA k (`sched_barrier`), not as a block in front of the row's MFMAs.
// Measured and dropped (same bits, profiles/r06_corr_bwd_notes.md): one wave computing BOTH x parities of a patch for half of the channels, so
// that a conflict-free ds_read_b64 feeds two MFMAs (18 LDS instructions per 24 MFMAs instead of 30): 60.9 / 64.7 us against 60.0 / 60.6 --
// the LDS conflicts of this kernel (54 % of its LDS cycles) are not what bounds it.
// Same products, same order (row by row, per row chunk-major over the 6 contraction patches, two accumulators per chunk): the bits of
// generation 3 (tests/test_gpu_parity.py::test_correlation_backward_generations_agree_bitwise).
// ---------------------------------------------------------------------------------------------------------------------
namespace g4 {
constexpr int R = 10, S2 = 2, D = 21;
using K = Cfg<S2, R>;
using lds_ptr_t = __attribute__((address_space(3))) void*;
constexpr int BSL = K::BPX / 4 + 1;             // 19 slots per channel of the staged row (one pad slot): stride 76 dwords
constexpr int BCS = 4 * BSL;
constexpr int BSLOTS = kCQ * BSL;               // 1216 = 19 whole runs
constexpr int BRUN = BSLOTS / 64;
constexpr int BBUF = BRUN * 256;                // floats per other-map buffer
constexpr int GS = 36, GSLS = GS / 4;           // G row: 36 dwords = 9 slots (bottom 0: 32 data + 4 padding; bottom 1: the 36-dword window)
constexpr int GSLOTS = 4 * D * GSLS;            // 756
constexpr int GRUN = cdiv(GSLOTS, 64);          // 12
constexpr int GBUF = GRUN * 256;                // floats per G slab
constexpr int BRPW = cdiv(BRUN, kWaves), GRPW = cdiv(GRUN, kWaves);     // 3, 2 DMA instructions per wave at most
constexpr int LDS_FLOATS = cmax(2 * BBUF + 3 * GBUF, K::OUT_FLOATS);
static_assert(BSLOTS % 64 == 0 && K::OUT_FLOATS <= 2 * BBUF, "the epilogue image fits the other-map buffers");
constexpr int LA = 3;                           // operand reads run this many MFMAs ahead
#ifndef FN2_G4_ABL
#define FN2_G4_ABL 0          // profiling builds (scripts/probes/corr_bwd_variants.sh; wrong results): 1 no G-slab DMA, 2 no other-map DMA, 4 no MFMAs, 8 no epilogue stores
#endif
constexpr int kAbl4 = FN2_G4_ABL;

template <int WHICH>
__device__ __forceinline__ void g4_body(const float* __restrict__ other, const float* __restrict__ top_diff, float* __restrict__ out, const Args& g, int L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int t = L;
  const int cq = t % g.NCQ; t /= g.NCQ;
  const int span = t % g.NSPAN; t /= g.NSPAN;
  const int I = t % g.NI; t /= g.NI;
  const int py = t % S2; t /= S2;
  const int n = t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = wave % S2, Jw = wave / S2;
  const int i0 = 4 * I, jS = K::SPANC * span;
  const int Hc = (g.H - py + S2 - 1) / S2;
  if (i0 >= Hc) return;
  const int plane = g.H * g.W;
  const float* src_n = other + ((size_t)n * g.C + (size_t)cq * kCQ) * plane;
  const float* td_n = top_diff + (size_t)n * D * D * plane;
  constexpr unsigned OOB = 0x7ffffff0u;

  const int kk = lane >> 4, pi = (lane & 15) >> 2, pj = lane & 3, ch = lane & 15;
  const int rlo = i0 - R < 0 ? 0 : i0 - R, rhi = i0 + 3 + R > Hc - 1 ? Hc - 1 : i0 + 3 + R;

  // ---- DMA plan: per-lane byte offsets relative to per-row base pointers (the image row / the displacement row move with r)
  unsigned vbB[BRPW], vbG[GRPW];
  int gpi[GRPW];
#pragma unroll
  for (int i = 0; i < BRPW; ++i) {
    const int s = (i * kWaves + wave) * 64 + lane;
    vbB[i] = OOB;
    if (s < BSLOTS) {
      const int c = s / BSL, gq = s % BSL;
      const int xb = S2 * (jS - R) + 4 * gq;
      if (gq < K::BPX / 4 && xb >= 0 && xb < g.W) vbB[i] = 4u * (unsigned)(c * plane + xb);
    }
  }
#pragma unroll
  for (int i = 0; i < GRPW; ++i) {
    const int sg = (i * kWaves + wave) * 64 + lane;
    vbG[i] = OOB; gpi[i] = -1;
    if (sg < GSLOTS) {
      const int spi = sg / (D * GSLS), oo = (sg / GSLS) % D, gq = sg % GSLS;
      if (WHICH == 0) {
        // top_diff[(q + R) * 21 + oo][y = 2 (i0 + pi) + py][2 jS + 4 gq ..], q = r - i0 - pi:
        //   channel = (r - i0 + R - 3) * 21  [per-row base]  +  21 * (3 - pi) + oo  [here];  slot 8 of a row is padding
        const int y = S2 * (i0 + spi) + py, x = S2 * jS + 4 * gq;
        if (gq < 8 && y < g.H && x < g.W) { vbG[i] = 4u * (unsigned)((D * (3 - spi) + oo) * plane + y * g.W + x); gpi[i] = spi; }
      } else {
        // top_diff[(q + R) * 21 + oo][y = 2 r + py][window of displacement column o = oo - R], q = i0 + pi - r: the 16 output columns
        // jS .. jS + 15 pair with contraction columns jS - o .. jS - o + 15 = pixels 2 (jS - o) .. + 31; the window starts at the 16-byte
        // slot below (2 (jS - o) is 0 or 2 mod 4: the A operand adds that) and is 9 slots long
        //   channel = (i0 - r + R) * 21, row y  [per-row base]  +  21 * pi + oo  [here]
        const int x0 = S2 * (jS - (oo - R));
        const int xs = x0 - (((x0 % 4) + 4) % 4);
        const int x = xs + 4 * gq;
        if (x >= 0 && x < g.W) { vbG[i] = 4u * (unsigned)((D * spi + oo) * plane + x); gpi[i] = spi; }
      }
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const unsigned g_base = lds_base + 4u * 2u * BBUF;

  // ---- operand addresses (dwords)
  const int bAddr = ch * BCS + S2 * (4 * Jw + kk) + px;                     // + c16 * 16 * BCS + 8 b
  int aAddr, aStep;
  int amask = 0;                                                            // bit b: displacement column of patch b is inside the band
  if (WHICH == 0) {
    aAddr = (pi * D + (kk - pj)) * GS + S2 * (4 * Jw + pj) + px;            // + b * 4 * GS   (oo = 4 b + kk - pj)
    aStep = 4 * GS;
#pragma unroll
    for (int b = 0; b < K::NB; ++b) { const int oo = 4 * b + kk - pj; if (oo >= 0 && oo < D) amask |= 1 << b; }
  } else {
    aAddr = (pi * D + (pj + 2 * R - kk)) * GS + S2 * (4 * Jw + pj) + px + 2 * ((pj + kk) & 1);   // + b * (-4 GS)   (oo = pj + 2R - 4 b - kk)
    aStep = -4 * GS;
#pragma unroll
    for (int b = 0; b < K::NB; ++b) { const int oo = pj + 2 * R - 4 * b - kk; if (oo >= 0 && oo < D) amask |= 1 << b; }
  }

  f32x4 acc[kNCH][2];
#pragma unroll
  for (int c = 0; c < kNCH; ++c) acc[c][0] = acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned td_bytes = 4u * D * D * (unsigned)plane, src_bytes = 4u * kCQ * (unsigned)plane;
  auto stage_b = [&](int rr, int slot) {
    const int yb = S2 * rr + py;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_n + (size_t)yb * g.W), 0,
                                                                        src_bytes - 4u * (unsigned)(yb * g.W), 0x00020000);
    const unsigned dst = lds_base + 4u * (unsigned)(slot * BBUF);
#pragma unroll
    for (int i = 0; i < BRPW; ++i) {
      const int run = i * kWaves + wave;
      if (run < BRUN && !(kAbl4 & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)run), 16, vbB[i], 0, 0, 0);
    }
  };
  auto stage_g = [&](int rr, int slot) {
    const int yb = S2 * rr + py;
    // (the base may point below the blob for displacement rows that do not exist: those lanes are masked)
    const long long goff = WHICH == 0 ? (long long)(rr - i0 + R - 3) * D * plane : (long long)(i0 - rr + R) * D * plane + (long long)yb * g.W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(td_n + goff), 0, (unsigned)((long long)td_bytes - 4 * goff), 0x00020000);
    const unsigned dst = g_base + 4u * (unsigned)(slot * GBUF);
#pragma unroll
    for (int i = 0; i < GRPW; ++i) {
      const int run = i * kWaves + wave;
      if (run < GRUN && !(kAbl4 & 1)) {
        const int q = WHICH == 0 ? rr - i0 - gpi[i] : i0 + gpi[i] - rr;          // displacement row of this slab row
        const unsigned v = (gpi[i] >= 0 && q >= -R && q <= R) ? vbG[i] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)run), 16, v, 0, 0, 0);
      }
    }
  };
  const bool two_g = wave + kWaves < GRUN;          // this wave issues two G instructions per slab (waves 0 .. 3), else one

  stage_b(rlo, 0);
  stage_g(rlo, 0);
  if (rlo + 1 <= rhi) stage_g(rlo + 1, 1);
  int bslot = 0, gslot = 0;
  for (int r = rlo; r <= rhi; ++r) {
    // in flight, oldest first: [other map r, G r] issued two rows ago / in the prologue, then G r + 1 (youngest): the wait leaves only that
    if (r + 1 <= rhi) {
      if (two_g) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (r + 1 <= rhi) stage_b(r + 1, bslot ^ 1);
    // (order matters for the wait above: the other map's row first, the slab of row r + 2 last)
    if (r + 2 <= rhi) stage_g(r + 2, gslot == 0 ? 2 : gslot - 1);
    const float* sb = smem + bslot * BBUF + bAddr;
    const float* sg = smem + 2 * BBUF + gslot * GBUF + aAddr;
    float Gv[K::NB];
#pragma unroll
    for (int b = 0; b < K::NB; ++b) {
      const float v = sg[b * aStep];
      Gv[b] = (amask >> b) & 1 ? v : 0.f;
    }
    // 24 MFMAs, chunk-major; the other-map operand of MFMA k + LA is read behind MFMA k
    float bv[kNCH * K::NB];
#pragma unroll
    for (int k = 0; k < LA; ++k) bv[k] = sb[(k / K::NB) * kKC * BCS + 8 * (k % K::NB)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kNCH * K::NB; ++k) {
      const int c16 = k / K::NB, b = k % K::NB;
      if constexpr (!(kAbl4 & 4)) acc[c16][b & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Gv[b], bv[k], acc[c16][b & 1], 0, 0, 0);
      else acc[c16][b & 1][0] += Gv[b] + bv[k];
      if (k + LA < kNCH * K::NB) bv[k + LA] = sb[((k + LA) / K::NB) * kKC * BCS + 8 * ((k + LA) % K::NB)];
      __builtin_amdgcn_sched_barrier(0);
    }
    bslot ^= 1;
    gslot = gslot == 2 ? 0 : gslot + 1;
  }

  // ---- epilogue (as the earlier generations): accumulators -> LDS -> 128-byte rows
  __syncthreads();
  const float sumelems = (float)g.C;
  const bool pow2 = (g.C & (g.C - 1)) == 0;
  const float rcp = 1.0f / sumelems;
  const int opi = lane >> 4;
#pragma unroll
  for (int c = 0; c < kNCH; ++c) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float v = acc[c][0][rr] + acc[c][1][rr];
      smem[((c * kKC + ch) * 4 + opi) * K::XS + S2 * (4 * Jw + rr) + px] = pow2 ? v * rcp : v / sumelems;
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
      if (y < g.H && !(kAbl4 & 8)) out_n[(size_t)c * plane + (size_t)y * g.W + x] = smem[rowid * K::XS + xl];
    }
  }
}

template <int WHICH>
__global__ void __launch_bounds__(kThreads, 2)
corr_bwd_g4(const float* __restrict__ other, const float* __restrict__ top_diff, float* __restrict__ out, Args g) {
  const int L = (int)(blockIdx.x % 8) * g.GP + (int)(blockIdx.x / 8);
  if (L >= g.G) return;
  g4_body<WHICH>(other, top_diff, out, g, L);
}

// BOTH bottoms in one launch: task L of bottom 0 and task L of bottom 1 are neighbours in the grid (same XCD: they read the same rows of
// top_diff).  Two launches of 640 workgroups on 512 slots are two rounds each -- the second a quarter full; 1280 workgroups in one grid are
// 2.5 rounds, and the tail of one bottom's work overlaps the head of the other's (profiles/r06_corr_bwd_notes.md).  Same per-task code,
// same bits.
__global__ void __launch_bounds__(kThreads, 2)
corr_bwd_g4_both(const float* __restrict__ bottom0, const float* __restrict__ bottom1, const float* __restrict__ top_diff,
                 float* __restrict__ diff0, float* __restrict__ diff1, Args g) {
  const int L2 = (int)(blockIdx.x % 8) * (2 * g.GP) + (int)(blockIdx.x / 8);
  if (L2 >= 2 * g.G) return;
  if (L2 & 1) g4_body<1>(bottom0, top_diff, diff1, g, L2 >> 1);
  else g4_body<0>(bottom1, top_diff, diff0, g, L2 >> 1);
}

}  // namespace g4

int g_corr_bwd_gen = 0;           // test hook (fn2_debug_set_correlation_impl(6 / 15)): 2 / 3 = run generation 2 / 3 where generation 4 applies
int g_corr_bwd_separate = 0;      // test hook (fn2_debug_set_correlation_impl(16)): one launch per bottom where the merged launch applies
int g_corr_bwd_first_gen = 0;     // test hook (fn2_debug_set_correlation_impl(5)): run the first-generation kernel where both apply

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
  if constexpr (S2 == 2 && R == 10) {
    const bool aligned = cg.W % 4 == 0 && ((reinterpret_cast<uintptr_t>(other) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (aligned && !g_corr_bwd_first_gen && g_corr_bwd_gen == 0 && ((reinterpret_cast<uintptr_t>(top_diff)) & 15) == 0) {
      const size_t lds4 = sizeof(float) * g4::LDS_FLOATS;
      static bool attr4_set = false;
      if (!attr4_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&g4::corr_bwd_g4<WHICH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        attr4_set = true;
      }
      hipLaunchKernelGGL((g4::corr_bwd_g4<WHICH>), dim3(8 * g.GP), dim3(kThreads), lds4, st, other, top_diff, out, g);
      return check_launch("correlation_backward (mfma, G ring)");
    }
    if (aligned && !g_corr_bwd_first_gen && g_corr_bwd_gen != 2 && ((reinterpret_cast<uintptr_t>(top_diff)) & 15) == 0) {
      const size_t lds3 = sizeof(float) * g3::G<WHICH>::LDS_FLOATS;
      static bool attr3_set = false;
      if (!attr3_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&g3::corr_bwd_g3<WHICH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        attr3_set = true;
      }
      hipLaunchKernelGGL((g3::corr_bwd_g3<WHICH>), dim3(8 * g.GP), dim3(kThreads), lds3, st, other, top_diff, out, g);
      return check_launch("correlation_backward (mfma, G through LDS)");
    }
    if (aligned && !g_corr_bwd_first_gen) {
      const size_t lds2 = sizeof(float) * dma::LDS_FLOATS;
      static bool attr2_set = false;
      if (!attr2_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dma::corr_bwd_dma<WHICH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        attr2_set = true;
      }
      hipLaunchKernelGGL((dma::corr_bwd_dma<WHICH>), dim3(8 * g.GP), dim3(kThreads), lds2, st, other, top_diff, out, g);
      return check_launch("correlation_backward (mfma, LDS-DMA staging)");
    }
  }
  const size_t lds = sizeof(float) * K::LDS_FLOATS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_bwd_mfma<S2, R, WHICH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((corr_bwd_mfma<S2, R, WHICH>), dim3(8 * g.GP), dim3(kThreads), lds, st, other, top_diff, out, g);
  return check_launch("correlation_backward (mfma)");
}

// both bottoms of the FlowNetC instance in one grid (generation 4); FN2_ERR_UNSUPPORTED where it does not apply (the caller launches one by one)
static int launch_both(const CorrGeom& cg, const float* b0, const float* b1, const float* top_diff, float* d0, float* d1, hipStream_t st) {
  using K = Cfg<2, 10>;
  if (cg.s2 != 2 || cg.ngr != 10 || g_corr_bwd_first_gen || g_corr_bwd_gen != 0 || g_corr_bwd_separate) return FN2_ERR_UNSUPPORTED;
  const bool aligned = cg.W % 4 == 0 && ((reinterpret_cast<uintptr_t>(b0) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(d0) |
                                          reinterpret_cast<uintptr_t>(d1) | reinterpret_cast<uintptr_t>(top_diff)) & 15) == 0;
  if (!aligned) return FN2_ERR_UNSUPPORTED;
  Args g;
  g.N = cg.N; g.C = cg.C; g.H = cg.H; g.W = cg.W;
  const int Hc = (cg.H + 1) / 2, Wc = (cg.W + 1) / 2;
  g.NI = (Hc + 3) / 4;
  g.NSPAN = (Wc + K::SPANC - 1) / K::SPANC;
  g.NCQ = cg.C / kCQ;
  const long long G = (long long)cg.N * 2 * g.NI * g.NSPAN * g.NCQ;
  if (G > (1ll << 29)) return FN2_ERR_UNSUPPORTED;
  g.G = (int)G;
  g.GP = (g.G + 7) / 8;
  const size_t lds4 = sizeof(float) * g4::LDS_FLOATS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&g4::corr_bwd_g4_both), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
    attr_set = true;
  }
  hipLaunchKernelGGL(g4::corr_bwd_g4_both, dim3(16 * g.GP), dim3(kThreads), lds4, st, b0, b1, top_diff, d0, d1, g);
  return check_launch("correlation_backward (mfma, both bottoms)");
}

}  // namespace bwd

int corr_bwd_mfma_launch_both(const CorrGeom& g, const float* b0, const float* b1, const float* top_diff, float* d0, float* d1, hipStream_t st) {
  return bwd::launch_both(g, b0, b1, top_diff, d0, d1, st);
}

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
