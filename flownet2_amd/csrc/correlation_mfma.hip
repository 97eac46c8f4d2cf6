// Correlation forward fast paths for gfx950: kernel_size 1, stride_1 1, MULTIPLY, pad == max_displacement
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
// Task = (sample n, y-parity py, 4 class rows I, one N patch-row a, a 32-pixel x span).  The 4 + 4 image rows a
// task needs are staged through LDS in 8-channel chunks by LDS-DMA (buffer_load ... lds) straight from NCHW through
// a raw buffer descriptor whose out-of-range answer (0.0f) IS the reference's zero padding; ring of 3 slots, issue
// two chunks ahead, hand-counted vmcnt, one raw s_barrier per chunk.  The accumulators are scattered into an LDS
// image of the output and written back as full 128-byte rows.  Two kernels share this skeleton:
//   corr_fwd_pair<R>      stride_2 2, R 10, W % 4 == 0 (FlowNetC / FlowNet2): 4 waves, a wave owns both x parities of
//                         its patch, 16-byte DMA, ds_read_b64 operands (second half of this file);
//   corr_fwd_glds<S2,R>   everything else the MFMA path covers: 8 waves (parity x patch), dword DMA with the x
//                         parities de-interleaved on the source side, ds_read_b32 operands.
// (A first generation with register staging and ds_read_b128 operands ran 62 us where these run 51 / 42 us; DESIGN.md.)
//
// Scheduling: decode_task below (live tasks first, dead tasks last, lists balanced over the 8 XCDs).
#include <algorithm>
#include "correlation.hpp"

#include <type_traits>

namespace fn2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kKC = 16;   // channel granularity of the MFMA paths (two 8-channel chunks per loop trip)

constexpr int up_mod(int v, int r, int m) { return v + ((r - v % m) + m) % m; }   // smallest >= v with == r (mod m)
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
  static constexpr int BPX = JW * S2;                 // staged pixels per row of the second map
  static constexpr int XS = SPANPX + 1;               // output-image row stride in LDS
  static constexpr int OROWS = 16 * D;                // (mi, ni, o) rows of the output image
  static constexpr int LO_MAX = (NB >= 3) ? 2 : 0;    // specialised N-tile ranges [lo, hi], lo <= LO_MAX, hi >= HI_MIN
  static constexpr int HI_MIN = (NB >= 3) ? NB - 3 : NB - 1;
  static_assert(kWaves % S2 == 0, "waves must split evenly over x parities");
  static_assert(kThreads % SPANPX == 0, "store phase mapping");
};

int g_corr_ablation = 0;   // FN2_ABLATION builds only (profiling): bit 0 no MFMA, bit 1 no staging loads, bit 2 no stores
unsigned long long* g_corr_dbg = nullptr;   // FN2_ABLATION builds: per-workgroup {start, loop end, end, hw id} trace

struct MfmaArgs {
  int N, C, H, W;
  int NI, NSPAN;        // M patch rows per y-parity class, x spans
  int TH, TD;           // live / dead tasks per sample
  int LP, DP;           // live / dead list entries per XCD
  int ctot, c0;         // top blob: channels of the whole blob, first channel of the D*D slice this layer writes
  int relu; float slope;   // fused ReLU{negative_slope} on the way out
};

// Which patch column Jw the wave on SIMD s of a workgroup takes (corr_fwd_pair): code bits [2s, 2s + 2) = Jw.  N tiles outside the image
// are skipped, so the four waves of a task carry 4 / 5 / 6 / 6 tile units (full 32-pixel span) or 6 / 5 / 4 / 0 (the ragged last span of a
// 56-pixel row), and the three workgroups that share a CU -- list entries j, j + 32, j + 64 of an XCD: the dispatcher deals one workgroup
// to every CU of the XCD, then the next round -- stacked 17 units on one SIMD against 13.5 on average when every workgroup let wave w take
// column w (per-wave trace, scripts/probes/corr_wave_trace.py: all SIMDs of a CU end together, at the pace of the busiest).  The host picks
// the three bijections SIMD -> Jw of a CU's workgroups that minimise the busiest SIMD (15 here); a wave reads its SIMD from HW_ID.
struct SimdPlan {
  int on;
  unsigned char p[8][96];          // [XCD][entry of the XCD's range]; entries >= 96 start when a slot frees up: no plan
};
constexpr unsigned char kIdentityPlan = 0xE4;

// live N patch-rows of M patch-row I: a in [alo, ahi] (may be empty)
template <int S2, int R>
__host__ __device__ inline void live_range(int I, int Hc, int& alo, int& ahi) {
  constexpr int NB = Cfg<S2, R>::NB;
  // i2_0 = 4I - R + 4a;  live iff i2_0 + 3 >= 0 and i2_0 < Hc
  const int lo_num = R - 3 - 4 * I;                       // a >= lo_num / 4
  alo = lo_num <= 0 ? 0 : (lo_num + 3) / 4;
  const int hi_num = Hc - 1 + R - 4 * I;                  // a <= hi_num / 4
  ahi = hi_num < 0 ? -1 : hi_num / 4;
  if (ahi > NB - 1) ahi = NB - 1;
  if (4 * I >= Hc) { alo = 0; ahi = -1; }
}

// Task of one workgroup: (sample n, y parity py, M patch-row I, N patch-row a, x span).  live: a is in the live range
// [alo, ahi] of (py, I), i.e. its second-map rows touch the image; dead tasks only write zeros (12 % of the run
// time of a live one).
struct Task { int n, py, I, a, span, alo, ahi; bool live, valid; };

// Task lists.  All live tasks (sample-major; inside a sample the x span is the slowest index, so the workgroups
// that land on one CU mix full and ragged spans) and all dead tasks form two global lists, each cut into 8 equal
// contiguous ranges, one per XCD: block b runs on XCD b % 8 and takes entry b / 8 of that XCD's ranges, live
// entries first.  Long tasks first / short tasks last is the classic LPT order: the dead tasks fill the slots the
// first finished live workgroups free, and their stores overlap the remaining compute.  A contiguous range of
// one XCD is (part of) one sample, so the ~9x re-reads of a sample's rows stay in that XCD's L2.
// (Folding the zero fill into the live workgroups instead measured 8-15 % slower on every shape tried.)
template <int S2, int R>
__device__ __forceinline__ Task decode_task(const MfmaArgs& g) {
  using K = Cfg<S2, R>;
  Task k{0, 0, 0, 0, 0, 0, -1, false, false};
  const int xcd = (int)(blockIdx.x % 8), j = (int)(blockIdx.x / 8);
  int t, per_sample;
  k.live = j < g.LP;
  if (k.live) {
    t = xcd * g.LP + j; per_sample = g.TH;
    if (t >= g.N * g.TH) return k;
  } else {
    t = xcd * g.DP + (j - g.LP); per_sample = g.TD;
    if (t >= g.N * g.TD) return k;
  }
  k.n = t / per_sample;
  t %= per_sample;
  const int per_span = per_sample / g.NSPAN;
  k.span = t / per_span;
  t %= per_span;
  // t-th live (or dead) (py, I, a) combination, py-major / I / a order; <= S2 * NI scalar iterations
  for (int c = 0; c < S2 * g.NI; ++c) {
    const int cpy = c / g.NI, cI = c % g.NI;
    int alo, ahi;
    live_range<S2, R>(cI, (g.H - cpy + S2 - 1) / S2, alo, ahi);
    const int nlive = ahi >= alo ? ahi - alo + 1 : 0;
    const int cnt = k.live ? nlive : K::NB - nlive;
    if (t < cnt) {
      k.py = cpy; k.I = cI; k.alo = alo; k.ahi = ahi; k.valid = true;
      if (k.live) k.a = alo + t;
      else k.a = (nlive == 0 || t < alo) ? t : t + nlive;     // dead: a in [0, alo) U (ahi, NB)
      break;
    }
    t -= cnt;
  }
  return k;
}

// N tiles [lo, hi] of this wave that can be non-zero (x direction), as the index of one of the 3 x 3 specialised
// ranges (an extra tile only multiplies staged zeros); 9 = wave without a live M tile.
template <int S2, int R>
__device__ __forceinline__ int tile_range_sel(int jw, int Wc) {
  using K = Cfg<S2, R>;
  int lo = K::NB, hi = -1;
  if (jw < Wc) {
#pragma unroll
    for (int b = 0; b < K::NB; ++b) {
      const int j2 = jw - R + 4 * b;
      if (j2 + 3 >= 0 && j2 < Wc) { lo = min(lo, b); hi = max(hi, b); }
    }
  }
  int sel = 9;
  if (hi >= lo) sel = min(lo, K::LO_MAX) * 3 + (max(hi, K::HI_MIN) - K::HI_MIN);
  return __builtin_amdgcn_readfirstlane(sel);
}

// Epilogue: accumulators -> LDS image [mi][ni][o][x] -> coalesced 128-byte rows of top.
template <int S2, int R, bool STORE, typename Acc>
__device__ __forceinline__ void epilogue(const Acc& acc, float* smem, float* __restrict__ top, const MfmaArgs& g, const Task& k,
                                         int tid, int lane, int px, int Jw) {
  using K = Cfg<S2, R>;
  const int ni = (lane & 15) >> 2, nj = lane & 3;
  const float sumelems = (float)g.C;      // kernel_size^2 * channels, correlation_layer.cu:108
  const bool pow2 = (g.C & (g.C - 1)) == 0;   // x / 2^k == x * 2^-k exactly; otherwise keep the true division
  const float rcp = 1.0f / sumelems;
  const int mi = lane >> 4;               // C/D layout of 16x16 MFMA: row = 4*(lane>>4) + reg, col = lane & 15
#pragma unroll
  for (int b = 0; b < K::NB; ++b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {         // reg r <-> mj
      const int oo = 4 * b + nj - r;      // o + R
      if (oo >= 0 && oo < K::D)
      {
        float v = pow2 ? acc[b][r] * rcp : acc[b][r] / sumelems;
        if (g.relu) v = v > 0.f ? v : v * g.slope;
        smem[((mi * 4 + ni) * K::D + oo) * K::XS + S2 * (4 * Jw + r) + px] = v;
      }
    }
  }
  __syncthreads();
  // rowid = (rmi * 4 + rni) * D + oo  <->  top[n, (qq = 4a + rni - rmi, oo), y = S2 (4I + rmi) + py, 32-pixel span];
  // 16 rows per pass, (rmi, rni, oo) carried incrementally, 32-bit offsets inside the sample's output (buffer store).
  const int jS = K::SPANC * k.span, i0 = 4 * k.I;
  const int xl = tid % K::SPANPX;
  const int x = S2 * jS + xl;
  if (x < g.W && STORE) {
    const unsigned plane = (unsigned)g.H * (unsigned)g.W;
    const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(
        top + ((size_t)k.n * g.ctot + g.c0) * plane, 0, 4u * K::D * K::D * plane, 0x00020000);
    const unsigned hw4 = 4u * plane, w4 = 4u * (unsigned)g.W;
    constexpr int STEP = kThreads / K::SPANPX;
    int rowid = tid / K::SPANPX, rmi = 0, rni = 0, oo = rowid;
    while (oo >= K::D) { oo -= K::D; ++rni; }
#pragma unroll 1
    for (; rowid < K::OROWS; rowid += STEP) {
      const int qq = 4 * k.a + rni - rmi;   // q + R
      const int y = S2 * (i0 + rmi) + k.py;
      if (qq >= 0 && qq < K::D && y < g.H) {
        const unsigned off = (unsigned)(qq * K::D + oo) * hw4 + (unsigned)y * w4 + 4u * (unsigned)x;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, smem[rowid * K::XS + xl]), rsT, off, 0, 0);
      }
      oo += STEP;
      while (oo >= K::D) { oo -= K::D; ++rni; }
      while (rni >= 4) { rni -= 4; ++rmi; }
    }
  }
}

// =====================================================================================================
// General LDS-DMA kernel (8 waves, dword DMA).  The staged rows go global -> LDS directly (buffer_load_dword ... lds):
// no staging VGPRs, no ds_write pass, no "wait for loads, then write" phase in front of every barrier -- with three
// workgroups per CU running in lock step that phase had the matrix pipe idle.  An LDS-DMA instruction writes M0 + lane * 4, i.e.
// one 64-dword run per wave instruction, so the LDS image is built from runs:
//   second map: [channel group][channel][row 4][x parity][class column JW]  (a group = GC channels = a whole
//               number of runs; groups GPADG dwords apart), lane -> pixel de-interleaves the x parities on the
//               SOURCE side (lane l of a row's parity plane reads pixel S2 * column + parity);
//   first map:  per channel two runs (rows 0-1, rows 2-3), AHALF apart, channels CSA apart.
// Operand fetch is one ds_read_b32 per (tile, k-step): lane (kk, ni, nj) contracts channel 4r + kk in k-step r.
// The paddings make every ds_read_b32 hit 64 distinct banks (brute-forced for all four instantiations).
// Chunks are KC = 8 channels (2 k-steps), ring of 3 buffers, loads issued two chunks ahead, hand-counted vmcnt
// (hipcc would drain the queue) and one raw s_barrier per chunk.
constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }

template <int S2, int R>
struct GCfg {
  using K = Cfg<S2, R>;
  static constexpr int KC = 8;                        // channels per chunk
  static constexpr int NBUF = 3;
  static constexpr int ROWB = K::BPX;                 // dwords per staged second-map row: [px][jcol]
  static constexpr int CHB = 4 * ROWB;                // ... per channel
  static constexpr int GC = 64 / cgcd(64, CHB);       // channels per run-aligned group
  static constexpr int GROUP = GC * CHB;
  static constexpr int GS = GROUP + 4;                // group stride (pad 4: conflict-free ds_read_b32)
  static constexpr int NGRP = KC / GC;
  static constexpr int BSZ = NGRP * GS;
  static constexpr int CSA = 136, AHALF = 68;         // first map: channel stride, offset of the rows-2-3 run
  static constexpr int ASZ = KC * CSA;
  static constexpr int CHUNK = BSZ + ASZ;             // floats per staged chunk
  static constexpr int SLOT = CHUNK + 64;             // ring slot = chunk + one scratch run (dummy LDS-DMA destination)
  static constexpr int NBR = KC * CHB / 64;           // runs per chunk: second map
  static constexpr int NAR = KC * 2;                  //                 first map
  static constexpr int NRUN = NBR + NAR;
  static constexpr int RB = cdiv(NBR, kWaves), RA = cdiv(NAR, kWaves);
  static constexpr int RPW = RB + RA;                 // LDS-DMA instructions per wave and chunk (dummies included)
  static constexpr int LDS_FLOATS = cmax(NBUF * SLOT, K::OROWS * K::XS);
  static_assert(GC == 1 || GC == 2 || GC == 4, "channel 4r + kk must stay group-affine");
  static_assert(KC % GC == 0 && GROUP % 64 == 0, "a run never straddles a group");
  static_assert(K::SPANPX == 32, "first-map run = two 32-pixel rows");
  static_assert(RPW >= 1 && RPW <= 15, "vmcnt immediates below");
};

using lds_ptr_t = __attribute__((address_space(3))) void*;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 15, "");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int S2, int R, int LO, int HI, int ABL, typename Acc>
__device__ __forceinline__ void k_loop_glds(Acc& acc, float* smem, const float* a_n, const float* b_n, const MfmaArgs& g,
                                            unsigned lds_base, int lane, int wave, int px, int Jw, int py, int i0, int i2_0, int jS) {
  using K = Cfg<S2, R>;
  using G = GCfg<S2, R>;
  const int plane = g.H * g.W;
  constexpr unsigned OOB = 0x7ffffff0u;

  // ---- run plan of this wave.  Slot i < RB: second-map run i * 8 + wave, slot RB + i: first-map run i * 8 + wave.
  // The slot -> map assignment is static so the issue code is straight-line (it is interleaved with MFMAs below);
  // a slot past the end of its map is a dummy: out-of-range source (reads as 0) into the scratch run that ends every ring slot.
  unsigned voff[G::RPW];     // per-lane byte offset inside the chunk's channel range
  int ldst[G::RPW];          // LDS byte offset from the ring slot base; wave-uniform
#pragma unroll
  for (int i = 0; i < G::RB; ++i) {
    const int rho = i * kWaves + wave;
    voff[i] = OOB; ldst[i] = 4 * G::CHUNK;
    if (rho < G::NBR) {
      const int grp = (rho * 64) / G::GROUP;                   // uniform
      const int rem = rho * 64 - grp * G::GROUP + lane;
      const int chl = rem / G::CHB, rem2 = rem % G::CHB;
      const int row = rem2 / G::ROWB, col = rem2 % G::ROWB;
      const int pxx = col / K::JW, jc = col % K::JW;
      const int ib = i2_0 + row, yb = S2 * ib + py, xb = S2 * (jS - R + jc) + pxx;
      if (ib >= 0 && yb < g.H && xb >= 0 && xb < g.W) voff[i] = 4u * (unsigned)((grp * G::GC + chl) * plane + yb * g.W + xb);
      ldst[i] = 4 * (grp * G::GS + (rho * 64 - grp * G::GROUP));
    }
  }
#pragma unroll
  for (int i = 0; i < G::RA; ++i) {
    const int ra = i * kWaves + wave;
    voff[G::RB + i] = OOB; ldst[G::RB + i] = 4 * G::CHUNK;
    if (ra < G::NAR) {
      const int ch = ra >> 1, half = ra & 1;
      const int row = 2 * half + (lane >> 5), col = lane & 31;
      const int pxx = col / K::SPANC, jc = col % K::SPANC;
      const int ya = S2 * (i0 + row) + py, xa = S2 * (jS + jc) + pxx;
      if (ya < g.H && xa < g.W) voff[G::RB + i] = 4u * (unsigned)(ch * plane + ya * g.W + xa);
      ldst[G::RB + i] = 4 * (G::BSZ + ch * G::CSA + half * G::AHALF);
    }
  }
  const unsigned chunk_bytes = 4u * G::KC * (unsigned)plane;
  // Descriptors span the whole sample; the chunk is selected by the scalar offset.  OOB (2 GiB - 16) is beyond any
  // supported sample (C * H * W < 2^28 floats), so such lanes read 0.0f = the zero padding.
  const unsigned sample_bytes = 4u * (unsigned)g.C * (unsigned)plane;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_n), 0, sample_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_n), 0, sample_bytes, 0x00020000);

  // one LDS-DMA run: M0 = destination, 64 lanes x 4 B
  auto run = [&](int i, unsigned slot_bytes, unsigned soff) {
    lds_ptr_t lp = (lds_ptr_t)(uintptr_t)(slot_bytes + (unsigned)ldst[i]);
    if constexpr ((ABL & 2) != 0) return;      // profiling: no staging traffic
    if (i < G::RB) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, lp, 4, voff[i], soff, 0, 0);
    else           __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lp, 4, voff[i], soff, 0, 0);
  };

  // ---- operand addresses: lane (kk, ni, nj); k-step r contracts channel 4r + kk ----
  const int kk = lane >> 4, ni = (lane & 15) >> 2, nj = lane & 3;
  const int bAddr = (kk / G::GC) * G::GS + (kk % G::GC) * G::CHB + ni * G::ROWB + px * K::JW + 4 * Jw + nj;
  const int aAddr = G::BSZ + kk * G::CSA + (ni >> 1) * G::AHALF + (ni & 1) * 32 + px * K::SPANC + 4 * Jw + nj;
  constexpr int BSTEP = (4 / G::GC) * G::GS, ASTEP = 4 * G::CSA;     // k-step r -> + r * STEP
  constexpr int NT = (LO <= HI && !(ABL & 1)) ? HI - LO + 1 : 0;     // ABL bit 0 (profiling): no operand reads, no MFMAs
  constexpr int KS = G::KC / 4;
  static_assert(KS == 2, "the half-iteration schedule below is written for two k-steps per chunk");

  struct Ops { float a[KS]; float b[KS][NT > 0 ? NT : 1]; };
  auto read_ops = [&](Ops& o, const float* buf) {
    if constexpr (NT > 0) {
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        o.a[r] = buf[aAddr + r * ASTEP];
#pragma unroll
        for (int t = 0; t < NT; ++t) o.b[r][t] = buf[bAddr + r * BSTEP + 4 * (LO + t)];
      }
    }
  };
  auto mfma_step = [&](const Ops& o, int r, int t) {
    if constexpr (NT > 0) acc[LO + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[r], o.b[r][t], acc[LO + t], 0, 0, 0);
  };

  const int nchunks = g.C / G::KC;
  // One chunk: [k-step 0 MFMAs with the LDS-DMA issue of chunk c+2 threaded between them] -> wait until chunk c+1
  // has landed -> barrier -> [operand reads of chunk c+1] -> [k-step 1 MFMAs of chunk c hide the read latency].
  // sched_barrier(0) pins that order; hipcc otherwise hoists the issue code in front of the MFMAs.
  // ISSUE / READ are compile-time so the steady-state body is branch-free.
  auto chunk_step = [&](auto issue_tag, auto read_tag, int c, int slot, Ops& cur, Ops& nxt) {
    constexpr bool ISSUE = decltype(issue_tag)::value, READ = decltype(read_tag)::value;
    int s2 = slot + 2; if (s2 >= G::NBUF) s2 -= G::NBUF;
    int s1 = slot + 1; if (s1 >= G::NBUF) s1 -= G::NBUF;
    if constexpr (ISSUE) {
      const unsigned slot2_bytes = lds_base + 4u * (unsigned)(s2 * G::SLOT);
      const unsigned soff = (unsigned)(c + 2) * chunk_bytes;
      constexpr int STEPS = NT > G::RPW ? NT : G::RPW;
#pragma unroll
      for (int j = 0; j < STEPS; ++j) {
        if (j < NT) mfma_step(cur, 0, j);
        __builtin_amdgcn_sched_barrier(0);
        if (j < G::RPW) run(j, slot2_bytes, soff);
        __builtin_amdgcn_sched_barrier(0);
      }
      wait_vmcnt<G::RPW>();
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) mfma_step(cur, 0, j);
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (READ) read_ops(nxt, smem + s1 * G::SLOT);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) mfma_step(cur, 1, j);
    __builtin_amdgcn_sched_barrier(0);
  };
  using T = std::true_type;
  using F = std::false_type;

  // prologue: chunks 0 and 1 in flight, chunk 0 landed and published, its operands in registers
#pragma unroll
  for (int i = 0; i < G::RPW; ++i) run(i, lds_base, 0u);
#pragma unroll
  for (int i = 0; i < G::RPW; ++i) run(i, lds_base + 4u * G::SLOT, chunk_bytes);
  wait_vmcnt<G::RPW>();
  __builtin_amdgcn_s_barrier();
  Ops o0, o1;
  read_ops(o0, smem);
  int slot = 0, c = 0;
  for (; c + 2 < nchunks; c += 2) {            // nchunks is even and >= 2 (C % 16 == 0)
    chunk_step(T{}, T{}, c, slot, o0, o1);
    slot = slot + 1 == G::NBUF ? 0 : slot + 1;
    chunk_step(T{}, T{}, c + 1, slot, o1, o0);
    slot = slot + 1 == G::NBUF ? 0 : slot + 1;
  }
  chunk_step(F{}, T{}, c, slot, o0, o1);
  slot = slot + 1 == G::NBUF ? 0 : slot + 1;
  chunk_step(F{}, F{}, c + 1, slot, o1, o0);
}

template <int S2, int R, int ABL>
__global__ void __launch_bounds__(kThreads, 6)
corr_fwd_glds(const float* __restrict__ b0, const float* __restrict__ b1, float* __restrict__ top, MfmaArgs g,
              unsigned long long* __restrict__ dbg) {
  using K = Cfg<S2, R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef FN2_ABLATION
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
  const Task k = decode_task<S2, R>(g);
  if (!k.valid) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = wave % S2, Jw = wave / S2;
  const int i0 = 4 * k.I, jS = K::SPANC * k.span, jw = jS + 4 * Jw;
  const int Hc = (g.H - k.py + S2 - 1) / S2;
  const int Wc = (g.W - px + S2 - 1) / S2;
  if (i0 >= Hc) return;
  const size_t plane = (size_t)g.H * g.W;
  const float* a_n = b0 + (size_t)k.n * g.C * plane;
  const float* b_n = b1 + (size_t)k.n * g.C * plane;
  const int i2_0 = i0 - R + 4 * k.a;

  f32x4 acc[K::NB];
#pragma unroll
  for (int b = 0; b < K::NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;     // LDS byte address of the ring
  if (k.live) {
    const int sel = tile_range_sel<S2, R>(jw, Wc);
#define FN2_KLOOP(LO_, HI_) k_loop_glds<S2, R, LO_, HI_, ABL>(acc, smem, a_n, b_n, g, lds_base, lane, wave, px, Jw, k.py, i0, i2_0, jS)
    switch (sel) {
      case 0: FN2_KLOOP(0, K::HI_MIN + 0); break;
      case 1: FN2_KLOOP(0, K::HI_MIN + 1); break;
      case 2: FN2_KLOOP(0, K::HI_MIN + 2); break;
      case 3: FN2_KLOOP(K::LO_MAX / 2, K::HI_MIN + 0); break;
      case 4: FN2_KLOOP(K::LO_MAX / 2, K::HI_MIN + 1); break;
      case 5: FN2_KLOOP(K::LO_MAX / 2, K::HI_MIN + 2); break;
      case 6: FN2_KLOOP(K::LO_MAX, K::HI_MIN + 0); break;
      case 7: FN2_KLOOP(K::LO_MAX, K::HI_MIN + 1); break;
      case 8: FN2_KLOOP(K::LO_MAX, K::HI_MIN + 2); break;
      default: FN2_KLOOP(1, 0); break;
    }
#undef FN2_KLOOP
  }
#ifdef FN2_ABLATION
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
  epilogue<S2, R, !(ABL & 4)>(acc, smem, top, g, k, tid, lane, px, Jw);
#ifdef FN2_ABLATION
  if (dbg && threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[4 * blockIdx.x + 0] = t_start;
    dbg[4 * blockIdx.x + 1] = t_loop;
    dbg[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    dbg[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid | ((unsigned long long)(k.live ? 1 : 0) << 63);
    dbg[4 * 1024 + 4 * 8 * 1024 + 2 * blockIdx.x + 0] = rt_start;                             // 100 MHz wall clock
    dbg[4 * 1024 + 4 * 8 * 1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// =====================================================================================================
// Paired-parity variant (stride_2 = 2, R = 10, W % 4 == 0): the kernel FlowNetC / FlowNet2 run.
// A wave owns BOTH x parities of its 4x4 class patch (2 M tiles, 12 accumulator tiles), a workgroup is 4 waves.
// That makes the staged rows usable in their natural pixel order -- a lane's operand for the two parities is the
// 8-byte pair (x, x+1) -- so
//   * the rows arrive by 16-byte LDS-DMA (buffer_load_dwordx4 ... lds), 13 wave instructions per 8-channel chunk
//     instead of 52 dword ones: the dword kernel spends as long filling LDS (~10 cycles per wave instruction in
//     the texture path) as it spends in the matrix pipe;
//   * one ds_read_b64 feeds two MFMAs (7 reads per 12 MFMAs instead of 14), conflict-free: ds_read_b64 is
//     serviced in two 32-lane halves over 64 banks; in a half, lane group kk = 0/1 takes bank bit 5 (channel
//     stride 288 = 32 mod 64 dwords) and the 4 x 4 positions spread over rows (72 dwords = 8 mod 64) x pairs;
//   * twice the MFMA run (24) per barrier, half the waves to synchronise.
// First-map rows are only 32 dwords, so their 16-byte slots are XOR-swizzled (row bit 1 -> slot bit 2, channel
// bit 0 -> slot bit 1) on the SOURCE side of the DMA and in the read address alike.
template <int R>
struct HCfg {
  using K = Cfg<2, R>;
  static constexpr int WAVES = 4, THREADS = 256;
  static constexpr int KC = 8, NBUF = 3;
  static constexpr int BQ = K::BPX / 4;               // 16-byte slots per staged second-map row (18)
  static constexpr int AQ = K::SPANPX / 4;            // ... first-map row (8)
  static constexpr int CHB = 4 * K::BPX;              // dwords per channel, second map (288)
  static constexpr int CHA = 4 * K::SPANPX;           //                     first map (128)
  static constexpr int BSZ = KC * CHB, ASZ = KC * CHA;
  static constexpr int CHUNK = BSZ + ASZ;             // floats per ring slot (3328)
  static constexpr int NBR = BSZ / 256, NAR = ASZ / 256, NRUN = NBR + NAR;     // 1 KiB runs per chunk: 9 + 4
  static constexpr int RPW = cdiv(NRUN, WAVES);       // runs per wave: RPW for waves < NRUN % WAVES (or all), else RPW - 1
  static constexpr int NFULL = NRUN % WAVES == 0 ? WAVES : NRUN % WAVES;
  static constexpr int ROWTAB = K::OROWS * K::XS;      // behind the output image: byte offset of every image row in top (epilogue)
  static constexpr int LDS_FLOATS = cmax(NBUF * CHUNK, ROWTAB + K::OROWS);
  static_assert(R == 10, "bank analysis above is for 72-pixel rows");
  static_assert(BSZ % 256 == 0 && ASZ % 256 == 0 && K::BPX % 4 == 0, "whole 1 KiB runs of 16-byte slots");
  static_assert(CHB % 64 == 32, "channel stride must flip bank bit 5");
};

// PROJ (profiling only, wrong results): 0 = the kernel; 1 = only 3 of every 8 MFMAs issue -- the matrix-pipe time a split-bf16
// (bf16 x 3, 6 products per fp32 product on v_mfma_f32_16x16x16_bf16: 48 cycles per 16 channels against 128) variant would have
// on the same staging, LDS traffic and epilogue, with the operand split taken as free; 2 = no MFMA at all (the data-movement
// floor of this structure); 3 = all the MFMAs, every second LDS-DMA run (half the staging traffic: 46.3 us against 47.3, i.e. the
// staging volume is NOT what holds this kernel back).  The operands are kept alive by empty asm statements, so the LDS reads stay.
template <int R, int LO, int HI, int PROJ = 0, typename Acc>
__device__ __forceinline__ void k_loop_pair(Acc& acc0, Acc& acc1, float* smem, const float* a_n, const float* b_n, const MfmaArgs& g,
                                            unsigned lds_base, int lane, int wave, int Jw, int py, int i0, int i2_0, int jS) {
  using K = Cfg<2, R>;
  using H = HCfg<R>;
  const int plane = g.H * g.W;
  constexpr unsigned OOB = 0x7ffffff0u;

  // ---- run plan: run rho = i * WAVES + wave; lane -> one 16-byte slot (4 pixels of one row and channel) ----
  unsigned voff[H::RPW];
  int ldst[H::RPW];
  bool isB[H::RPW];
#pragma unroll
  for (int i = 0; i < H::RPW; ++i) {
    const int rho = i * H::WAVES + wave;
    voff[i] = OOB; ldst[i] = 0; isB[i] = rho < H::NBR;
    if (rho < H::NBR) {
      const int sl = rho * 64 + lane;                        // slot index in [ch][row][group]
      const int ch = sl / (4 * H::BQ), rem = sl % (4 * H::BQ);
      const int row = rem / H::BQ, gq = rem % H::BQ;
      const int ib = i2_0 + row, yb = 2 * ib + py, xb = 2 * (jS - R) + 4 * gq;
      if (ib >= 0 && yb < g.H && xb >= 0 && xb < g.W) voff[i] = 4u * (unsigned)(ch * plane + yb * g.W + xb);
      ldst[i] = rho * 1024;
    } else if (rho < H::NRUN) {
      const int sl = (rho - H::NBR) * 64 + lane;             // slot index in [ch][row][swizzled group]
      const int ch = sl / (4 * H::AQ), rem = sl % (4 * H::AQ);
      const int row = rem / H::AQ, gs = rem % H::AQ;
      const int gq = gs ^ ((row >> 1) << 2) ^ ((ch & 1) << 1);
      const int ya = 2 * (i0 + row) + py, xa = 2 * jS + 4 * gq;
      if (ya < g.H && xa < g.W) voff[i] = 4u * (unsigned)(ch * plane + ya * g.W + xa);
      ldst[i] = 4 * H::BSZ + (rho - H::NBR) * 1024;
    }
  }
  const bool full = wave < H::NFULL;                         // this wave issues RPW (else RPW - 1) runs per chunk
  const unsigned chunk_bytes = 4u * H::KC * (unsigned)plane;
  const unsigned sample_bytes = 4u * (unsigned)g.C * (unsigned)plane;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_n), 0, sample_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_n), 0, sample_bytes, 0x00020000);
  auto run = [&](int i, unsigned slot_bytes, unsigned soff) {
    if (i == H::RPW - 1 && !full) return;                    // wave-uniform
    if constexpr (PROJ == 3) { if (i & 1) return; }          // profiling: half the staging traffic, all the MFMAs
    lds_ptr_t lp = (lds_ptr_t)(uintptr_t)(slot_bytes + (unsigned)ldst[i]);
    if (isB[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, lp, 16, voff[i], soff, 0, 0);
    else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lp, 16, voff[i], soff, 0, 0);
  };
  auto wait_landed = [&]() {                                  // everything but the newest chunk's runs has landed
    if constexpr (PROJ == 3) { wait_vmcnt<H::RPW / 2>(); return; }
    if (full) wait_vmcnt<H::RPW>(); else wait_vmcnt<H::RPW - 1>();
  };

  // ---- operand addresses (dwords): lane (kk, ni, nj); k-step r contracts channel 4r + kk ----
  const int kk = lane >> 4, ni = (lane & 15) >> 2, nj = lane & 3;
  const int bAddr = kk * H::CHB + ni * K::BPX + 2 * (4 * Jw + nj);
  const int gA = (2 * Jw + (nj >> 1)) ^ ((ni >> 1) << 2) ^ ((kk & 1) << 1);
  const int aAddr = H::BSZ + kk * H::CHA + ni * K::SPANPX + 4 * gA + 2 * (nj & 1);
  constexpr int BSTEP = 4 * H::CHB, ASTEP = 4 * H::CHA;
  constexpr int NT = (LO <= HI) ? HI - LO + 1 : 0;
  constexpr int KS = H::KC / 4;
  static_assert(KS == 2, "half-iteration schedule below");
  using f32x2 = __attribute__((ext_vector_type(2))) float;

  struct Ops { f32x2 a[KS]; f32x2 b[KS][NT > 0 ? NT : 1]; };
  auto read_ops = [&](Ops& o, const float* buf) {
    if constexpr (NT > 0) {
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        o.a[r] = *reinterpret_cast<const f32x2*>(buf + aAddr + r * ASTEP);
#pragma unroll
        for (int t = 0; t < NT; ++t) o.b[r][t] = *reinterpret_cast<const f32x2*>(buf + bAddr + r * BSTEP + 8 * (LO + t));
      }
    }
  };
  // step j of k-step r: tile j / 2, parity j % 2
  auto mfma_step = [&](const Ops& o, int r, int j) {
    if constexpr (NT > 0) {
      const int t = j >> 1;
      if constexpr (PROJ != 0) {
        if (PROJ == 2 || (PROJ == 1 && (j + 2 * NT * r) % 8 >= 3)) {
          asm volatile("" ::"v"(o.a[r].x), "v"(o.a[r].y), "v"(o.b[r][t].x), "v"(o.b[r][t].y));
          return;
        }
      }
      if (j & 1) acc1[LO + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[r].y, o.b[r][t].y, acc1[LO + t], 0, 0, 0);
      else       acc0[LO + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[r].x, o.b[r][t].x, acc0[LO + t], 0, 0, 0);
    }
  };

  const int nchunks = g.C / H::KC;
  auto chunk_step = [&](auto issue_tag, auto read_tag, int c, int slot, Ops& cur, Ops& nxt) {
    constexpr bool ISSUE = decltype(issue_tag)::value, READ = decltype(read_tag)::value;
    int s2 = slot + 2; if (s2 >= H::NBUF) s2 -= H::NBUF;
    int s1 = slot + 1; if (s1 >= H::NBUF) s1 -= H::NBUF;
    if constexpr (ISSUE) {
      const unsigned slot2_bytes = lds_base + 4u * (unsigned)(s2 * H::CHUNK);
      const unsigned soff = (unsigned)(c + 2) * chunk_bytes;
      constexpr int GAP = (2 * NT) / (H::RPW + 1) > 0 ? (2 * NT) / (H::RPW + 1) : 1;     // MFMAs between two DMA issues
      int issued = 0;
#pragma unroll
      for (int j = 0; j < 2 * NT; ++j) {
        mfma_step(cur, 0, j);
        if ((j + 1) % GAP == 0 && issued < H::RPW) {
          __builtin_amdgcn_sched_barrier(0);
          run(issued++, slot2_bytes, soff);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int i = 0; i < H::RPW; ++i) if (i >= issued) run(i, slot2_bytes, soff);       // NT == 0 or tiny: issue the rest
      __builtin_amdgcn_sched_barrier(0);
      wait_landed();
    } else {
#pragma unroll
      for (int j = 0; j < 2 * NT; ++j) mfma_step(cur, 0, j);
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (READ && NT > 0 && PROJ == 0) {
      // The operand reads of chunk c + 1 go BETWEEN the MFMAs of k-step 1, one group (the two first-map operands, then two second-map
      // tiles at a time) behind each of the first MFMAs.  Issued in one block in front of them (rounds 1-2) they are ~100 cycles in which
      // this wave feeds nothing to the matrix pipe: hidden while three waves share the SIMD, exposed once the workgroups of a CU have
      // drifted apart (per-workgroup trace at FlowNetC's shape: the three loops of a CU end at 48k / 68k / 88k cycles).  -1 us of 47.
      const float* buf = smem + s1 * H::CHUNK;
      constexpr int GROUPS = 1 + KS * ((NT + 1) / 2);
      static_assert(GROUPS <= 2 * NT, "one read group per MFMA");
#pragma unroll
      for (int j = 0; j < 2 * NT; ++j) {
        mfma_step(cur, 1, j);
        if (j < GROUPS) {
          __builtin_amdgcn_sched_barrier(0);
          if (j == 0) {
#pragma unroll
            for (int r = 0; r < KS; ++r) nxt.a[r] = *reinterpret_cast<const f32x2*>(buf + aAddr + r * ASTEP);
          } else {
            const int r = (j - 1) / ((NT + 1) / 2), t0 = 2 * ((j - 1) % ((NT + 1) / 2));
#pragma unroll
            for (int t = t0; t < t0 + 2 && t < NT; ++t) nxt.b[r][t] = *reinterpret_cast<const f32x2*>(buf + bAddr + r * BSTEP + 8 * (LO + t));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      if constexpr (READ) read_ops(nxt, smem + s1 * H::CHUNK);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2 * NT; ++j) mfma_step(cur, 1, j);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using T = std::true_type;
  using F = std::false_type;

#pragma unroll
  for (int i = 0; i < H::RPW; ++i) run(i, lds_base, 0u);
#pragma unroll
  for (int i = 0; i < H::RPW; ++i) run(i, lds_base + 4u * H::CHUNK, chunk_bytes);
  wait_landed();
  __builtin_amdgcn_s_barrier();
  Ops o0, o1;
  read_ops(o0, smem);
  int slot = 0, c = 0;
  for (; c + 2 < nchunks; c += 2) {            // nchunks is even and >= 2 (C % 16 == 0)
    chunk_step(T{}, T{}, c, slot, o0, o1);
    slot = slot + 1 == H::NBUF ? 0 : slot + 1;
    chunk_step(T{}, T{}, c + 1, slot, o1, o0);
    slot = slot + 1 == H::NBUF ? 0 : slot + 1;
  }
  chunk_step(F{}, T{}, c, slot, o0, o1);
  slot = slot + 1 == H::NBUF ? 0 : slot + 1;
  chunk_step(F{}, F{}, c + 1, slot, o1, o0);
}

template <int R, int PROJ = 0>
__global__ void __launch_bounds__(256, 3)
corr_fwd_pair(const float* __restrict__ b0, const float* __restrict__ b1, float* __restrict__ top, MfmaArgs g,
              unsigned long long* __restrict__ dbg, SimdPlan plan) {
  using K = Cfg<2, R>;
  using H = HCfg<R>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef FN2_ABLATION
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
  const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif
  const Task k = decode_task<2, R>(g);
  if (!k.valid) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int Jw = wave;
  if (k.live && plan.on && blockIdx.x < 8 * 96) {          // SimdPlan: the wave on SIMD s takes the patch column the host planned for it
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = (int)((hwid >> 4) & 3u);
    int* note = reinterpret_cast<int*>(smem + H::NBUF * H::CHUNK);      // behind the ring; the output image reaches it only in the epilogue
    static_assert(H::NBUF * H::CHUNK + 4 <= H::LDS_FLOATS, "room for the four SIMD notes");
    if (lane == 0) note[wave] = simd;
    __syncthreads();
    const int seen = (1 << note[0]) | (1 << note[1]) | (1 << note[2]) | (1 << note[3]);
    if (seen == 15)                                          // the four waves sit on four SIMDs (else: wave w keeps column w)
      Jw = __builtin_amdgcn_readfirstlane((plan.p[blockIdx.x % 8][blockIdx.x / 8] >> (2 * simd)) & 3);
  }
  const int i0 = 4 * k.I, jS = K::SPANC * k.span, jw = jS + 4 * Jw;
  const int Hc = (g.H - k.py + 1) / 2;
  const int Wc = g.W / 2;                          // W % 4 == 0: both x parities have W / 2 class columns
  if (i0 >= Hc) return;
  const size_t plane = (size_t)g.H * g.W;
  const float* a_n = b0 + (size_t)k.n * g.C * plane;
  const float* b_n = b1 + (size_t)k.n * g.C * plane;
  const int i2_0 = i0 - R + 4 * k.a;
  const size_t top_n = (size_t)k.n * g.ctot + g.c0;

  // Output rows of this task: rowid = (rmi * 4 + rni) * D + oo  <->  top[n, (qq = 4a + rni - rmi, oo), y = 2 (4I + rmi) + py,
  // 32-pixel span].  8 threads x 16 bytes per row, 32 rows per pass; offsets are 32-bit inside the sample's output (buffer store).
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(
      top + top_n * plane, 0, (unsigned)(4u * K::D * K::D * (unsigned)plane), 0x00020000);
  // Dead tasks write zeros for the rows that exist; live tasks look the row's offset up in a table
  // the workgroup builds once (the decode, its range tests and the offset arithmetic cost ~25 VALU instructions per row and thread before --
  // VALU time is matrix-pipe time for the workgroups still in their K loops, and the epilogue + DMA plan were 1.4 VALU instructions per MFMA
  // of the whole launch, profiles/r03_corr_stall_counters.txt).
  constexpr unsigned NOROW = 0xffffffffu;
  const unsigned hw4 = 4u * (unsigned)plane, w4 = 4u * (unsigned)g.W;
  auto row_offset = [&](int rowid) -> unsigned {              // rowid = (rmi * 4 + rni) * D + oo
    const int blk = rowid / K::D, oo = rowid - blk * K::D, rmi = blk >> 2, rni = blk & 3;
    const int qq = 4 * k.a + rni - rmi, y = 2 * (i0 + rmi) + k.py;
    return (qq >= 0 && qq < K::D && y < g.H) ? (unsigned)(qq * K::D + oo) * hw4 + (unsigned)y * w4 : NOROW;
  };
  auto store_zero_rows = [&]() {
    const int xq = tid & 7, x = 2 * jS + 4 * xq;
    if (x >= g.W) return;                                   // W % 4 == 0: a quad is inside or outside as a whole
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int rowid = tid >> 3; rowid < K::OROWS; rowid += H::THREADS / 8) {
      const unsigned off = row_offset(rowid);
      if (off != NOROW)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, z), rsT, off + 4u * (unsigned)x, 0, 0);
    }
  };
  if (!k.live) {          // dead task: zeros, no LDS round trip
    store_zero_rows();
    return;
  }

  f32x4 acc0[K::NB], acc1[K::NB];
#pragma unroll
  for (int b = 0; b < K::NB; ++b) { acc0[b] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  [[maybe_unused]] int wave_sel = 9;
  {
    const int sel = tile_range_sel<2, R>(jw, Wc);
    wave_sel = sel;
#define FN2_KLOOP(LO_, HI_) k_loop_pair<R, LO_, HI_, PROJ>(acc0, acc1, smem, a_n, b_n, g, lds_base, lane, wave, Jw, k.py, i0, i2_0, jS)
    switch (sel) {
      case 0: FN2_KLOOP(0, K::HI_MIN + 0); break;
      case 1: FN2_KLOOP(0, K::HI_MIN + 1); break;
      case 2: FN2_KLOOP(0, K::HI_MIN + 2); break;
      case 3: FN2_KLOOP(K::LO_MAX / 2, K::HI_MIN + 0); break;
      case 4: FN2_KLOOP(K::LO_MAX / 2, K::HI_MIN + 1); break;
      case 5: FN2_KLOOP(K::LO_MAX / 2, K::HI_MIN + 2); break;
      case 6: FN2_KLOOP(K::LO_MAX, K::HI_MIN + 0); break;
      case 7: FN2_KLOOP(K::LO_MAX, K::HI_MIN + 1); break;
      case 8: FN2_KLOOP(K::LO_MAX, K::HI_MIN + 2); break;
      default: FN2_KLOOP(1, 0); break;
    }
#undef FN2_KLOOP
  }
#ifdef FN2_ABLATION
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: accumulators -> LDS image [mi][ni][o][x] -> coalesced 128-byte rows of top ----
  {
    const int ni = (lane & 15) >> 2, nj = lane & 3, mi = lane >> 4;
    const float sumelems = (float)g.C;
    const bool pow2 = (g.C & (g.C - 1)) == 0;
    const float rcp = 1.0f / sumelems;
    // 1 / C: exact for power-of-two channel counts (x / 2^k == x * 2^-k) and applied to the rows on their way out, together with the
    // ReLU; otherwise the reference's true division, here
    if (!pow2) {
#pragma unroll
      for (int b = 0; b < K::NB; ++b) { acc0[b] /= sumelems; acc1[b] /= sumelems; }
    }
#pragma unroll
    for (int b = 0; b < K::NB; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oo = 4 * b + nj - r;
        if (oo >= 0 && oo < K::D) {
          float* dst = smem + ((mi * 4 + ni) * K::D + oo) * K::XS + 2 * (4 * Jw + r);
          dst[0] = acc0[b][r];
          dst[1] = acc1[b][r];
        }
      }
    }
    unsigned* rowtab = reinterpret_cast<unsigned*>(smem + H::ROWTAB);
    for (int rowid = tid; rowid < K::OROWS; rowid += H::THREADS) rowtab[rowid] = row_offset(rowid);
    __syncthreads();
    const int xq = tid & 7, x = 2 * jS + 4 * xq;
    if (x < g.W) {
      const float scale = pow2 ? rcp : 1.0f;
      const float slope = g.slope;
      const bool relu = g.relu != 0;
#pragma unroll 1
      for (int rowid = tid >> 3; rowid < K::OROWS; rowid += H::THREADS / 8) {
        const unsigned off = rowtab[rowid];
        if (off == NOROW) continue;
        const float* src = smem + rowid * K::XS + 4 * xq;
        f32x4 v = f32x4{src[0], src[1], src[2], src[3]} * scale;
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsT, off + 4u * (unsigned)x, 0, 0);
      }
    }
  }
#ifdef FN2_ABLATION
  if (dbg && lane == 0 && blockIdx.x < 1024) {      // per wave (scripts/probes/corr_wave_trace.py): start, loop end, end, {HW_ID, tile-range selector, Jw}
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned long long* w = dbg + 4 * 1024 + 4 * (8 * blockIdx.x + Jw);
    w[0] = t_start;
    w[1] = t_loop;
    w[2] = __builtin_amdgcn_s_memtime();
    w[3] = hwid | ((unsigned long long)wave_sel << 32) | ((unsigned long long)Jw << 40) | (1ull << 63);
  }
  if (dbg && threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[4 * blockIdx.x + 0] = t_start;
    dbg[4 * blockIdx.x + 1] = t_loop;
    dbg[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    dbg[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid | (1ull << 63);
    dbg[4 * 1024 + 4 * 8 * 1024 + 2 * blockIdx.x + 0] = rt_start;
    dbg[4 * 1024 + 4 * 8 * 1024 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

int g_corr_skip_dead = 0;     // profiling hook (fn2_debug_set_correlation_impl(14)): launch no zero-fill workgroups
int g_corr_simd_plan = 1;      // profiling / test hook (fn2_debug_set_correlation_impl(13) switches the SIMD plan of corr_fwd_pair off)

// Tile units of wave Jw of a span (the specialised N-tile ranges of tile_range_sel).
template <int R>
static int pair_tiles(int span, int Jw, int Wc) {
  using K = Cfg<2, R>;
  const int jw = K::SPANC * span + 4 * Jw;
  if (jw >= Wc) return 0;
  int lo = K::NB, hi = -1;
  for (int b = 0; b < K::NB; ++b) {
    const int j2 = jw - R + 4 * b;
    if (j2 + 3 >= 0 && j2 < Wc) { lo = std::min(lo, b); hi = std::max(hi, b); }
  }
  if (hi < lo) return 0;
  return std::max(hi, K::HI_MIN) - std::min(lo, K::LO_MAX) + 1;
}

template <int R>
static void build_simd_plan(const MfmaArgs& g, SimdPlan& plan) {
  static const int perms[24][4] = {{0,1,2,3},{0,1,3,2},{0,2,1,3},{0,2,3,1},{0,3,1,2},{0,3,2,1},{1,0,2,3},{1,0,3,2},{1,2,0,3},{1,2,3,0},{1,3,0,2},{1,3,2,0},
                                   {2,0,1,3},{2,0,3,1},{2,1,0,3},{2,1,3,0},{2,3,0,1},{2,3,1,0},{3,0,1,2},{3,0,2,1},{3,1,0,2},{3,1,2,0},{3,2,0,1},{3,2,1,0}};
  auto code = [](const int* pm) { return (unsigned char)(pm[0] | (pm[1] << 2) | (pm[2] << 4) | (pm[3] << 6)); };
  const int Wc = g.W / 2, per_span = g.TH / g.NSPAN;
  // the plan is a function of (N, H, W) only: the 147k-step search below runs once per geometry, not once per launch
  struct Cached { int N, H, W; SimdPlan plan; };
  static thread_local Cached cache[4] = {};
  static thread_local int next = 0;
  for (const Cached& c : cache)
    if (c.N == g.N && c.H == g.H && c.W == g.W && c.N > 0) { plan = c.plan; plan.on = g_corr_simd_plan; return; }
  plan.on = g_corr_simd_plan;
  for (int x = 0; x < 8; ++x)
    for (int j = 0; j < 32; ++j) {
      int tv[3][4];
      for (int s = 0; s < 3; ++s) {
        const int e = j + 32 * s;
        const long long t = (long long)x * g.LP + e;
        const bool live = e < g.LP && t < (long long)g.N * g.TH;
        const int span = live ? (int)((t % g.TH) / per_span) : 0;
        for (int w = 0; w < 4; ++w) tv[s][w] = live ? pair_tiles<R>(span, w, Wc) : 0;
      }
      int best = 1 << 30, bb = 0, bc = 0;
      for (int b = 0; b < 24; ++b)
        for (int c = 0; c < 24; ++c) {
          int mx = 0, sq = 0;
          for (int sd = 0; sd < 4; ++sd) {
            const int l = tv[0][sd] + tv[1][perms[b][sd]] + tv[2][perms[c][sd]];
            mx = std::max(mx, l); sq += l * l;
          }
          const int cost = mx * 4096 + sq;          // busiest SIMD first, then the spread
          if (cost < best) { best = cost; bb = b; bc = c; }
        }
      plan.p[x][j] = kIdentityPlan;
      plan.p[x][j + 32] = code(perms[bb]);
      plan.p[x][j + 64] = code(perms[bc]);
    }
  cache[next] = Cached{g.N, g.H, g.W, plan};
  next = (next + 1) % 4;
}

int g_corr_force_dword = 0;   // test hook: run the general (dword LDS-DMA) kernel even where the paired one applies
int g_corr_proj = 0;          // profiling hook (fn2_debug_set_correlation_impl(7 / 8)): the PROJ = 1 / 2 builds of corr_fwd_pair

template <int S2, int R>
static int launch(const CorrGeom& cg, const float* b0, const float* b1, float* top, hipStream_t st) {
  using K = Cfg<S2, R>;
  MfmaArgs g;
  g.N = cg.N; g.C = cg.C; g.H = cg.H; g.W = cg.W;
  g.ctot = cg.top_ctot; g.c0 = cg.top_c0; g.relu = cg.relu; g.slope = cg.slope;
  const int Hc = (cg.H + S2 - 1) / S2, Wc = (cg.W + S2 - 1) / S2;
  g.NI = (Hc + 3) / 4;
  g.NSPAN = (Wc + K::SPANC - 1) / K::SPANC;
  int nlive = 0;
  for (int py = 0; py < S2; ++py)
    for (int I = 0; I < g.NI; ++I) {
      int alo, ahi;
      live_range<S2, R>(I, (cg.H - py + S2 - 1) / S2, alo, ahi);
      if (ahi >= alo) nlive += ahi - alo + 1;
    }
  g.TH = nlive * g.NSPAN;
  g.TD = S2 * g.NI * K::NB * g.NSPAN - g.TH;
  const long long NL = (long long)cg.N * g.TH, ND = (long long)cg.N * g.TD;
  if (NL + ND > (1ll << 30)) return fail(FN2_ERR_UNSUPPORTED, "correlation: problem too large for the MFMA path");
  g.LP = (int)((NL + 7) / 8);
  g.DP = (int)((ND + 7) / 8);
  if (g_corr_skip_dead) g.DP = 0;          // profiling only (wrong output): what the zero-fill workgroups cost
  const unsigned grid = 8u * (unsigned)(g.LP + g.DP);
  if constexpr (S2 == 2 && R == 10) {
    const bool aligned = cg.W % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(b0) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(top)) & 15) == 0;
    if (!g_corr_force_dword && !g_corr_ablation && aligned) {
      const size_t lds3 = sizeof(float) * HCfg<R>::LDS_FLOATS;
      SimdPlan plan;
      build_simd_plan<R>(g, plan);
      static bool attr3_set = false;
      if (!attr3_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_pair<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        attr3_set = true;
      }
      if (g_corr_proj == 3) {
        static bool attrq_set = false;
        if (!attrq_set) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_pair<R, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
          attrq_set = true;
        }
        hipLaunchKernelGGL((corr_fwd_pair<R, 3>), dim3(grid), dim3(HCfg<R>::THREADS), lds3, st, b0, b1, top, g, g_corr_dbg, plan);
        return check_launch("correlation_forward (mfma, paired parities, projection build)");
      }
      if (g_corr_proj == 1 || g_corr_proj == 2) {
        static bool attrp_set = false;
        if (!attrp_set) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_pair<R, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_pair<R, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
          attrp_set = true;
        }
        if (g_corr_proj == 1) hipLaunchKernelGGL((corr_fwd_pair<R, 1>), dim3(grid), dim3(HCfg<R>::THREADS), lds3, st, b0, b1, top, g, g_corr_dbg, plan);
        else                  hipLaunchKernelGGL((corr_fwd_pair<R, 2>), dim3(grid), dim3(HCfg<R>::THREADS), lds3, st, b0, b1, top, g, g_corr_dbg, plan);
        return check_launch("correlation_forward (mfma, paired parities, projection build)");
      }
      hipLaunchKernelGGL((corr_fwd_pair<R>), dim3(grid), dim3(HCfg<R>::THREADS), lds3, st, b0, b1, top, g, g_corr_dbg, plan);
      return check_launch("correlation_forward (mfma, paired parities)");
    }
  }
  const size_t lds2 = sizeof(float) * GCfg<S2, R>::LDS_FLOATS;
  static bool attr2_set = false;
  if (!attr2_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_glds<S2, R, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    attr2_set = true;
  }
#ifdef FN2_ABLATION
  if (S2 == 2 && R == 10 && g_corr_ablation) {     // profiling builds, fn2_debug_set_correlation_impl(64 + bits)
    switch (g_corr_ablation) {
      case 1: hipLaunchKernelGGL((corr_fwd_glds<2, 10, 1>), dim3(grid), dim3(kThreads), lds2, st, b0, b1, top, g, g_corr_dbg); break;
      case 2: hipLaunchKernelGGL((corr_fwd_glds<2, 10, 2>), dim3(grid), dim3(kThreads), lds2, st, b0, b1, top, g, g_corr_dbg); break;
      case 3: hipLaunchKernelGGL((corr_fwd_glds<2, 10, 3>), dim3(grid), dim3(kThreads), lds2, st, b0, b1, top, g, g_corr_dbg); break;
      case 4: hipLaunchKernelGGL((corr_fwd_glds<2, 10, 4>), dim3(grid), dim3(kThreads), lds2, st, b0, b1, top, g, g_corr_dbg); break;
      case 6: hipLaunchKernelGGL((corr_fwd_glds<2, 10, 6>), dim3(grid), dim3(kThreads), lds2, st, b0, b1, top, g, g_corr_dbg); break;
      default: hipLaunchKernelGGL((corr_fwd_glds<2, 10, 7>), dim3(grid), dim3(kThreads), lds2, st, b0, b1, top, g, g_corr_dbg); break;
    }
    return check_launch("correlation_forward (mfma, lds-dma, ablation)");
  }
#endif
  hipLaunchKernelGGL((corr_fwd_glds<S2, R, 0>), dim3(grid), dim3(kThreads), lds2, st, b0, b1, top, g, g_corr_dbg);
  return check_launch("correlation_forward (mfma, lds-dma)");
}

bool corr_fwd_mfma_supported(const CorrGeom& g) {
  if (g.K != 1 || g.s1 != 1 || g.type != FN2_CORR_MULTIPLY || g.pad != g.md) return false;
  if (g.C % kKC != 0) return false;
  if ((long long)g.C * g.H * g.W >= (1ll << 28)) return false;      // 32-bit byte offsets in the staging loads
  if ((long long)g.topC * g.H * g.W >= (1ll << 30)) return false;   // ... and in the output stores
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
