// Correlation forward, FlowNetC / FlowNet2 instance (kernel_size 1, stride_1 1, stride_2 2, max_displacement 20 = pad, MULTIPLY),
// third generation: (patch, N-tile) UNITS dealt evenly to the waves, a loader wave, a ring that runs one chunk ahead of the readers.
//
// Replaces blob_rearrange_kernel2 + CorrelateData (reference: src/caffe/layers/correlation_layer.cu:23-114) like its predecessor
// corr_fwd_pair (correlation_mfma.hip, which stays as the kernel of the shapes this one does not take).  Same formulation -- in the
// class coordinates of one y parity (y = 2 i + py, x = 2 j + px) the op is a 2-D banded matrix product between 4 x 4 patches of
// positions of the first map (M) and of the second (N), contracted over channels on v_mfma_f32_16x16x4_f32 (exact fp32, a k-ordered
// fma chain: the bits are those of corr_fwd_pair and of the sequential generic kernel) -- but a different decomposition:
//
//   unit  = (M patch p, N tile s) of one (sample, y parity, patch row I, N patch row a): two accumulator tiles (the two x parities
//           of the patch: a lane's operands for both are the 8-byte pair (x, x + 1) of the staged rows), 4 MFMAs per 8-channel chunk.
//           A patch p meets the tiles s = p + b, b = 0 .. 5 (o = 4 b + nj - mj - 10); tiles entirely left / right of the image do not exist.
//   task  = (sample, py, I, a, column task): a column task is 1 .. 3 neighbouring patches with ALL their units (so every output row
//           segment is written by one workgroup), cut by the host so that a row of the image gives 3 - 4 tasks of 9 - 18 units
//           (corr_fwd_pair: 2 tasks of 21 / 15 and one round of workgroups per launch; here 1.5 - 2 rounds, so the tail of a CU is a small task
//           and its epilogue overlaps the K loops of the workgroups that are still running).
//   waves = 4 consumers + 1 loader.  The units of a task are dealt to the consumers by COUNT (5 / 4 / 4 / 4 ...), not by patch column
//           (corr_fwd_pair: 4 / 5 / 6 / 6 tile units per wave, the workgroup runs at the pace of the 6): every unit carries its own LDS
//           operand addresses, so any unit can sit on any wave.  The consumers execute ds_read_b64 + MFMA only: one operand read behind
//           every MFMA, three units ahead of their use, across chunk boundaries.  The loader owns every LDS-DMA instruction of the workgroup
//           (an LDS-DMA issue blocks the issuing wave for 60+ cycles: in corr_fwd_pair that was matrix-pipe time of a wave that
//           runs alone on its SIMD) and the vmcnt bookkeeping.
//   ring  = 4 slots of one 8-channel chunk: [first map: k-step 2][channel 4][row 4][24 px] [second map: k-step 2][channel 4][row 4][8 nb px],
//           rows in natural pixel order, 16-byte LDS-DMA, 1 KiB runs.  Row lengths of an ODD number of 8-pixel tiles make every
//           ds_read_b64 conflict-free in natural layout (row stride 8 n = 8 mod 16 dwords -> the four rows of a half-wave take four
//           disjoint 8-bank windows, channel stride 32 n = 32 mod 64 the other half of the banks): no swizzle, no padding bytes.
//           The barrier that ends chunk c guarantees chunk c + 2 has landed: the readers prefetch into the next chunk without a
//           bubble at the barrier; DMA of chunk c + 4 goes into the slot chunk c just left.
//
// The epilogue is corr_fwd_pair's: accumulators -> LDS image [mi][ni][o][x] (over the ring) -> 16-byte stores of whole row
// segments, 1 / C and the fused ReLU on the way out; N patch rows outside the image are zero-fill tasks without LDS.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "correlation.hpp"

namespace fn2 {
namespace cu3 {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int R = 10, D = 2 * R + 1, NBT = 6;         // displacement radius in class units, displacements per axis, N tiles around a patch
constexpr int CONS = 4, WAVES = 5, THREADS = WAVES * 64;
constexpr int MAXPATCH = 4;                             // patches a task may touch (output image rows of 32 px)
constexpr int MAXRUNS = 12;                             // 1 KiB LDS-DMA runs per 8-channel chunk: na (first map) + nb (second map), both odd
constexpr int SLOTF = 256 * MAXRUNS;                    // floats per ring slot (3072 = 12 KiB)
constexpr int NSLOT = 4;
constexpr int MAXSEG = 12, SEGW = 16, MAXU = 20, MAXMISS = 12, MAXCOMBO = 1024, MAXNU = 5;
constexpr int OROWS = 16 * D;                           // (mi, ni, o) rows of the output image
constexpr int IMGF = OROWS * (8 * MAXPATCH + 1);        // floats of the largest output image; the row table sits behind it
constexpr int LDS_FLOATS = NSLOT * SLOTF;
static_assert(IMGF + OROWS <= LDS_FLOATS, "image + row table fit the ring");
constexpr unsigned OOB = 0x7ffffff0u;                   // beyond any supported sample: reads as 0.0f = the zero padding
constexpr unsigned NOROW = 0xffffffffu;

struct Args {
  int N, C, H, W;
  int TH, TD;            // live / dead tasks per sample
  int LP, DP;            // live / dead list entries per XCD
  int G;                 // > 0: 8 % N == 0, a sample is striped over G = 8 / N XCDs; 0: contiguous ranges of the global lists
  int ctot, c0, relu; float slope;
  int nseg;
  int reserved;
  // Segment task = a contiguous run of the (patch, b) sequence of an image row (patch-major; b = 0 .. 5):
  //  w0 = p0 | patches << 8 | s0 << 16 | nb << 24          first patch, patches touched (<= 4), first second-map tile, tiles staged (odd)
  //  w1 = first unit of consumer wave 0 .. 3 (bytes)        w2 = units of consumer wave 0 .. 3 (bytes)
  //  w3 = 2^16 / (2 nb) + 1 | absent units << 16
  //  w4 = 2^16 / (2 na) + 1 | na << 16 | dp0 << 20 | dnp << 24     na = first-map tiles staged (odd); zero-fill ownership: patches [p0 + dp0, + dnp)
  //  w5 = (blo | bhi << 3) << 6 k, k = 0 .. 3               the b range of patch p0 + k this task owns (a split patch is shared by two tasks)
  //  w6 .. w10 = unit bytes (p_l << 4 | s_l), MAXU          w11 .. w13 = absent units (p_l << 4 | b): tiles entirely outside the image
  unsigned seg[MAXSEG][SEGW];
  unsigned combo[MAXCOMBO / 2];     // 16 bits each: py | I << 1 | a << 6 | seg << 9; live combos [0, TH), then dead ones [TH, TH + TD)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 15, "");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wg_barrier() {
  // "memory": the consumers' loop contains no store the compiler can see (the LDS is written by the loader's DMA), and s_barrier is
  // not a memory operation to LLVM -- without the clobber the loop-invariant LDS loads may be hoisted out of the K loop
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Consumer: NU units, 2 NU items per chunk (k-step r, unit u), each item = 2 operand reads + 2 MFMAs; reads run LA items ahead.
// ABL (FN2_ABLATION builds, profiling only, wrong results): bit 0 no MFMA, bit 1 no LDS-DMA, bit 2 no output stores, bit 3 no barrier inside the K
// loop, bit 4 no operand reads inside the K loop.
template <int NU, int ABL>
__device__ __forceinline__ void consume(float* smem, const Args& g, int lane, int na, int nb, int np, const int (&upl)[MAXNU], const int (&usl)[MAXNU],
                                        const int (&ub)[MAXNU], unsigned long long& t_bar) {
  constexpr int LA = NU == 1 ? 1 : 3, NBUF = LA + 1;
  constexpr int IPC = 2 * NU, WIN = NSLOT * IPC;            // items per chunk, per 4-chunk window
  static_assert(WIN % NBUF == 0, "static operand-buffer indices");
  const int kk = lane >> 4, pi = (lane & 15) >> 2, pj = lane & 3;       // operand role: channel in the k-step, position (row, column) in the patch / tile
  // slot = [first map: k-step 2][channel 4][row 4][8 na px][second map: k-step 2][channel 4][row 4][8 nb px]
  const float* pa[NU][2];
  const float* pb[NU][2];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    pa[u][0] = smem + kk * (32 * na) + pi * (8 * na) + 8 * upl[u] + 2 * pj;
    pa[u][1] = pa[u][0] + 128 * na;
    pb[u][0] = smem + 256 * na + kk * (32 * nb) + pi * (8 * nb) + 8 * usl[u] + 2 * pj;
    pb[u][1] = pb[u][0] + 128 * nb;
  }
  f32x4 acc0[NU], acc1[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) { acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x2 oa[NBUF], ob[NBUF];
  // window item w: slot w / IPC, k-step (w % IPC) / NU, unit w % NU -- static after unrolling, so the slot / k-step offsets are immediates
  auto read_a = [&](int w) { const int i = w % IPC; return *reinterpret_cast<const f32x2*>(pa[i % NU][i / NU] + (w / IPC) * SLOTF); };
  auto read_b = [&](int w) { const int i = w % IPC; return *reinterpret_cast<const f32x2*>(pb[i % NU][i / NU] + (w / IPC) * SLOTF); };

  const int nchunks = g.C / 8;
  wg_barrier();                                             // P: chunks 0 and 1 have landed
#pragma unroll
  for (int w = 0; w < LA; ++w) { oa[w % NBUF] = read_a(w); ob[w % NBUF] = read_b(w); }
#pragma unroll 1
  for (int c4 = 0; c4 < nchunks; c4 += NSLOT) {
#pragma unroll
    for (int w = 0; w < WIN; ++w) {
      const int u = w % NU, cur = w % NBUF, nxt = (w + LA) % NBUF, wn = (w + LA) % WIN;
      if constexpr (ABL & 1) asm volatile("" ::"v"(oa[cur].x), "v"(ob[cur].x));
      else acc0[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[cur].x, ob[cur].x, acc0[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      f32x2 na = oa[nxt], nbv = ob[nxt];
      if constexpr (!(ABL & 16)) na = read_a(wn);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABL & 1) asm volatile("" ::"v"(oa[cur].y), "v"(ob[cur].y));
      else acc1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[cur].y, ob[cur].y, acc1[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 16)) nbv = read_b(wn);
      __builtin_amdgcn_sched_barrier(0);
      oa[nxt] = na; ob[nxt] = nbv;
      if (w % IPC == IPC - 1) {                             // end of a chunk: every read of its slot has fed an MFMA that is issued
        if constexpr (!(ABL & 8)) wg_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#ifdef FN2_ABLATION
  t_bar = __builtin_amdgcn_s_memtime();
#endif
  // ---- accumulators -> LDS image [mi][ni][o][x] (over the ring: every wave is past the last barrier, no DMA is in flight) ----
  const int mi = lane >> 4, ni = (lane & 15) >> 2, nj = lane & 3;   // accumulator role: row = 4 mi + reg (M position (mi, reg)), column = N position (ni, nj)
  const int XS = 8 * np + 1;
  const bool pow2 = (g.C & (g.C - 1)) == 0;
  if (!pow2) {                                              // x / 2^k == x * 2^-k exactly (applied on the way out); otherwise the reference's true division, here
    const float sumelems = (float)g.C;
#pragma unroll
    for (int u = 0; u < NU; ++u) { acc0[u] /= sumelems; acc1[u] /= sumelems; }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oo = 4 * ub[u] + nj - r;                    // o + R
      if (oo >= 0 && oo < D) {
        float* dst = smem + ((mi * 4 + ni) * D + oo) * XS + 8 * upl[u] + 2 * r;
        dst[0] = acc0[u][r];
        dst[1] = acc1[u][r];
      }
    }
  }
}

// Consumer wave without a unit (tiny images): the barriers only.
template <int ABL>
__device__ __forceinline__ void consume_none(const Args& g) {
  const int nchunks = g.C / 8;
  wg_barrier();
  if constexpr (!(ABL & 8))
    for (int c = 0; c < nchunks; ++c) wg_barrier();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Loader: the run plan of the task (one 16-byte slot per lane and run), then per chunk: issue chunk c + 3, wait until c + 2 has landed, barrier.
template <int ABL>
__device__ __forceinline__ void load_task(const float* a_n, const float* b_n, const Args& g, unsigned lds_base, int lane,
                                          int py, int I, int a, int p0, int s0, int na, int nb, unsigned magic_a, unsigned magic_b) {
  const int plane = g.H * g.W;
  const int Rn = na + nb;                                   // 1 KiB runs per chunk: na of the first map, nb of the second
  unsigned voff[MAXRUNS];
#pragma unroll
  for (int i = 0; i < MAXRUNS; ++i) {
    voff[i] = OOB;
    if (i < na) {                                           // [k-step 2][channel 4][row 4][2 na slots of 16 bytes]
      const unsigned q = (unsigned)(i * 64 + lane);
      const unsigned rr = (q * magic_a) >> 16, xs = q - rr * (unsigned)(2 * na);     // rr = (k-step * 4 + channel) * 4 + row
      const int rk = (int)(rr >> 2), row = (int)(rr & 3u);
      const int y = 2 * (4 * I + row) + py, x = 8 * p0 + 4 * (int)xs;
      if (y < g.H && x < g.W) voff[i] = 4u * (unsigned)(rk * plane + y * g.W + x);
    } else if (i < Rn) {                                    // [k-step 2][channel 4][row 4][2 nb slots]
      const unsigned q = (unsigned)((i - na) * 64 + lane);
      const unsigned rr = (q * magic_b) >> 16, xs = q - rr * (unsigned)(2 * nb);
      const int rk = (int)(rr >> 2), row = (int)(rr & 3u);
      const int i2 = 4 * I - R + 4 * a + row, y = 2 * i2 + py, x = 8 * s0 - 2 * R + 4 * (int)xs;
      if (i2 >= 0 && y < g.H && x >= 0 && x < g.W) voff[i] = 4u * (unsigned)(rk * plane + y * g.W + x);
    }
  }
  const unsigned chunk_bytes = 32u * (unsigned)plane;       // 8 channels
  const unsigned sample_bytes = 4u * (unsigned)g.C * (unsigned)plane;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_n), 0, sample_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_n), 0, sample_bytes, 0x00020000);
  auto issue = [&](int c, int slot) {
    if constexpr (ABL & 2) return;
    const unsigned soff = (unsigned)c * chunk_bytes;
    const unsigned base = lds_base + 4u * (unsigned)(slot * SLOTF);
#pragma unroll
    for (int i = 0; i < MAXRUNS; ++i) {
      if (i < Rn) {
        lds_ptr_t lp = (lds_ptr_t)(uintptr_t)(base + 1024u * (unsigned)i);
        if (i < na) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, lp, 16, voff[i], soff, 0, 0);
        else        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, lp, 16, voff[i], soff, 0, 0);
      }
    }
  };
  auto wait_all_but_newest_chunk = [&]() {                  // Rn is even, 2 .. 12
    switch (Rn) {
      case 2: wait_vmcnt<2>(); break;
      case 4: wait_vmcnt<4>(); break;
      case 6: wait_vmcnt<6>(); break;
      case 8: wait_vmcnt<8>(); break;
      case 10: wait_vmcnt<10>(); break;
      default: wait_vmcnt<12>(); break;
    }
  };
  const int nchunks = g.C / 8;
  issue(0, 0);
  if (nchunks > 1) issue(1, 1);
  if (nchunks > 2) { issue(2, 2); wait_all_but_newest_chunk(); } else wait_vmcnt<0>();
  wg_barrier();                                             // P
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    if (c + 3 < nchunks) { issue(c + 3, (c + 3) & 3); wait_all_but_newest_chunk(); }
    else wait_vmcnt<0>();
    if constexpr (!(ABL & 8)) wg_barrier();                 // B_c: chunks <= c + 2 have landed, nobody reads chunk c any more
  }
}

template <int ABL>
__global__ void __launch_bounds__(THREADS, 4)
corr_fwd_units(const float* __restrict__ b0, const float* __restrict__ b1, float* __restrict__ top, Args g, unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef FN2_ABLATION
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  // ---- task decode (scalar) ----
  const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
  const bool live = j < g.LP;
  const int jj = live ? j : j - g.LP;
  const int per = live ? g.TH : g.TD;
  int n, tt;
  if (g.G > 0) {
    n = xcd / g.G; tt = xcd % g.G + g.G * jj;
    if (tt >= per) return;
  } else {
    const long long t = (long long)xcd * (live ? g.LP : g.DP) + jj;
    if (t >= (long long)g.N * per) return;
    n = (int)(t / per); tt = (int)(t % per);
  }
  const int ci = (live ? 0 : g.TH) + tt;
  const unsigned cw = g.combo[ci >> 1];
  const unsigned cb = (ci & 1) ? (cw >> 16) : (cw & 0xffffu);
  const int py = (int)(cb & 1u), I = (int)((cb >> 1) & 31u), a = (int)((cb >> 6) & 7u), segI = (int)((cb >> 9) & 15u);
  const unsigned* sw = g.seg[segI];
  const unsigned w0 = sw[0], w4 = sw[4];
  const int p0 = (int)(w0 & 255u), np = (int)((w0 >> 8) & 255u), s0 = (int)((w0 >> 16) & 255u), nb = (int)(w0 >> 24);
  const int na = (int)((w4 >> 16) & 15u), dp0 = (int)((w4 >> 20) & 15u), dnp = (int)((w4 >> 24) & 15u);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = 4 * I;
  const size_t plane = (size_t)g.H * g.W;
  const size_t top_n = (size_t)n * g.ctot + g.c0;
  // Output rows of this task: rowid = (rmi * 4 + rni) * D + oo  <->  top[n, (qq = 4a + rni - rmi, oo), y = 2 (4I + rmi) + py, 8 np px from 8 p0];
  // offsets are 32-bit inside the sample's output (buffer store).
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(top + top_n * plane, 0, (unsigned)(4u * D * D * (unsigned)plane), 0x00020000);
  const unsigned hw4 = 4u * (unsigned)plane, w4b = 4u * (unsigned)g.W;
  auto row_offset = [&](int rowid) -> unsigned {
    const int blk = rowid / D, oo = rowid - blk * D, rmi = blk >> 2, rni = blk & 3;
    const int qq = 4 * a + rni - rmi, y = 2 * (i0 + rmi) + py;
    return (qq >= 0 && qq < D && y < g.H) ? (unsigned)(qq * D + oo) * hw4 + (unsigned)y * w4b : NOROW;
  };
  if (!live) {                                              // N patch row outside the image: zeros for the patches this task owns, no LDS
    const int lpr = 2 * dnp, rpp = THREADS / lpr;           // dnp >= 1: the host lists no zero-fill task for a segment that owns no patch
    const int trow = tid / lpr, xq = tid - trow * lpr;
    const int x = 8 * (p0 + dp0) + 4 * xq;
    if (trow >= rpp || x >= g.W) return;                    // W % 4 == 0: a quad is inside or outside as a whole
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int rowid = trow; rowid < OROWS; rowid += rpp) {
      const unsigned off = row_offset(rowid);
      if (off != NOROW) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, z), rsT, off + 4u * (unsigned)x, 0, 0);
    }
    return;
  }

  const float* a_n = b0 + (size_t)n * g.C * plane;
  const float* b_n = b1 + (size_t)n * g.C * plane;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  [[maybe_unused]] unsigned long long t_bar = 0, t_sync = 0;
  if (wave == CONS) {
    load_task<ABL>(a_n, b_n, g, lds_base, lane, py, I, a, p0, s0, na, nb, w4 & 0xffffu, sw[3] & 0xffffu);
#ifdef FN2_ABLATION
    t_bar = __builtin_amdgcn_s_memtime();
#endif
  } else {
    const int u0 = (int)((sw[1] >> (8 * wave)) & 255u), nu = (int)((sw[2] >> (8 * wave)) & 255u);
    int upl[MAXNU], usl[MAXNU], ub[MAXNU];
#pragma unroll
    for (int u = 0; u < MAXNU; ++u) {
      const int k = u0 + (u < nu ? u : 0);
      const unsigned byte = (sw[6 + (k >> 2)] >> (8 * (k & 3))) & 255u;
      upl[u] = (int)(byte >> 4); usl[u] = (int)(byte & 15u);
      ub[u] = usl[u] + s0 - (upl[u] + p0);                  // b = s - p
    }
    switch (nu) {
      case 0: consume_none<ABL>(g); break;
      case 1: consume<1, ABL>(smem, g, lane, na, nb, np, upl, usl, ub, t_bar); break;
      case 2: consume<2, ABL>(smem, g, lane, na, nb, np, upl, usl, ub, t_bar); break;
      case 3: consume<3, ABL>(smem, g, lane, na, nb, np, upl, usl, ub, t_bar); break;
      case 4: consume<4, ABL>(smem, g, lane, na, nb, np, upl, usl, ub, t_bar); break;
      default: consume<5, ABL>(smem, g, lane, na, nb, np, upl, usl, ub, t_bar); break;
    }
  }
#ifdef FN2_ABLATION
  const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
  // ---- units that do not exist (N tile entirely outside the image: nothing staged, nothing multiplied) still own a part of the output
  // image: zeros (the reference's zero padding).  Disjoint from every other unit's part; all five waves share them.
  {
    const int nmiss = (int)(sw[3] >> 16);
    const int mi = lane >> 4, ni = (lane & 15) >> 2, nj = lane & 3;
    const int XS = 8 * np + 1;
    for (int k = wave; k < nmiss; k += WAVES) {
      const unsigned byte = (sw[11 + (k >> 2)] >> (8 * (k & 3))) & 255u;
      const int pl = (int)(byte >> 4), b = (int)(byte & 15u);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oo = 4 * b + nj - r;
        if (oo >= 0 && oo < D) {
          float* dst = smem + ((mi * 4 + ni) * D + oo) * XS + 8 * pl + 2 * r;
          dst[0] = 0.f;
          dst[1] = 0.f;
        }
      }
    }
  }
  // ---- rows out: 1 / C (power-of-two channel counts) and the fused ReLU applied on the way.  thread -> (row of a pass, 16-byte quad of the
  // row segment): 2 np quads per row.  A quad is the pixels (r0, r0 + 1) x both x parities of ONE patch; element (oo, r) of a patch comes
  // from the unit b = (oo + r) >> 2, so where two tasks share a patch (a split in b) each stores the halves of the quad whose b it owns.
  unsigned* rowtab = reinterpret_cast<unsigned*>(smem + IMGF);
  for (int rowid = tid; rowid < OROWS; rowid += THREADS) rowtab[rowid] = row_offset(rowid);
  __syncthreads();
#ifdef FN2_ABLATION
  t_sync = __builtin_amdgcn_s_memtime();
#endif
  const int lpr = 2 * np, rpp = THREADS / lpr;              // lanes per row, rows per pass
  const int trow = tid / lpr, xq = tid - trow * lpr;
  const int x = 8 * p0 + 4 * xq;
  if (trow < rpp && x < g.W) {
    const bool pow2 = (g.C & (g.C - 1)) == 0;
    const float scale = pow2 ? 1.0f / (float)g.C : 1.0f;
    const float slope = g.slope;
    const bool relu = g.relu != 0;
    const int XS = 8 * np + 1;
    const unsigned br = (sw[5] >> (6 * (xq >> 1))) & 63u;
    const int blo = (int)(br & 7u), bhi = (int)(br >> 3), r0 = 2 * (xq & 1);
    const bool whole = blo == 0 && bhi == NBT - 1;
#pragma unroll 1
    for (int rowid = trow; rowid < OROWS; rowid += rpp) {
      const unsigned off = rowtab[rowid];
      if (off == NOROW) continue;
      const float* src = smem + rowid * XS + 4 * xq;
      float f0 = src[0] * scale, f1 = src[1] * scale, f2 = src[2] * scale, f3 = src[3] * scale;
      if (relu) {
        f0 = f0 > 0.f ? f0 : f0 * slope; f1 = f1 > 0.f ? f1 : f1 * slope;
        f2 = f2 > 0.f ? f2 : f2 * slope; f3 = f3 > 0.f ? f3 : f3 * slope;
      }
      if constexpr (ABL & 4) continue;
      const unsigned dst = off + 4u * (unsigned)x;
      bool m0 = true, m1 = true;                            // this task owns pixel pair r0 / r0 + 1 of the quad in this row
      if (!whole) {
        const int oo = rowid - (rowid / D) * D;
        const int q0 = (oo + r0) >> 2, q1 = (oo + r0 + 1) >> 2;
        m0 = q0 >= blo && q0 <= bhi;
        m1 = q1 >= blo && q1 <= bhi;
      }
      const unsigned u0 = __float_as_uint(f0), u1 = __float_as_uint(f1), u2 = __float_as_uint(f2), u3 = __float_as_uint(f3);
      if (m0 && m1) {
        u32x4 q4; q4.x = u0; q4.y = u1; q4.z = u2; q4.w = u3;
        __builtin_amdgcn_raw_buffer_store_b128(q4, rsT, dst, 0, 0);
      } else if (m0) {
        u32x2 q2; q2.x = u0; q2.y = u1;
        __builtin_amdgcn_raw_buffer_store_b64(q2, rsT, dst, 0, 0);
      } else if (m1) {
        u32x2 q2; q2.x = u2; q2.y = u3;
        __builtin_amdgcn_raw_buffer_store_b64(q2, rsT, dst + 8u, 0, 0);
      }
    }
  }
#ifdef FN2_ABLATION
  if (dbg && lane == 0 && blockIdx.x < 4096) {             // per wave: start, K loop end, end, {HW_ID, XCC_ID, wave, units of the task}
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* wrec = dbg + 6 * (WAVES * blockIdx.x + wave);
    wrec[0] = t_start;
    wrec[1] = t_loop;
    wrec[2] = __builtin_amdgcn_s_memtime();
    wrec[4] = t_bar;
    wrec[5] = t_sync;
    wrec[3] = (unsigned long long)hwid | ((unsigned long long)(xcc & 15u) << 32) | ((unsigned long long)wave << 36) | ((unsigned long long)segI << 40) | (1ull << 63);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Host side: column tasks, unit lists, task tables -- a function of (N, H, W, policy), built once per geometry.
struct Plan { int N, H, W, policy; bool ok; Args a; unsigned grid; };

static void live_range(int I, int Hc, int& alo, int& ahi) {          // live N patch rows of M patch row I (second-map rows touch the image)
  const int lo_num = R - 3 - 4 * I;
  alo = lo_num <= 0 ? 0 : (lo_num + 3) / 4;
  const int hi_num = Hc - 1 + R - 4 * I;
  ahi = hi_num < 0 ? -1 : hi_num / 4;
  if (ahi > NBT - 1) ahi = NBT - 1;
  if (4 * I >= Hc) { alo = 0; ahi = -1; }
}

// policy: 0 = automatic (below), 1 .. 15 = that many tasks per image row (tests / profiling); + 16: task order inside a sample = image
// order instead of most-units-first.
static bool build_plan(int N, int H, int W, int policy, Plan& pl) {
  pl.N = N; pl.H = H; pl.W = W; pl.policy = policy; pl.ok = false;
  Args& g = pl.a;
  std::memset(&g, 0, sizeof(g));
  if (W % 4 != 0 || H < 1 || H > 255 || N < 1) return false;
  const int Wc = W / 2, NP = (Wc + 3) / 4;
  const int smax = (Wc - 1 + R) / 4;
  if (NP > 64) return false;
  // (py, I, a) combinations of one sample
  struct Cb { int py, I, a, seg, units; };
  std::vector<Cb> rows_live, rows_dead;
  const int NI = ((H + 1) / 2 + 3) / 4;
  if (NI > 32) return false;
  for (int py = 0; py < 2; ++py) {
    const int Hc = (H - py + 1) / 2;
    for (int I = 0; I < NI; ++I) {
      if (4 * I >= Hc) continue;                            // no output row in this patch row (odd heights, y parity 1)
      int alo, ahi;
      live_range(I, Hc, alo, ahi);
      for (int a = 0; a < NBT; ++a) (a >= alo && a <= ahi ? rows_live : rows_dead).push_back({py, I, a, 0, 0});
    }
  }
  if (rows_live.empty()) return false;
  // the (patch, b) sequence of an image row; live[k]: the N tile p + b touches the image
  const int L = NBT * NP;
  std::vector<char> livek(L);
  int U_row = 0;
  for (int k = 0; k < L; ++k) { const int s = k / NBT + k % NBT; livek[k] = (s >= 2 && s <= smax); U_row += livek[k]; }
  if (U_row == 0) return false;
  int nseg;
  const bool automatic = (policy & 15) == 0;
  if (!automatic) nseg = policy & 15;
  else {
    // ONE round of the chip (768 workgroup slots: three per CU) of the largest tasks that fit a workgroup: the launch has no second,
    // half-empty round (measured at config A: 1,152 tasks of 15 / 12 / 9 units 40.2 us, 768 of 18 units 39.4) and every task ends within the
    // same K loop.  Geometries whose rows would need more than 768 tasks of <= 20 units keep corr_fwd_pair (config B: 960 tasks 50.9 us
    // against 42.8), geometries with few rows are cut finer (config D, 72 rows: 720 tasks of 9 units).
    const long long rows_all = (long long)N * (long long)rows_live.size();
    nseg = (int)std::min<long long>(MAXSEG, 768 / std::max<long long>(1, rows_all));
    nseg = std::min(nseg, std::max(1, U_row / 8));          // ... but no task below eight units (two per consumer wave)
    if (nseg < 1 || (U_row + nseg - 1) / nseg > CONS * MAXNU) return false;
  }
  nseg = std::max(1, std::min(nseg, MAXSEG));
  int units_of[MAXSEG];
  for (;; ++nseg) {                                         // more, smaller tasks until every task fits (patches, staged tiles, units per wave)
    if (nseg > MAXSEG || nseg > U_row) return false;
    if (automatic && (long long)N * (long long)rows_live.size() * nseg > 768) return false;
    std::memset(g.seg, 0, sizeof(g.seg));
    bool fits = true;
    int k0 = 0, done = 0;
    for (int c = 0; c < nseg && fits; ++c) {
      const int want = U_row * (c + 1) / nseg - done;       // live units of this segment
      int k1 = k0, got = 0;
      while (k1 < L && (got < want || (c == nseg - 1))) { got += livek[k1]; ++k1; }
      if (c < nseg - 1) while (k1 < L && !livek[k1] && (k1 % NBT) != 0) ++k1;     // absent tail of a patch stays with the task that holds its last live b
      done += got;
      units_of[c] = got;
      const int pf = k0 / NBT, plast = (k1 - 1) / NBT, np = plast - pf + 1;
      if (got == 0 || np > MAXPATCH || got > MAXU || got > CONS * MAXNU) { fits = false; break; }
      unsigned* w = g.seg[c];
      int s_lo = 1 << 30, s_hi = -1, U = 0, nmiss = 0;
      for (int k = k0; k < k1; ++k) if (livek[k]) { const int sx = k / NBT + k % NBT; s_lo = std::min(s_lo, sx); s_hi = std::max(s_hi, sx); }
      int nb = s_hi - s_lo + 1;
      if (nb % 2 == 0) ++nb;                                // odd: conflict-free natural row layout (the extra tile is staged, never multiplied)
      const int na = np | 1;
      if (na + nb > MAXRUNS) { fits = false; break; }
      for (int k = k0; k < k1; ++k) {
        const int q = k / NBT, b = k % NBT;
        if (livek[k]) { w[6 + (U >> 2)] |= (unsigned)(((q - pf) << 4) | (q + b - s_lo)) << (8 * (U & 3)); ++U; }
        else {
          if (nmiss >= MAXMISS) { fits = false; break; }
          w[11 + (nmiss >> 2)] |= (unsigned)(((q - pf) << 4) | b) << (8 * (nmiss & 3)); ++nmiss;
        }
      }
      if (!fits) break;
      w[0] = (unsigned)pf | ((unsigned)np << 8) | ((unsigned)s_lo << 16) | ((unsigned)nb << 24);
      int u0 = 0;
      for (int wv = 0; wv < CONS; ++wv) {                   // wave 0 (shares its SIMD with the loader) takes the smaller share
        const int nu = U / CONS + (wv >= CONS - U % CONS ? 1 : 0);
        w[1] |= (unsigned)u0 << (8 * wv);
        w[2] |= (unsigned)nu << (8 * wv);
        u0 += nu;
      }
      const unsigned db = 2u * (unsigned)nb, mb = 65536u / db + 1u, da = 2u * (unsigned)na, ma = 65536u / da + 1u;
      for (unsigned q = 0; q < 64u * (unsigned)nb; ++q) if (((q * mb) >> 16) != q / db) return false;      // the loader's divisions by multiplication,
      for (unsigned q = 0; q < 64u * (unsigned)na; ++q) if (((q * ma) >> 16) != q / da) return false;      // checked for every slot index it can see
      w[3] = mb | ((unsigned)nmiss << 16);
      // zero-fill ownership: the patches whose b = 0 lies in this segment
      int dfirst = -1, dn = 0;
      for (int q = pf; q <= plast; ++q) if (q * NBT >= k0 && q * NBT < k1) { if (dfirst < 0) dfirst = q; ++dn; }
      w[4] = ma | ((unsigned)na << 16) | ((unsigned)(dfirst < 0 ? 0 : dfirst - pf) << 20) | ((unsigned)dn << 24);
      for (int q = pf; q <= plast; ++q) {
        const int blo = std::max(k0, q * NBT) - q * NBT, bhi = std::min(k1, (q + 1) * NBT) - 1 - q * NBT;
        w[5] |= (unsigned)(blo | (bhi << 3)) << (6 * (q - pf));
      }
      k0 = k1;
    }
    if (fits && k0 == L) break;
  }
  g.nseg = nseg;
  std::vector<Cb> lv, dd;
  for (const Cb& r : rows_live) for (int c = 0; c < nseg; ++c) lv.push_back({r.py, r.I, r.a, c, units_of[c]});
  for (const Cb& r : rows_dead) for (int c = 0; c < nseg; ++c) if ((g.seg[c][4] >> 24) != 0) dd.push_back({r.py, r.I, r.a, c, 0});
  if (lv.size() + dd.size() > (size_t)MAXCOMBO) return false;
  if (!(policy & 16)) std::stable_sort(lv.begin(), lv.end(), [](const Cb& x, const Cb& y) { return x.units > y.units; });
  auto put = [&](size_t idx, const Cb& c) {
    const unsigned v = (unsigned)c.py | ((unsigned)c.I << 1) | ((unsigned)c.a << 6) | ((unsigned)c.seg << 9);
    g.combo[idx >> 1] |= v << (16 * (idx & 1));
  };
  for (size_t i = 0; i < lv.size(); ++i) put(i, lv[i]);
  for (size_t i = 0; i < dd.size(); ++i) put(lv.size() + i, dd[i]);
  g.N = N; g.H = H; g.W = W;
  g.TH = (int)lv.size(); g.TD = (int)dd.size();
  if (N <= 8 && 8 % N == 0) {
    g.G = 8 / N;
    g.LP = (g.TH + g.G - 1) / g.G;
    g.DP = (g.TD + g.G - 1) / g.G;
  } else {
    g.G = 0;
    const long long NL = (long long)N * g.TH, ND = (long long)N * g.TD;
    if (NL + ND > (1ll << 28)) return false;
    g.LP = (int)((NL + 7) / 8);
    g.DP = (int)((ND + 7) / 8);
  }
  pl.grid = 8u * (unsigned)(g.LP + g.DP);
  pl.ok = true;
  return true;
}

// Returned BY VALUE under the lock (3 KB): the cache may evict an entry while another host thread still launches with it.
static Plan plan_for(int N, int H, int W, int policy) {
  static std::mutex mu;
  static std::vector<Plan*> cache;
  std::lock_guard<std::mutex> lock(mu);
  for (const Plan* p : cache)
    if (p->N == N && p->H == H && p->W == W && p->policy == policy) return *p;
  Plan* p = new Plan;
  build_plan(N, H, W, policy, *p);
  if (cache.size() >= 64) { delete cache.front(); cache.erase(cache.begin()); }
  cache.push_back(p);
  return *p;
}

}  // namespace cu3

extern unsigned long long* g_corr_dbg;
int g_corr_units_abl = 0;      // FN2_ABLATION builds: ablation bits of corr_fwd_units
int g_corr_units = 1;          // test / profiling hook (fn2_debug_set_correlation_impl): 0 = corr_fwd_pair where both apply, 1 + policy = this kernel
int g_corr_units_lds = 0;      // profiling hook: extra dynamic LDS per workgroup (bytes) -- fewer workgroups per CU

bool corr_fwd_units_supported(const CorrGeom& g, const float* b0, const float* b1, const float* top) {
  if (g_corr_units == 0) return false;
  if (g.K != 1 || g.s1 != 1 || g.type != FN2_CORR_MULTIPLY || g.pad != g.md || g.s2 != 2 || g.ngr != cu3::R) return false;
  if (g.C % 32 != 0 || g.W % 4 != 0) return false;
  if ((long long)g.C * g.H * g.W >= (1ll << 28) || (long long)g.topC * g.H * g.W >= (1ll << 30)) return false;
  if (((reinterpret_cast<uintptr_t>(b0) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(top)) & 15) != 0) return false;
  return cu3::plan_for(g.N, g.H, g.W, g_corr_units - 1).ok;
}

// test hook: the plan of a geometry as raw 32-bit words (tests/test_corr_units_plan.py walks it on the CPU: every output element written exactly once)
int corr_fwd_units_plan_words(int N, int H, int W, int policy, unsigned* out, int max_words) {
  const cu3::Plan pl = cu3::plan_for(N, H, W, policy);
  if (!pl.ok) return 0;
  const int nw = (int)(sizeof(cu3::Args) / 4);
  if (out && max_words >= nw + 1) { std::memcpy(out, &pl.a, sizeof(cu3::Args)); out[nw] = pl.grid; }
  return nw + 1;
}

int corr_fwd_units_launch(const CorrGeom& cg, const float* b0, const float* b1, float* top, hipStream_t st) {
  const cu3::Plan pl = cu3::plan_for(cg.N, cg.H, cg.W, g_corr_units - 1);
  if (!pl.ok) return fail(FN2_ERR_UNSUPPORTED, "correlation: no unit plan for %d x %d x %d", cg.N, cg.H, cg.W);
  cu3::Args a = pl.a;
  a.C = cg.C; a.ctot = cg.top_ctot; a.c0 = cg.top_c0; a.relu = cg.relu; a.slope = cg.slope;
#ifdef FN2_ABLATION
  static const int env_pad = getenv("FN2_CORR_LDS_PAD") ? atoi(getenv("FN2_CORR_LDS_PAD")) : 0;     // profiling builds: fewer workgroups per CU
  const size_t lds = sizeof(float) * cu3::LDS_FLOATS + (size_t)(g_corr_units_lds ? g_corr_units_lds : env_pad);
#else
  const size_t lds = sizeof(float) * cu3::LDS_FLOATS + (size_t)g_corr_units_lds;
#endif
  auto go = [&](auto kernel) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return fail(FN2_ERR_UNSUPPORTED, "correlation_forward (units): %zu bytes of dynamic LDS refused by the runtime", lds);
    }
    hipLaunchKernelGGL(kernel, dim3(pl.grid), dim3(cu3::THREADS), lds, st, b0, b1, top, a, g_corr_dbg);
    return check_launch("correlation_forward (mfma, units)");
  };
#ifdef FN2_ABLATION
  switch (g_corr_units_abl) {        // profiling builds (fn2_debug_set_correlation_impl(100 + bits)): wrong results
    case 1: return go(&cu3::corr_fwd_units<1>);
    case 2: return go(&cu3::corr_fwd_units<2>);
    case 3: return go(&cu3::corr_fwd_units<3>);
    case 4: return go(&cu3::corr_fwd_units<4>);
    case 8: return go(&cu3::corr_fwd_units<8>);
    case 16: return go(&cu3::corr_fwd_units<16>);
    case 18: return go(&cu3::corr_fwd_units<18>);
    case 26: return go(&cu3::corr_fwd_units<26>);
    case 22: return go(&cu3::corr_fwd_units<22>);
    case 5: return go(&cu3::corr_fwd_units<5>);
    default: break;
  }
#endif
  return go(&cu3::corr_fwd_units<0>);
}

}  // namespace fn2
