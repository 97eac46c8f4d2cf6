// Correlation forward, FlowNetC / FlowNet2 instance (kernel_size 1, stride_1 1, stride_2 2, max_displacement 20 = pad, MULTIPLY):
// the PERSISTENT form of the unit kernel (correlation_units.hip, where the formulation, the unit / segment vocabulary and the ring
// layout are described).  Replaces blob_rearrange_kernel2 + CorrelateData (reference: src/caffe/layers/correlation_layer.cu:23-114).
//
// What the traces of the one-task-per-workgroup kernels (corr_fwd_pair, corr_fwd_units) showed (profiles/r05_corr_notes.md): their K loops
// keep the matrix pipes ~97 % busy while three workgroups share a CU, but the launch is one round of the chip, so (a) every epilogue
// -- 31.6 MB of stores at config A -- falls into the same few microseconds at the end, nothing left to overlap it with, or (b) where
// the oldest-first arbitration staggers the workgroups of a CU, the last one runs alone at 57 % (a ring of four chunks gives the
// LDS-DMA two chunk times of lead; a lone workgroup needs four), and (c) VALU / LDS / store instructions of an epilogue take 3-4x
// longer while other waves stream MFMAs on the same SIMD.  Here instead:
//
//   * ONE workgroup per CU (256 of them, 141 KB of LDS), which walks a host-planned list of 2 - 4 segment tasks; the list of a CU adds up to
//     the same unit count on every CU (config A: 54 = tasks of 20 / 16 / 18 units cut from 1.5 image rows), and per-wave counts that
//     add up evenly over the four SIMDs (14 / 14 / 13 / 13): no dynamic balance, no tail;
//   * twelve waves: EIGHT consumers (ds_read_b64 + MFMA only, two per SIMD: one wave alone issues these MFMAs every ~41 cycles, two
//     interleave at the pipe's 32 -- measured, profiles/r05_corr_notes.md), three LOADERS (every LDS-DMA instruction, a third of a chunk's
//     runs each -- one wave's LDS-DMA lands at ~25 GB/s, MI355X_MICROARCH.md "ldsdma-fill", and a CU that multiplies at full rate eats
//     45 GB/s; with ONE loader this kernel ran at 1,800 cycles per chunk step instead of 640: a ring of EIGHT chunks, issued five
//     chunks ahead, across task boundaries -- the first chunks of task t + 1 land while task t computes), and a STORER: consumers drop their accumulators into the LDS output image at the end of a task and go straight on; the storer writes the
//     image out during the next task's K loop, two 64-lane store instructions per chunk step (and, during the first task, the zero
//     rows of the N patch rows that lie outside the image).  Only the LAST task's image is stored by all twelve waves at the end;
//   * one s_barrier per chunk step for all twelve waves; the barrier of step g guarantees chunk g + 2 has landed, so the consumers'
//     operand prefetch crosses the barrier without a bubble.
//
// Arithmetic: v_mfma_f32_16x16x4_f32, channels in order -- the bits of corr_fwd_pair, corr_fwd_units and the sequential generic kernel.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "correlation.hpp"

namespace fn2 {
namespace cu4 {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int R = 10, D = 2 * R + 1, NBT = 6;
constexpr int CONS = 8, NLOAD = 3, NSTORE = 2, WAVES = CONS + NLOAD + NSTORE, THREADS = WAVES * 64;      // waves 0-7 consumers (w and w + 4 share a SIMD), 8-10 loaders, 11-12 storers
constexpr int MAXPATCH = 4, MAXRUNS = 12;
constexpr int SLOTF = 256 * MAXRUNS;                    // floats per ring slot (12 KiB)
constexpr int NSLOT = 8;                                // ring slots; the K loop is unrolled over one turn of the ring
constexpr int AHEAD = 5;                                // chunks the loader runs ahead of the chunk being multiplied
constexpr int MAXSEG = 12, SEGW = 16, MAXU = 20, MAXMISS = 12, MAXNU = 3, MAXT = 4, MAXWG = 256;
constexpr int OROWS = 16 * D, XS = 8 * MAXPATCH + 1;    // output image: (mi, ni, o) rows of 32 px + 1
constexpr int RINGF = NSLOT * SLOTF, IMGF = OROWS * XS;
constexpr int LDS_FLOATS = RINGF + IMGF + OROWS;        // ring | output image | row table
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "one workgroup per CU");
constexpr int QUADS = OROWS * 2 * MAXPATCH;             // 16-byte quads of an output image
constexpr unsigned OOB = 0x7ffffff0u, NOROW = 0xffffffffu, NOTASK = 0xffffu;

struct Args {
  int N, C, H, W;
  int G;                 // XCDs per sample (8 / N); a sample's 32 G workgroups share its task table
  int ctot, c0, relu; float slope;
  int RN;                // LDS-DMA runs per chunk, the same for every task of the launch (tasks that stage less issue dummy runs)
  int ND;                // dead (py, I, a) row tasks per sample; zero-filled in quarters (one mi each), dealt to the sample's workgroups
  int ZP;                // zero quarters per workgroup
  int flags;
  unsigned seg[MAXSEG][SEGW];       // segment tasks: the word layout of correlation_units.hip
  unsigned short task[MAXWG][MAXT]; // per workgroup of a sample: py | I << 1 | a << 6 | seg << 9 | rot << 13; 0xffff = none
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");       // the LDS is written by another wave's DMA: nothing the compiler sees -- keep its loads where they are
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct TaskInfo { int py, I, a, seg, rot, p0, np, s0, nb, na; bool valid; };

__device__ __forceinline__ TaskInfo decode(const Args& g, const unsigned (&entries)[MAXT], int t) {
  TaskInfo k{};
  unsigned e = NOTASK;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) if (i == t) e = entries[i];
  k.valid = e != NOTASK;
  if (!k.valid) return k;
  k.py = (int)(e & 1u); k.I = (int)((e >> 1) & 31u); k.a = (int)((e >> 6) & 7u); k.seg = (int)((e >> 9) & 15u); k.rot = (int)((e >> 13) & 3u);
  const unsigned w0 = g.seg[k.seg][0], w4 = g.seg[k.seg][4];
  k.p0 = (int)(w0 & 255u); k.np = (int)((w0 >> 8) & 255u); k.s0 = (int)((w0 >> 16) & 255u); k.nb = (int)(w0 >> 24);
  k.na = (int)((w4 >> 16) & 15u);
  return k;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Consumer, one task: NU units, 2 NU items per chunk; operand reads LA items ahead of their MFMAs; one barrier per chunk.
template <int NU>
__device__ __forceinline__ void consume(float* smem, const Args& g, int lane, int wave, const TaskInfo& k, const int (&upl)[MAXNU], const int (&usl)[MAXNU]) {
  constexpr int LA = NU == 1 ? 1 : 3, NBUF = LA + 1;
  constexpr int IPC = 2 * NU, WIN = NSLOT * IPC;
  static_assert(WIN % NBUF == 0, "static operand-buffer indices");
  const int kk = lane >> 4, pi = (lane & 15) >> 2, pj = lane & 3;
  const int na = k.na, nb = k.nb;
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
  auto read_a = [&](int w) { const int i = w % IPC; return *reinterpret_cast<const f32x2*>(pa[i % NU][i / NU] + (w / IPC) * SLOTF); };
  auto read_b = [&](int w) { const int i = w % IPC; return *reinterpret_cast<const f32x2*>(pb[i % NU][i / NU] + (w / IPC) * SLOTF); };
  const int nchunks = g.C / 8;                              // a multiple of NSLOT: every task starts in ring slot 0
#pragma unroll
  for (int w = 0; w < LA; ++w) { oa[w % NBUF] = read_a(w); ob[w % NBUF] = read_b(w); }
#pragma unroll 1
  for (int c8 = 0; c8 < nchunks; c8 += NSLOT) {
#pragma unroll
    for (int w = 0; w < WIN; ++w) {
      const int u = w % NU, cur = w % NBUF, nxt = (w + LA) % NBUF, wn = (w + LA) % WIN;
      acc0[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[cur].x, ob[cur].x, acc0[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const f32x2 ra = read_a(wn);
      __builtin_amdgcn_sched_barrier(0);
      acc1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[cur].y, ob[cur].y, acc1[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const f32x2 rb = read_b(wn);
      __builtin_amdgcn_sched_barrier(0);
      oa[nxt] = ra; ob[nxt] = rb;
      if (w % IPC == IPC - 1) {
        wg_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // ---- accumulators -> LDS output image [mi][ni][o][x] (its own LDS: the storer emptied it during this task's K loop) ----
  float* img = smem + RINGF;
  const int mi = lane >> 4, ni = (lane & 15) >> 2, nj = lane & 3;
  const bool pow2 = (g.C & (g.C - 1)) == 0;
  if (!pow2) {
    const float sumelems = (float)g.C;
#pragma unroll
    for (int u = 0; u < NU; ++u) { acc0[u] /= sumelems; acc1[u] /= sumelems; }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int b = usl[u] + k.s0 - (upl[u] + k.p0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oo = 4 * b + nj - r;
      if (oo >= 0 && oo < D) {
        float* dst = img + ((mi * 4 + ni) * D + oo) * XS + 8 * upl[u] + 2 * r;
        dst[0] = acc0[u][r];
        dst[1] = acc1[u][r];
      }
    }
  }
}

__device__ __forceinline__ void consume_none(const Args& g) {
  const int nchunks = g.C / 8;
  for (int c = 0; c < nchunks; ++c) wg_barrier();
}

// absent units of a task (N tile entirely outside the image): zeros into the output image; shared by the four consumer waves
__device__ __forceinline__ void zero_absent(float* smem, const Args& g, const TaskInfo& k, int lane, int wave) {
  float* img = smem + RINGF;
  const unsigned* sw = g.seg[k.seg];
  const int nmiss = (int)(sw[3] >> 16);
  const int mi = lane >> 4, ni = (lane & 15) >> 2, nj = lane & 3;
  for (int m = wave; m < nmiss; m += CONS) {
    const unsigned byte = (sw[11 + (m >> 2)] >> (8 * (m & 3))) & 255u;
    const int pl = (int)(byte >> 4), b = (int)(byte & 15u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oo = 4 * b + nj - r;
      if (oo >= 0 && oo < D) {
        float* dst = img + ((mi * 4 + ni) * D + oo) * XS + 8 * pl + 2 * r;
        dst[0] = 0.f;
        dst[1] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Loader: per task a run plan (one 16-byte slot per lane and run); the chunk stream runs AHEAD chunks in front of the multipliers.
constexpr int MYRUNS = (MAXRUNS + NLOAD - 1) / NLOAD;    // runs of a chunk one loader issues: run i = li + NLOAD * slot
struct LoadPlan { unsigned voff[MYRUNS]; int na; };

__device__ __forceinline__ void make_plan(LoadPlan& lp, const Args& g, const TaskInfo& k, int lane, int li) {
  const int plane = g.H * g.W;
  const unsigned* sw = g.seg[k.seg];
  const unsigned magic_a = sw[4] & 0xffffu, magic_b = sw[3] & 0xffffu;
  const int na = k.na, nb = k.nb, Rn = na + nb;
  lp.na = na;
#pragma unroll
  for (int sl = 0; sl < MYRUNS; ++sl) {
    const int i = li + NLOAD * sl;
    lp.voff[sl] = OOB;                                      // runs beyond na + nb: dummies (zeros into the unused end of the slot)
    if (i < na) {
      const unsigned q = (unsigned)(i * 64 + lane);
      const unsigned rr = (q * magic_a) >> 16, xs = q - rr * (unsigned)(2 * na);
      const int rk = (int)(rr >> 2), row = (int)(rr & 3u);
      const int y = 2 * (4 * k.I + row) + k.py, x = 8 * k.p0 + 4 * (int)xs;
      if (y < g.H && x < g.W) lp.voff[sl] = 4u * (unsigned)(rk * plane + y * g.W + x);
    } else if (i < Rn) {
      const unsigned q = (unsigned)((i - na) * 64 + lane);
      const unsigned rr = (q * magic_b) >> 16, xs = q - rr * (unsigned)(2 * nb);
      const int rk = (int)(rr >> 2), row = (int)(rr & 3u);
      const int i2 = 4 * k.I - R + 4 * k.a + row, y = 2 * i2 + k.py, x = 8 * k.s0 - 2 * R + 4 * (int)xs;
      if (i2 >= 0 && y < g.H && x >= 0 && x < g.W) lp.voff[sl] = 4u * (unsigned)(rk * plane + y * g.W + x);
    }
  }
}

__device__ __forceinline__ void wait_chunks_but(int chunks, int runs) {    // at most chunks x runs of this wave's LDS-DMA instructions stay outstanding
  const int n = chunks * runs;                              // runs 1 .. 4, chunks 0 .. 3
  switch (n) {
    case 0: wait_vmcnt<0>(); break;  case 1: wait_vmcnt<1>(); break;  case 2: wait_vmcnt<2>(); break;  case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;  case 6: wait_vmcnt<6>(); break;  case 8: wait_vmcnt<8>(); break;  case 9: wait_vmcnt<9>(); break;
    case 12: wait_vmcnt<12>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int TRACE>
__global__ void __launch_bounds__(THREADS)
corr_fwd_persist(const float* __restrict__ b0, const float* __restrict__ b1, float* __restrict__ top, Args g, unsigned long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
  const int n = xcd / g.G, w = (xcd % g.G) * 32 + j;        // sample, workgroup of the sample
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nch = g.C / 8;
  const size_t plane = (size_t)g.H * g.W;
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(top + ((size_t)n * g.ctot + g.c0) * plane, 0, (unsigned)(4u * D * D * (unsigned)plane), 0x00020000);
  const unsigned hw4 = 4u * (unsigned)plane, w4b = 4u * (unsigned)g.W;
  auto row_offset = [&](const TaskInfo& k, int rowid) -> unsigned {
    const int blk = rowid / D, oo = rowid - blk * D, rmi = blk >> 2, rni = blk & 3;
    const int qq = 4 * k.a + rni - rmi, y = 2 * (4 * k.I + rmi) + k.py;
    return (qq >= 0 && qq < D && y < g.H) ? (unsigned)(qq * D + oo) * hw4 + (unsigned)y * w4b : NOROW;
  };
  float* img = smem + RINGF;
  unsigned* rowtab = reinterpret_cast<unsigned*>(smem + RINGF + IMGF);
  unsigned entries[MAXT];                                   // this workgroup's task list, read ONCE (a 16-bit kernarg read is a vector load: inside
  int ntask = 0;                                            // the loops its s_waitcnt vmcnt(0) would drain a loader's whole LDS-DMA queue)
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    entries[t] = (unsigned)__builtin_amdgcn_readfirstlane((int)g.task[w][t]);
    ntask += entries[t] != NOTASK ? 1 : 0;
  }
  [[maybe_unused]] unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (TRACE) tr[0] = __builtin_amdgcn_s_memtime();

  // One 64-lane store instruction's worth of image quads (quad = 16 bytes: pixel pairs r0, r0 + 1 of one patch, both x parities), as the
  // storer -- or, for the last task, every wave -- executes it.  idx = quad index in [0, QUADS): row = idx >> 3, quad of the row = idx & 7.
  auto store_quads = [&](const TaskInfo& k, int idx, float scale, bool relu, float slope) {
    const int rowid = idx >> 3, xq = idx & 7;
    const unsigned off = rowtab[rowid];
    const int x = 8 * k.p0 + 4 * xq;
    if (off == NOROW || xq >= 2 * k.np || x >= g.W) return;
    const unsigned br = (g.seg[k.seg][5] >> (6 * (xq >> 1))) & 63u;
    const int blo = (int)(br & 7u), bhi = (int)(br >> 3), r0 = 2 * (xq & 1);
    const float* src = img + rowid * XS + 4 * xq;
    float f0 = src[0] * scale, f1 = src[1] * scale, f2 = src[2] * scale, f3 = src[3] * scale;
    if (relu) {
      f0 = f0 > 0.f ? f0 : f0 * slope; f1 = f1 > 0.f ? f1 : f1 * slope;
      f2 = f2 > 0.f ? f2 : f2 * slope; f3 = f3 > 0.f ? f3 : f3 * slope;
    }
    const int oo = rowid - (rowid / D) * D;
    const int q0 = (oo + r0) >> 2, q1 = (oo + r0 + 1) >> 2;
    const bool m0 = q0 >= blo && q0 <= bhi, m1 = q1 >= blo && q1 <= bhi;       // element (oo, r) of a patch comes from unit b = (oo + r) >> 2
    const unsigned dst = off + 4u * (unsigned)x;
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
  };
  const bool pow2 = (g.C & (g.C - 1)) == 0;
  const float scale = pow2 ? 1.0f / (float)g.C : 1.0f;
  const bool relu = g.relu != 0;

  if (wave < CONS) {
    // =================================================== consumers ===================================================
    wg_barrier();                                           // P: chunks 0 and 1 of the first task have landed
    for (int t = 0; t < ntask; ++t) {
      const TaskInfo k = decode(g, entries, t);
      const unsigned* sw = g.seg[k.seg];
      const int blk = ((wave + k.rot) & 3) + (wave & 4);    // which share of the task's unit list this wave takes: shares k and 4 + k sit on one SIMD
      const int u0 = (int)((sw[blk < 4 ? 1 : 14] >> (8 * (blk & 3))) & 255u), nu = (int)((sw[blk < 4 ? 2 : 15] >> (8 * (blk & 3))) & 255u);
      int upl[MAXNU], usl[MAXNU];
#pragma unroll
      for (int u = 0; u < MAXNU; ++u) {
        const int kx = u0 + (u < nu ? u : 0);
        const unsigned byte = (sw[6 + (kx >> 2)] >> (8 * (kx & 3))) & 255u;
        upl[u] = (int)(byte >> 4); usl[u] = (int)(byte & 15u);
      }
      switch (nu) {
        case 0: consume_none(g); break;
        case 1: consume<1>(smem, g, lane, wave, k, upl, usl); break;
        case 2: consume<2>(smem, g, lane, wave, k, upl, usl); break;
        default: consume<3>(smem, g, lane, wave, k, upl, usl); break;
      }
      zero_absent(smem, g, k, lane, wave);
      if constexpr (TRACE) { if (1 + 2 * t < 8) tr[1 + 2 * t] = __builtin_amdgcn_s_memtime(); }
      wg_barrier();                                         // E_t: the image of task t is complete
      if constexpr (TRACE) { if (2 + 2 * t < 8) tr[2 + 2 * t] = __builtin_amdgcn_s_memtime(); }
    }
  } else if (wave < CONS + NLOAD) {
    // ===================================================== loaders =====================================================
    // Run i of a chunk belongs to loader i % 3.  The steady-state step is straight-line: MYR LDS-DMA instructions (chunk g + 5), one
    // s_waitcnt, the barrier -- everything it touches sits in registers (the first version of this loop re-derived its state every step
    // and paced the barrier at ~1,050 cycles where the consumers need 560 - 690).
    const int li = wave - CONS;
    const float* a_n = b0 + (size_t)n * g.C * plane;
    const float* b_n = b1 + (size_t)n * g.C * plane;
    const unsigned chunk_bytes = 32u * (unsigned)plane;
    const unsigned sample_bytes = 4u * (unsigned)g.C * (unsigned)plane;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_n), 0, sample_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_n), 0, sample_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int RN = g.RN;
    const int myruns = (RN - li + NLOAD - 1) / NLOAD;       // runs li, li + 3, ... < RN: 1 .. 4
    auto body = [&](auto myr_tag) {
      constexpr int MYR = decltype(myr_tag)::value;
      LoadPlan lp;
      auto dma = [&](int chunk_of_job, int chunk_of_task) {
        const unsigned soff = (unsigned)chunk_of_task * chunk_bytes;
        const unsigned base = lds_base + 4u * (unsigned)((chunk_of_job & (NSLOT - 1)) * SLOTF) + 1024u * (unsigned)li;
#pragma unroll
        for (int sl = 0; sl < MYR; ++sl) {
          lds_ptr_t p = (lds_ptr_t)(uintptr_t)(base + 1024u * (unsigned)(NLOAD * sl));
          if (li + NLOAD * sl < lp.na) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, p, 16, lp.voff[sl], soff, 0, 0);
          else                         __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, p, 16, lp.voff[sl], soff, 0, 0);
        }
      };
      if (g.flags & 4) __builtin_amdgcn_s_setprio(2);       // experiment: loaders ahead of the consumers
      make_plan(lp, g, decode(g, entries, 0), lane, li);
#pragma unroll 1
      for (int c = 0; c < AHEAD; ++c) dma(c, c);            // nch >= 8 > AHEAD
      wait_vmcnt<3 * MYR>();                                // chunks 0 and 1 have landed
      wg_barrier();                                         // P
      for (int t = 0; t < ntask; ++t) {
        const int g0 = t * nch;
#pragma unroll 1
        for (int c = 0; c < nch - AHEAD; ++c) {             // steady state: chunk c + 5 of this task, into the slot chunk c - 3 left
          dma(g0 + c + AHEAD, c + AHEAD);
          wait_vmcnt<3 * MYR>();                            // chunks <= c + 2 have landed
          wg_barrier();                                     // B_g
        }
        if (t + 1 < ntask) {                                // the last five steps of a task fetch the first five chunks of the next one
          make_plan(lp, g, decode(g, entries, t + 1), lane, li);
#pragma unroll 1
          for (int c = nch - AHEAD; c < nch; ++c) {
            dma(g0 + c + AHEAD, c + AHEAD - nch);
            wait_vmcnt<3 * MYR>();
            wg_barrier();
          }
        } else {                                            // end of the job: nothing left to issue, the queue drains
          wait_vmcnt<2 * MYR>(); wg_barrier();
          wait_vmcnt<MYR>(); wg_barrier();
          wait_vmcnt<0>(); wg_barrier();
          for (int c = nch - AHEAD + 3; c < nch; ++c) wg_barrier();
        }
        wg_barrier();                                       // E_t
      }
    };
    switch (myruns) {
      case 1: body(std::integral_constant<int, 1>{}); break;
      case 2: body(std::integral_constant<int, 2>{}); break;
      case 3: body(std::integral_constant<int, 3>{}); break;
      default: body(std::integral_constant<int, 4>{}); break;
    }
  } else {
    // ===================================================== storer =====================================================
    // Per chunk step: `zper` 64-lane store instructions of zeros (first task: the dead N patch rows this workgroup owns, a quarter = one
    // (py, I, a, mi) at a time) and `sper` store instructions of the previous task's image.  Lane -> (row, quad) advances incrementally:
    // no table, no division inside the loops.
    const int si = wave - CONS - NLOAD;                     // storer 0 .. NSTORE - 1: takes every NSTORE-th store instruction
    const int W = g.W, H = g.H, ZP = g.ZP, ND = g.ND;
    const float slope = g.slope;
    const int lpr = W / 4;                                  // 16-byte quads per image row
    const int per_q = 4 * D * lpr;                          // quads of a zero quarter: (4 ni x 21 o) rows
    const int adv_row = (64 * NSTORE) / lpr, adv_x = 64 * NSTORE - adv_row * lpr;
    auto dead_task = [&](int z, int& py, int& I, int& a) -> bool {     // dead (py, I, a) row task number z of the sample: scalar walk
      const int NI = ((H + 1) / 2 + 3) / 4;
      for (py = 0; py < 2; ++py) {
        const int Hc = (H - py + 1) / 2;
        for (I = 0; I < NI; ++I) {
          if (4 * I >= Hc) continue;
          const int lo_num = R - 3 - 4 * I, hi_num = Hc - 1 + R - 4 * I;
          const int alo = lo_num <= 0 ? 0 : (lo_num + 3) / 4;
          int ahi = hi_num < 0 ? -1 : hi_num / 4;
          if (ahi > NBT - 1) ahi = NBT - 1;
          const int nlive = ahi >= alo ? ahi - alo + 1 : 0, ndead = NBT - nlive;
          if (z < ndead) { a = (nlive == 0 || z < alo) ? z : z + nlive; return true; }
          z -= ndead;
        }
      }
      return false;
    };
    int zq = 0, zleft = 0;                                  // current zero quarter, store instructions left in it
    int zrow = 0, zx = 0;                                   // this lane's (row of the quarter, quad of the row)
    unsigned zbase = 0; int zqq0 = 0; bool zok = false;
    const int zinstr_per_q = (per_q + 64 * NSTORE - 1) / (64 * NSTORE);      // store instructions of THIS storer per quarter
    auto zero_step = [&]() -> bool {                        // one 64-lane store instruction of zeros; false when nothing is left
      if (zleft == 0) {
        if (zq >= ZP) return false;
        const int piece = w * ZP + zq;                      // quarter number inside the sample: (dead row task, mi)
        int zpy = 0, zI = 0, za = 0;
        const int zmi = piece & 3;
        zok = piece < 4 * ND && dead_task(piece >> 2, zpy, zI, za);
        const int y = 2 * (4 * zI + zmi) + zpy;
        zok = zok && y < H;
        zqq0 = 4 * za - zmi;                                // qq = zqq0 + rni
        zbase = (unsigned)y * w4b;
        zrow = (lane + 64 * si) / lpr; zx = lane + 64 * si - zrow * lpr;
        zleft = zinstr_per_q;
        ++zq;
      }
      if (zok && zrow < 4 * D) {
        const int rni = (zrow >= D) + (zrow >= 2 * D) + (zrow >= 3 * D), oo = zrow - rni * D, qq = zqq0 + rni;
        if (qq >= 0 && qq < D) {
          const u32x4 z4 = {0u, 0u, 0u, 0u};
          __builtin_amdgcn_raw_buffer_store_b128(z4, rsT, (unsigned)(qq * D + oo) * hw4 + zbase + 16u * (unsigned)zx, 0, 0);
        }
      }
      zrow += adv_row; zx += adv_x;
      if (zx >= lpr) { zx -= lpr; ++zrow; }
      --zleft;
      return true;
    };
    // image of the previous task: this lane's quad column is fixed (xq = lane & 7), its row advances by 8 per instruction
    const int xq = lane & 7, r0 = 2 * (xq & 1);
    int s_left = 0, s_oo = 0, s_blk = 0;                    // store instructions left; (oo, blk) of this lane's current row
    const float* s_src = img;
    bool s_on = false, s_whole = true; int s_blo = 0, s_bhi = NBT - 1; unsigned s_xoff = 0;
    int s_a4 = 0, s_y0 = 0;                                 // 4 a and 2 (4 I) + py of the task being stored
    auto begin_image = [&](const TaskInfo& k) {
      const unsigned br = (g.seg[k.seg][5] >> (6 * (xq >> 1))) & 63u;
      s_blo = (int)(br & 7u); s_bhi = (int)(br >> 3);
      s_whole = s_blo == 0 && s_bhi == NBT - 1;
      const int x = 8 * k.p0 + 4 * xq;
      s_on = xq < 2 * k.np && x < W;
      s_xoff = 4u * (unsigned)x;
      s_a4 = 4 * k.a; s_y0 = 8 * k.I + k.py;
      const int row = (lane >> 3) + 8 * si;                 // < D
      s_oo = row; s_blk = 0;
      s_src = img + row * XS + 4 * xq;
      s_left = QUADS / 64 / NSTORE;
    };
    // One 64-lane store instruction of image quads, in two halves so that a batch can have all its LDS reads in flight before the first
    // value is needed (the storer is ONE wave: done one after the other, two instructions per chunk step cost ~600 cycles of latency and paced
    // the whole workgroup's barrier).
    struct Quad { float f[4]; int oo, blk; };
    auto image_load = [&](Quad& q) {
      q.f[0] = s_src[0]; q.f[1] = s_src[1]; q.f[2] = s_src[2]; q.f[3] = s_src[3];
      q.oo = s_oo; q.blk = s_blk;
      s_oo += 8 * NSTORE; s_src += 8 * NSTORE * XS;
      if (s_oo >= D) { s_oo -= D; ++s_blk; }
      --s_left;
    };
    auto image_emit = [&](const Quad& q) {
      const int rmi = q.blk >> 2, rni = q.blk & 3;
      const int qq = s_a4 + rni - rmi, y = s_y0 + 2 * rmi;
      if (s_on && qq >= 0 && qq < D && y < H) {
        float f0 = q.f[0] * scale, f1 = q.f[1] * scale, f2 = q.f[2] * scale, f3 = q.f[3] * scale;
        if (relu) {
          f0 = f0 > 0.f ? f0 : f0 * slope; f1 = f1 > 0.f ? f1 : f1 * slope;
          f2 = f2 > 0.f ? f2 : f2 * slope; f3 = f3 > 0.f ? f3 : f3 * slope;
        }
        bool m0 = true, m1 = true;
        if (!s_whole) {
          const int q0 = (q.oo + r0) >> 2, q1 = (q.oo + r0 + 1) >> 2;
          m0 = q0 >= s_blo && q0 <= s_bhi; m1 = q1 >= s_blo && q1 <= s_bhi;
        }
        const unsigned dst = (unsigned)(qq * D + q.oo) * hw4 + (unsigned)y * w4b + s_xoff;
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
    };
    auto image_one = [&]() { Quad qa; image_load(qa); image_emit(qa); };
    static_assert((QUADS / 64) % NSTORE == 0 && 8 * NSTORE < D, "the storers split the image's store instructions evenly");
    if (g.flags & 2) __builtin_amdgcn_s_setprio(3);         // experiment: the storer's VALU / LDS / store instructions ahead of the consumers' MFMAs
    wg_barrier();                                           // P
    const int zsteps = nch > 3 ? nch - 2 : 1;
    const int zper = (ZP * zinstr_per_q + zsteps - 1) / zsteps;        // zero store instructions per step: all of them inside the first task's K loop
    const int ssteps = nch > 7 ? nch - 6 : 1;
    const int sper = (QUADS / 64 / NSTORE + ssteps - 1) / ssteps;     // image store instructions per step and storer: done 6 steps before the next scatter
    const bool idle = (g.flags & 1) != 0;
    for (int t = 0; t < ntask; ++t) {
#pragma unroll 1
      for (int c = 0; c < nch; ++c) {
        if (!idle) {
          if (t == 0) { for (int sx = 0; sx < zper; ++sx) zero_step(); }
          for (int sx = 0; sx < sper; ++sx) if (s_left > 0) image_one();
        }
        wg_barrier();                                       // B_g
      }
      while (zero_step()) {}                                // (only when the first task is too short to hide them)
      while (s_left > 0) image_one();
      wg_barrier();                                         // E_t
      if (t + 1 < ntask) begin_image(decode(g, entries, t));      // the image of task t goes out during task t + 1; the last one is stored by everybody
    }
  }
  // ===================================================== last image: all waves =====================================================
  if (ntask > 0) {
    const TaskInfo last = decode(g, entries, ntask - 1);
    for (int rowid = tid; rowid < OROWS; rowid += THREADS) rowtab[rowid] = row_offset(last, rowid);
    __syncthreads();
    for (int idx = tid; idx < QUADS; idx += THREADS) store_quads(last, idx, scale, relu, g.slope);
  }
  if constexpr (TRACE) {
    if (dbg && lane == 0 && wave == 0) {
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long* rec = dbg + 10 * blockIdx.x;
#pragma unroll
      for (int i = 0; i < 8; ++i) rec[i] = tr[i];
      rec[8] = __builtin_amdgcn_s_memtime();
      rec[9] = (unsigned long long)hwid | ((unsigned long long)(xcc & 15u) << 32) | (1ull << 63);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Host: the task lists.  The (patch, b) sequence of an image row is the one of correlation_units.hip; a workgroup of the sample gets the
// units [w U, (w + 1) U) of the rows laid end to end (U = units of the sample / workgroups), cut at row ends and into tasks of <= 20
// units, sizes a multiple of four where the cut allows it.
struct Plan { int N, H, W; bool ok; Args a; };

static void live_range(int I, int Hc, int& alo, int& ahi) {
  const int lo_num = R - 3 - 4 * I;
  alo = lo_num <= 0 ? 0 : (lo_num + 3) / 4;
  const int hi_num = Hc - 1 + R - 4 * I;
  ahi = hi_num < 0 ? -1 : hi_num / 4;
  if (ahi > NBT - 1) ahi = NBT - 1;
  if (4 * I >= Hc) { alo = 0; ahi = -1; }
}

// segment words for the live units [ua, ub) of an image row (sequence positions found by counting live entries)
static bool make_segment(const std::vector<char>& livek, int NP, int smax, int ua, int ub, int U_row, unsigned* w) {
  const int L = NBT * NP;
  int k0 = -1, k1 = -1, cnt = 0;
  for (int k = 0; k < L; ++k) {
    if (livek[k]) {
      if (cnt == ua && k0 < 0) k0 = k;
      ++cnt;
      if (cnt == ub) { k1 = k + 1; break; }
    }
  }
  if (k0 < 0 || k1 < 0) return false;
  // absent entries: leading ones of the first patch belong to the segment that holds the patch's first live b (ua == 0 or a patch start);
  // trailing ones of a patch stay with the segment that holds its last live b
  while (k0 > 0 && !livek[k0 - 1] && (k0 % NBT) != 0) --k0;
  if (ua == 0) k0 = 0;
  while (k1 < L && !livek[k1] && (k1 % NBT) != 0) ++k1;
  if (ub == U_row) k1 = L;
  const int pf = k0 / NBT, plast = (k1 - 1) / NBT, np = plast - pf + 1;
  if (np > MAXPATCH) return false;
  std::memset(w, 0, sizeof(unsigned) * SEGW);
  int s_lo = 1 << 30, s_hi = -1, U = 0, nmiss = 0;
  for (int k = k0; k < k1; ++k) if (livek[k]) { const int sx = k / NBT + k % NBT; s_lo = std::min(s_lo, sx); s_hi = std::max(s_hi, sx); }
  if (s_hi < s_lo) return false;
  int nb = s_hi - s_lo + 1;
  if (nb % 2 == 0) ++nb;
  const int na = np | 1;
  if (na + nb > MAXRUNS) return false;
  for (int k = k0; k < k1; ++k) {
    const int q = k / NBT, b = k % NBT;
    if (livek[k]) { if (U >= MAXU) return false; w[6 + (U >> 2)] |= (unsigned)(((q - pf) << 4) | (q + b - s_lo)) << (8 * (U & 3)); ++U; }
    else { if (nmiss >= MAXMISS) return false; w[11 + (nmiss >> 2)] |= (unsigned)(((q - pf) << 4) | b) << (8 * (nmiss & 3)); ++nmiss; }
  }
  if (U != ub - ua || U > MAXU) return false;
  w[0] = (unsigned)pf | ((unsigned)np << 8) | ((unsigned)s_lo << 16) | ((unsigned)nb << 24);
  // eight shares: base = U / 8, the remainder goes to shares 7, 6, 5, 4, then 3, 2, 1, 0 -- shares k and 4 + k land on one SIMD, so the pair
  // sums differ by at most one; the task's rotation decides which SIMD takes which pair
  int u0 = 0;
  for (int sh = 0; sh < CONS; ++sh) {
    const int rem = U % CONS, order = sh >= 4 ? 7 - sh : 3 - sh + 4;       // how early this share is served by the remainder
    const int nu = U / CONS + (order < rem ? 1 : 0);
    if (nu > MAXNU) return false;
    w[sh < 4 ? 1 : 14] |= (unsigned)u0 << (8 * (sh & 3));
    w[sh < 4 ? 2 : 15] |= (unsigned)nu << (8 * (sh & 3));
    u0 += nu;
  }
  const unsigned db = 2u * (unsigned)nb, mb = 65536u / db + 1u, da = 2u * (unsigned)na, ma = 65536u / da + 1u;
  for (unsigned q = 0; q < 64u * (unsigned)nb; ++q) if (((q * mb) >> 16) != q / db) return false;
  for (unsigned q = 0; q < 64u * (unsigned)na; ++q) if (((q * ma) >> 16) != q / da) return false;
  w[3] = mb | ((unsigned)nmiss << 16);
  w[4] = ma | ((unsigned)na << 16);
  for (int q = pf; q <= plast; ++q) {
    const int blo = std::max(k0, q * NBT) - q * NBT, bhi = std::min(k1, (q + 1) * NBT) - 1 - q * NBT;
    w[5] |= (unsigned)(blo | (bhi << 3)) << (6 * (q - pf));
  }
  return true;
}

static bool build_plan(int N, int H, int W, Plan& pl) {
  pl.N = N; pl.H = H; pl.W = W; pl.ok = false;
  Args& g = pl.a;
  std::memset(&g, 0, sizeof(g));
  std::memset(g.task, 0xff, sizeof(g.task));
  if (W % 4 != 0 || H < 1 || H > 255 || (N != 1 && N != 2 && N != 4 && N != 8)) return false;
  const int Wc = W / 2, NP = (Wc + 3) / 4, smax = (Wc - 1 + R) / 4;
  if (NP > 64) return false;
  struct Row { int py, I, a; };
  std::vector<Row> rows;
  int ND = 0;
  const int NI = ((H + 1) / 2 + 3) / 4;
  if (NI > 32) return false;
  for (int py = 0; py < 2; ++py) {
    const int Hc = (H - py + 1) / 2;
    for (int I = 0; I < NI; ++I) {
      if (4 * I >= Hc) continue;
      int alo, ahi;
      live_range(I, Hc, alo, ahi);
      for (int a = 0; a < NBT; ++a) { if (a >= alo && a <= ahi) rows.push_back({py, I, a}); else ++ND; }
    }
  }
  const int L = NBT * NP;
  std::vector<char> livek(L);
  int U_row = 0;
  for (int k = 0; k < L; ++k) { const int s = k / NBT + k % NBT; livek[k] = (s >= 2 && s <= smax); U_row += livek[k]; }
  if (rows.empty() || U_row == 0) return false;
  const int G = 8 / N, NW = 32 * G;                         // workgroups of a sample
  const long long total = (long long)rows.size() * U_row;
  if (total < 4LL * NW) return false;                       // too little work for a persistent launch
  std::map<std::pair<int, int>, int> seg_of;                // (ua, ub) of a row -> segment number
  int nseg = 0, RN = 0;
  std::vector<int> wave_load(4);                            // units per SIMD (consumer waves sd, sd + 4) of the workgroup being filled
  for (int wg = 0; wg < NW; ++wg) {
    const long long lo = total * wg / NW, hi = total * (wg + 1) / NW;
    std::fill(wave_load.begin(), wave_load.end(), 0);
    int nt = 0;
    for (long long pos = lo; pos < hi;) {
      const int r = (int)(pos / U_row), ua = (int)(pos % U_row);
      const int ub_row = (int)std::min<long long>(U_row, ua + (hi - pos));
      // cut [ua, ub_row) into tasks of <= MAXU units: as few as possible, sizes a multiple of 4 where that works
      int left = ub_row - ua, a0 = ua;
      while (left > 0) {
        int pieces = (left + MAXU - 1) / MAXU;
        int sz = (left + pieces - 1) / pieces;
        if (pieces > 1) { sz = std::min(MAXU, (sz + 3) / 4 * 4); }
        sz = std::min(sz, left);
        unsigned words[SEGW];
        // shrink until the segment fits (patches, staged tiles)
        while (sz > 0 && !make_segment(livek, NP, smax, a0, a0 + sz, U_row, words)) --sz;
        if (sz == 0) return false;
        auto key = std::make_pair(a0, a0 + sz);
        auto it = seg_of.find(key);
        int si;
        if (it == seg_of.end()) {
          if (nseg >= MAXSEG) return false;
          si = nseg++;
          seg_of[key] = si;
          std::memcpy(g.seg[si], words, sizeof(words));
          RN = std::max(RN, (int)((words[4] >> 16) & 15u) + (int)(words[0] >> 24));
        } else si = it->second;
        if (nt >= MAXT) return false;
        // rotation: pick the one that keeps the waves' totals level; a tie goes to waves 0, 1 (wave 3 shares its SIMD with the storer)
        int best_rot = 0, best_cost = 1 << 30;
        auto pair_units = [&](int k) { return (int)((g.seg[si][2] >> (8 * k)) & 255u) + (int)((g.seg[si][15] >> (8 * k)) & 255u); };
        for (int rot = 0; rot < 4; ++rot) {
          int mx = 0, tie = 0;
          for (int sd = 0; sd < 4; ++sd) {                  // sd: consumer waves sd and sd + 4 (one SIMD)
            const int ld = wave_load[sd] + pair_units((sd + rot) & 3);
            mx = std::max(mx, ld);
            tie += ld * (sd == 3 ? 3 : 2);                  // a level tie goes away from the storer's SIMD (waves 3, 7, 11)
          }
          const int cost = mx * 1000 + tie;
          if (cost < best_cost) { best_cost = cost; best_rot = rot; }
        }
        for (int sd = 0; sd < 4; ++sd) wave_load[sd] += pair_units((sd + best_rot) & 3);
        const Row& rw = rows[r];
        g.task[wg][nt++] = (unsigned short)(rw.py | (rw.I << 1) | (rw.a << 6) | (si << 9) | (best_rot << 13));
        a0 += sz; left -= sz; pos += sz;
      }
    }
  }
  g.N = N; g.H = H; g.W = W; g.G = G; g.RN = RN; g.ND = ND;
  g.ZP = (4 * ND + NW - 1) / NW;
  pl.ok = true;
  return true;
}

static const Plan& plan_for(int N, int H, int W) {
  static std::mutex mu;
  static std::vector<Plan*> cache;
  std::lock_guard<std::mutex> lock(mu);
  for (const Plan* p : cache)
    if (p->N == N && p->H == H && p->W == W) return *p;
  Plan* p = new Plan;
  build_plan(N, H, W, *p);
  cache.push_back(p);                                       // a handful of geometries per process; never evicted (references stay valid)
  return *p;
}

}  // namespace cu4

extern unsigned long long* g_corr_dbg;
int g_corr_persist_flags = 0;  // experiment hook (wrong results): bit 0 the storer stays idle inside the K loops, bit 1 no LDS-DMA
int g_corr_persist = 0;        // test / profiling hook (fn2_debug_set_correlation_impl(17)): 1 = run the persistent kernel where it has a plan

bool corr_fwd_persist_supported(const CorrGeom& g, const float* b0, const float* b1, const float* top) {
  if (g_corr_persist == 0) return false;
  if (g.K != 1 || g.s1 != 1 || g.type != FN2_CORR_MULTIPLY || g.pad != g.md || g.s2 != 2 || g.ngr != cu4::R) return false;
  if (g.C % 64 != 0 || g.W % 4 != 0) return false;          // whole turns of the eight-chunk ring per task
  if ((long long)g.C * g.H * g.W >= (1ll << 28) || (long long)g.topC * g.H * g.W >= (1ll << 30)) return false;
  if (((reinterpret_cast<uintptr_t>(b0) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(top)) & 15) != 0) return false;
  return cu4::plan_for(g.N, g.H, g.W).ok;
}

int corr_fwd_persist_plan_words(int N, int H, int W, unsigned* out, int max_words) {
  const cu4::Plan& pl = cu4::plan_for(N, H, W);
  if (!pl.ok) return 0;
  const int nw = (int)(sizeof(cu4::Args) / 4);
  if (out && max_words >= nw) std::memcpy(out, &pl.a, sizeof(cu4::Args));
  return nw;
}

int corr_fwd_persist_launch(const CorrGeom& cg, const float* b0, const float* b1, float* top, hipStream_t st) {
  const cu4::Plan& pl = cu4::plan_for(cg.N, cg.H, cg.W);
  if (!pl.ok) return fail(FN2_ERR_UNSUPPORTED, "correlation: no persistent plan for %d x %d x %d", cg.N, cg.H, cg.W);
  cu4::Args a = pl.a;
  a.C = cg.C; a.ctot = cg.top_ctot; a.c0 = cg.top_c0; a.relu = cg.relu; a.slope = cg.slope;
  a.flags = g_corr_persist_flags;
  const size_t lds = sizeof(float) * cu4::LDS_FLOATS;
  auto go = [&](auto kernel) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return fail(FN2_ERR_UNSUPPORTED, "correlation_forward (persistent): %zu bytes of dynamic LDS refused by the runtime", lds);
    }
    hipLaunchKernelGGL(kernel, dim3(256), dim3(cu4::THREADS), lds, st, b0, b1, top, a, g_corr_dbg);
    return check_launch("correlation_forward (mfma, persistent)");
  };
#ifdef FN2_ABLATION
  if (g_corr_dbg) return go(&cu4::corr_fwd_persist<1>);
#endif
  return go(&cu4::corr_fwd_persist<0>);
}

}  // namespace fn2
