// 3x3 / stride 1 convolution + bias + ReLU as Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate).
//
// Reference: ConvolutionLayer::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23: im2col + cublasSgemm per sample, then
// forward_gpu_bias, base_conv_layer.cpp:326-348) + the in-place ReLU (relu_layer.cu:8-27), for kernel_size 3, stride 1.
// Same contraction with 2.25x fewer multiplies (Lavin & Gray's minimal filtering, the algorithm MIOpen's fp32 Winograd kernels
// and cuDNN use for this layer class):
//     Y = A^T [ sum_c (G g_c G^T) (.) (B^T d_c B) ] A          per 2x2 output tile, d = the 4x4 input patch under it
// The 16 element-wise products are 16 independent [tiles x channels] x [channels x Cout] GEMMs -- one MFMA accumulator tile per
// Winograd position:  rows = 16 output TILES (a TGY x TGX block of 2x2 tiles), columns = 16 output channels, k = 4 channels.
//   * d: the input window of the workgroup tile arrives by 16-byte LDS-DMA in natural [channel][row][column] order (zero padding
//     = out-of-range for the buffer descriptor); every lane (tile, kq) reads its 4x4 patch of channel 4 cq + kq as aligned
//     8-byte pairs and applies B^T . B in registers (32 add / sub) -> its 16 A operands.
//   * U = G g G^T is precomputed once per weight blob (fn2_conv_wino_pack_weights) in MFMA operand order
//     [Cout/16][channel quad][position quad][lane][4] and staged through LDS next to the window (one 4 KiB slab per channel
//     quad, shared by the 4 waves of a workgroup, which own different pixels of the same 16 output channels).
//   * the result layout hands every lane the 16 positions of 4 horizontally adjacent tiles of one channel: A^T . A in registers,
//     bias + ReLU, two 16-byte stores per output row.
// Numerics: exact fp32 products and k-ordered fma chains per position; the transforms add rounding of a few ulp of the largest
// patch element (measured against the reference's Convolution layer and fp64 in tests/test_conv_wino.py).  The oracle twin
// (oracle/fn2_oracle.c: fn2_conv_wino_forward_cpu) performs the same operations in the same order: bit-identical.
#include "fn2_common.hpp"
#include "autotune.hpp"

namespace fn2 {
namespace wino {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int up_mod(int v, int r, int m) { return v + ((r - v % m) + m) % m; }

struct Args {
  const float* in; const float* up; const float* bias; float* out;
  int N, Cin, Hin, Win, in_ctot, in_c0;
  int Cout, Hout, Wout, out_ctot, out_c0;
  int nchunks, kquads;     // chunks of CQ channel quads; quads per output-channel group in the packed weights (incl. padding)
  int tx, ty, ng;          // workgroup tiles per sample along x / y; output-channel groups (Cout / 16)
  unsigned total, nbig;    // tiles; split-tail launch: tiles that run whole (multiple of 256), the rest as two x-halves (MT / 2)
  float slope; int relu;
};

constexpr int kCQ = 2;     // channel quads per chunk (weights are padded to whole chunks)
constexpr int kPad = 1;    // the only padding instantiated (3x3 "same" convolution)

template <int TGY_, int TGX_, int MT_, int WNY_, int WNX_, int WM_ = 1, int MW_ = 1>
struct Cfg {
  static constexpr int TGY = TGY_, TGX = TGX_, MT = MT_, WNY = WNY_, WNX = WNX_, WM = WM_, MW = MW_, CQ = kCQ;      // MW: 16-channel groups per WAVE (they share the transformed patch)
  static constexpr int NW = WNY * WNX * WM, THREADS = 64 * NW;      // WM waves work on WM different 16-channel groups of the same pixels
  static constexpr int PADL = 4;
  static constexpr int TH = 2 * TGY * WNY, TW = 2 * TGX * MT * WNX;     // output pixels of a workgroup tile
  static constexpr int WR = TH + 2, WC = TW + 2 + PADL;                 // window rows / columns (columns from x0 - PADL)
  // Row / channel strides of the window.  The conflict-free choice (RS == 4 or 8 mod 16, CS == 32 mod 64) pads a 38-column row to 52:
  // the kernel is bound by what it stages per MFMA (24 B/clk/CU through L2 at 3 workgroups per CU), not by LDS cycles (28 % busy), so the
  // rows are kept dense and the pair reads take their 2-way conflicts.
  static constexpr int RS = cdiv(WC, 4) * 4;
  static constexpr int CS = cdiv(WR * RS, 8) * 8;
  static constexpr int URUN = 4 * CQ * WM * MW;                              // 1 KiB runs of U per chunk (4 per channel quad and channel group)
  static constexpr int WSLOTS = 4 * CQ * CS / 4;
  static constexpr int WRUN = cdiv(WSLOTS, 64);
  static constexpr int NRUN = URUN + WRUN;
  static constexpr int RPW = cdiv(NRUN, NW);
  static constexpr int BUF = NRUN * 256;                                // dwords per buffer: [U | window]
  static constexpr int WOFF = URUN * 256;
  // The software-pipelined variants (MT == 1, MW == 1) run 3 waves per SIMD under a 168-VGPR cap; with the per-lane DMA offsets in
  // registers the loop spilled them, and every reload in front of a buffer_load ... lds drained the vector-memory queue
  // (s_waitcnt vmcnt(0)).  Their offsets live in an LDS table behind the two buffers instead: one ds_read_b32 per DMA instruction.
  // (Only the window runs need a table: a U run is a linear copy, lane * 16 bytes behind a scalar offset.)
  static constexpr bool VOFF_LDS = MT_ == 1 && MW_ == 1;
  static constexpr int UI = URUN / NW;                                    // the first UI DMA instructions of every wave are U runs
  static_assert(URUN % NW == 0, "U runs are dealt evenly");
  static constexpr int LDS_FLOATS = 2 * BUF + (VOFF_LDS ? (RPW - UI) * THREADS : 0);
  static_assert((NW == 4 || NW == 8) && TGY * TGX == 16 && LDS_FLOATS * 4 <= 160 * 1024, "workgroup shape");
};

// LDS-DMA of one chunk: runs [0, URUN) = the chunk's U slab (a linear copy), runs [URUN, NRUN) = the input window.
// (A __device__ function, not a lambda: the host pass of a __global__ template cannot see amdgcn builtins inside a lambda body.)
template <class K>
__device__ __forceinline__ void stage_chunk(__amdgpu_buffer_rsrc_t rsU, __amdgpu_buffer_rsrc_t rsW, const unsigned (&voff)[K::RPW], unsigned dst,
                                            int wave, unsigned soff_u, unsigned soff_w) {
#pragma unroll
  for (int i = 0; i < K::RPW; ++i) {
    const int r = i * K::NW + wave;
    if (r < K::NRUN) {
      lds_ptr_t lp = (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)r);
      if (r < K::URUN) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, lp, 16, voff[i], soff_u, 0, 0);
      else             __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lp, 16, voff[i], soff_w, 0, 0);
    }
  }
}

// the same with the per-lane offsets of the window runs in an LDS table (entry i - UI of this thread at tab[(i - UI) * THREADS])
template <class K>
__device__ __forceinline__ void stage_chunk_tab(__amdgpu_buffer_rsrc_t rsU, __amdgpu_buffer_rsrc_t rsW, const unsigned* tab, unsigned lane16, int kquads,
                                                unsigned dst, int wave, unsigned soff_u, unsigned soff_w) {
#pragma unroll
  for (int i = 0; i < K::RPW; ++i) {
    const int r = i * K::NW + wave;
    if (r < K::NRUN) {
      lds_ptr_t lp = (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)r);
      if (i < K::UI) {
        const unsigned so = soff_u + 4096u * (unsigned)((r / (4 * K::CQ)) * kquads) + 1024u * (unsigned)(r % (4 * K::CQ));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, lp, 16, lane16, so, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, lp, 16, tab[(i - K::UI) * K::THREADS], soff_w, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Operands of one k-step (channel quad) of one tile block as they come out of LDS: U for the 16 positions, and the 4x4 input patch
// d[i][0..3] of row i as a dword, the ALIGNED 8-byte pair m[i] = (d[i][1], d[i][2]) and a dword (window columns dcol + 1 .. dcol + 4, dcol
// even).  The inner columns are transformed with packed-fp32 instructions on the pairs (two additions per lane and instruction); the
// matrix pipe and the vector ALU do not overlap on this chip, every VALU instruction in the k loop is matrix time lost.
// (Pairs of NEIGHBOURS (d0, d1), (d2, d3) would make both stages packed -- 16 instructions per patch -- but they sit at odd dword
// addresses: the misaligned 8-byte LDS reads ran every variant 1.6-2.7x slower; measured in round 4 and dropped.  hipcc's own vectoriser
// found some pairs in the all-scalar form of rounds 2-3 and paid for them with up to 20 v_mov_b32 per patch to assemble the operands.)
struct WOps { f32x4 u[4]; float d0[4]; f32x2 m[4]; float d3[4]; };

template <class K, class P>
__device__ __forceinline__ void wino_load_patch(WOps& o, P dp) {           // dp: window column dcol of the lane's first patch row
#pragma unroll
  for (int i = 0; i < 4; ++i) {           // patch columns 0 .. 3 = window columns dcol + 1 .. dcol + 4: dword, aligned pair, dword
    o.d0[i] = dp[i * K::RS + 1];
    o.m[i] = *reinterpret_cast<const f32x2*>(dp + i * K::RS + 2);
    o.d3[i] = dp[i * K::RS + 4];
  }
}

template <class K>
__device__ __forceinline__ void wino_load(WOps& o, const float* ub, const float* dp) {
#pragma unroll
  for (int pq = 0; pq < 4; ++pq) o.u[pq] = *reinterpret_cast<const f32x4*>(ub + pq * 256);
  wino_load_patch<K>(o, dp);
}

// V = B^T d B, rows first.  Per element the operations of the scalar form, in its order (the oracle twin's):
//   w[0][j] = d[0][j] - d[2][j], w[1][j] = d[1][j] + d[2][j], w[2][j] = d[2][j] - d[1][j], w[3][j] = d[1][j] - d[3][j]
//   v[i][0] = w[i][0] - w[i][2], v[i][1] = w[i][1] + w[i][2], v[i][2] = w[i][2] - w[i][1], v[i][3] = w[i][1] - w[i][3]
// The inner columns j = 1, 2 travel as the pair M = (w[i][1], w[i][2]): stage one is four packed additions; in stage two
// (v[i][1], v[i][2]) = (M.x + M.y, M.y - M.x) is ONE v_pk_fma_f32 (M.x, M.x) * (1, -1) + (M.y, M.y) -- x * 1 + y and x * -1 + y are the
// IEEE sum and difference, bit for bit -- with the lane selections in op_sel and the constant in scalar registers (the compiler's own
// instruction selection: no inline assembly, so its hazard handling between the vector ALU and the matrix pipe stays in charge).
// 24 VALU instructions per patch instead of 32 + operand moves.  v[4 i + j] is the A operand of Winograd position (i, j).
__device__ __forceinline__ void wino_transform(const WOps& o, float (&v)[16]) {
  const f32x2 pm = {1.0f, -1.0f};
  f32x2 wm[4];
  float w0[4], w3[4];
  wm[0] = o.m[0] - o.m[2]; w0[0] = o.d0[0] - o.d0[2]; w3[0] = o.d3[0] - o.d3[2];
  wm[1] = o.m[1] + o.m[2]; w0[1] = o.d0[1] + o.d0[2]; w3[1] = o.d3[1] + o.d3[2];
  wm[2] = o.m[2] - o.m[1]; w0[2] = o.d0[2] - o.d0[1]; w3[2] = o.d3[2] - o.d3[1];
  wm[3] = o.m[1] - o.m[3]; w0[3] = o.d0[1] - o.d0[3]; w3[3] = o.d3[1] - o.d3[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 inner = __builtin_elementwise_fma(__builtin_shufflevector(wm[i], wm[i], 0, 0), pm, __builtin_shufflevector(wm[i], wm[i], 1, 1));
    v[4 * i + 0] = w0[i] - wm[i][1];
    v[4 * i + 1] = inner[0];
    v[4 * i + 2] = inner[1];
    // w[i][1] - w[i][3], written as w[i][3] * -1 + w[i][1] (the same IEEE difference): a subtraction here would look like the twin of
    // the one two lines up to hipcc's vectoriser, which then packs the two at the price of four v_mov_b32 to assemble its operands
    v[4 * i + 3] = __builtin_fmaf(w3[i], -1.0f, wm[i][0]);
  }
}

__device__ __forceinline__ void wino_mfma(f32x4 (&acc)[16], const float (&v)[16], const f32x4 (&u)[4]) {
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[p], u[p >> 2][p & 3], acc[p], 0, 0, 0);
}

// One workgroup tile: 16 output channels (group g) x TH x TW pixels of sample n.
template <class K>
__device__ __forceinline__ void wino_body(const Args& a, int g, int bx, int by, int n) {
  constexpr int MT = K::MT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % K::WM, wnx = (wave / K::WM) % K::WNX, wny = wave / (K::WM * K::WNX);
  const int x0 = bx * K::TW, y0 = by * K::TH;

  const size_t plane = (size_t)a.Hin * a.Win;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + ((size_t)n * a.in_ctot + a.in_c0) * plane), 0, (unsigned)(4u * a.Cin * plane), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.up + (size_t)g * K::WM * K::MW * a.kquads * 1024), 0, (unsigned)(4096u * a.kquads * K::WM * K::MW), 0x00020000);
  constexpr unsigned OOB = 0x7ffffff0u;
  unsigned voff[K::RPW];
#pragma unroll
  for (int i = 0; i < K::RPW; ++i) {
    const int r = i * K::NW + wave;
    voff[i] = OOB;
    if (r < K::URUN) voff[i] = 4096u * (unsigned)((r / (4 * K::CQ)) * a.kquads) + 16u * (unsigned)((r % (4 * K::CQ)) * 64 + lane);   // slab r / (4 CQ) of the WM channel groups
    else if (r < K::NRUN) {
      const int s = (r - K::URUN) * 64 + lane;
      const int c = s / (K::CS / 4), rem = s % (K::CS / 4);
      const int row = rem / (K::RS / 4), gq = rem % (K::RS / 4);
      const int yi = y0 - kPad + row, xi = x0 - K::PADL + 4 * gq;
      if (c < 4 * K::CQ && row < K::WR && 4 * gq < K::WC && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win)
        voff[i] = 4u * (unsigned)(c * plane + (size_t)yi * a.Win + xi);
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const unsigned chunk_w = 16u * K::CQ * (unsigned)plane, chunk_u = 4096u * K::CQ;

  // ---- operands: lane (tile t16 = (ty, tx) of the TGY x TGX block, kq)
  const int kq = lane >> 4, t16 = lane & 15;
  const int ty = K::TGX == 8 ? t16 >> 3 : t16 >> 2, tx = K::TGX == 8 ? t16 & 7 : t16 & 3;
  // patch column j of this lane <-> window column 2 * (global tile column) + PADL - pad + j; dcol = that for j = -1: even,
  // so the patch is the middle four of the three aligned pairs (dcol, +1), (+2, +3), (+4, +5)
  const int dcol = 2 * (K::TGX * MT * wnx + tx) + K::PADL - kPad - 1;
  const int dbase = K::WOFF + kq * K::CS + (2 * (K::TGY * wny + ty)) * K::RS + dcol;

  constexpr int MW = K::MW;
  f32x4 acc[MT * MW][16];                     // [tile block m][channel group j] -> index m * MW + j
#pragma unroll
  for (int m = 0; m < MT * MW; ++m)
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[m][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_chunk<K>(rsU, rsW, voff, lds_base, wave, 0u, 0u);
  if constexpr (MT == 1 && MW == 1 && K::CQ == 2) {
    unsigned* vtab = reinterpret_cast<unsigned*>(smem + 2 * K::BUF) + tid;      // this thread's column of the DMA offset table
#pragma unroll
    for (int i = K::UI; i < K::RPW; ++i) vtab[(i - K::UI) * K::THREADS] = voff[i];
    // Software pipeline over the k-steps (two per chunk), one barrier per chunk:
    //   the MFMAs of k-step k run while the operands of k+1 -- read from LDS one step earlier -- are transformed (independent
    //   VALU work in the MFMA issue gaps), and the LDS reads of k+2 are in flight.  The barrier that publishes chunk c+1 and frees
    //   chunk c's buffer sits between the two k-steps of chunk c, when every wave holds (c, q1) in registers.
    const float* ub0 = smem + (wm * K::CQ * 4) * 256 + lane * 4;           // U slab of this wave's channel group, k-step q0
    const float* dp0 = smem + dbase;
    float vA[16], vB[16];
    WOps LA, LB;
    wait_vm0();
    __builtin_amdgcn_s_barrier();
    if (a.nchunks > 1) stage_chunk<K>(rsU, rsW, voff, lds_base + 4u * (unsigned)K::BUF, wave, chunk_u, chunk_w);
    wino_load<K>(LA, ub0, dp0);
    wino_load<K>(LB, ub0 + 4 * 256, dp0 + 4 * K::CS);
    wino_transform(LA, vA);
    for (int c = 0; c < a.nchunks; ++c) {
      const bool more = c + 1 < a.nchunks;
      const int nb = (c + 1) & 1;
      // k-step (c, q0); the operands of (c, q1) are transformed in its shadow
      wino_mfma(acc[0], vA, LA.u);
      wino_transform(LB, vB);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        wait_vm0();
        __builtin_amdgcn_s_barrier();
        if (c + 2 < a.nchunks)
        {
          // the lane number is re-derived here (2 VALU instructions per chunk) instead of living in two VGPRs across the loop: at the
          // 168-VGPR cap of three waves per SIMD the allocator otherwise spills exactly these two, and a scratch reload in front of a
          // buffer_load ... lds waits for vmcnt(0), i.e. drains the DMA queue six times per chunk
          unsigned ln;
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
          const unsigned* tab = reinterpret_cast<const unsigned*>(smem + 2 * K::BUF) + (wave * 64 + ln);
          stage_chunk_tab<K>(rsU, rsW, tab, 16u * ln, a.kquads, lds_base + 4u * (unsigned)((c & 1) * K::BUF), wave, (unsigned)(c + 2) * chunk_u, (unsigned)(c + 2) * chunk_w);
        }
        wino_load<K>(LA, ub0 + nb * K::BUF, dp0 + nb * K::BUF);
      }
      __builtin_amdgcn_sched_barrier(0);
      // k-step (c, q1); (c + 1, q0) is transformed in its shadow, (c + 1, q1) goes into flight
      wino_mfma(acc[0], vB, LB.u);
      if (more) {
        wino_transform(LA, vA);
        __builtin_amdgcn_sched_barrier(0);
        wino_load<K>(LB, ub0 + nb * K::BUF + 4 * 256, dp0 + nb * K::BUF + 4 * K::CS);
      }
    }
  } else {
  for (int c = 0; c < a.nchunks; ++c) {
    const int buf = c & 1;
    wait_vm0();                              // this wave's part of chunk c (issued one chunk ago) has landed
    __builtin_amdgcn_s_barrier();            // ... and everybody's; everybody is also done reading the other buffer
    if (c + 1 < a.nchunks)
      stage_chunk<K>(rsU, rsW, voff, lds_base + 4u * (unsigned)((buf ^ 1) * K::BUF), wave, (unsigned)(c + 1) * chunk_u, (unsigned)(c + 1) * chunk_w);
    const float* sb = smem + buf * K::BUF;
#pragma unroll
    for (int q = 0; q < K::CQ; ++q) {
      // B operands: U[position][16 channels of the group][channel 4 q + kq] for the 16 positions, 4 per 16-byte read
      f32x4 u[MW][4];
#pragma unroll
      for (int j = 0; j < MW; ++j)
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) u[j][pq] = *reinterpret_cast<const f32x4*>(sb + (((wm * MW + j) * K::CQ + q) * 4 + pq) * 256 + lane * 4);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        // the 4x4 patch of channel 4 q + kq under tile (ty, tx) of block m, then V = B^T d B
        WOps po;
        wino_load_patch<K>(po, sb + dbase + q * 4 * K::CS + m * 2 * K::TGX);
        float v[16];
        wino_transform(po, v);
#pragma unroll
        for (int j = 0; j < MW; ++j)
#pragma unroll
          for (int p = 0; p < 16; ++p)
            acc[m * MW + j][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[p], u[j][p >> 2][p & 3], acc[m * MW + j][p], 0, 0, 0);
      }
    }
  }
  }

  // ---- epilogue: lane (tile-row block rb = lane >> 4, channel lane & 15) holds, per position, the 4 tiles rb * 4 + r:
  // block row trow, 4 consecutive tile columns tc0 .. tc0 + 3 -> Y = A^T M A per tile: 2 output rows x 8 consecutive columns
  // (the lane number is re-derived: nothing of the prologue's per-lane values has to stay in a VGPR across the k loop)
  unsigned lane_e;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
  const int rb = (int)(lane_e >> 4), lc = (int)(lane_e & 15);
  const int trow = K::TGX == 8 ? rb >> 1 : rb, tc0 = K::TGX == 8 ? 4 * (rb & 1) : 0;
  const int oy = y0 + 2 * (K::TGY * wny + trow);
#pragma unroll
  for (int j = 0; j < MW; ++j) {
  const int co = 16 * ((g * K::WM + wm) * MW + j) + lc;
  const float bv = a.bias ? a.bias[co] : 0.f;
  float* oplane = a.out + ((size_t)n * a.out_ctot + a.out_c0 + co) * a.Hout * a.Wout;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ox = x0 + 2 * (K::TGX * (MT * wnx + m) + tc0);
    const f32x4 (&am)[16] = acc[m * MW + j];
    float y[2][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t[2][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t[0][k] = am[k][r] + am[4 + k][r] + am[8 + k][r];
        t[1][k] = am[4 + k][r] - am[8 + k][r] - am[12 + k][r];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        y[h][2 * r] = t[h][0] + t[h][1] + t[h][2];
        y[h][2 * r + 1] = t[h][1] - t[h][2] - t[h][3];
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (oy + h >= a.Hout) continue;
      float* orow = oplane + (size_t)(oy + h) * a.Wout;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s = y[h][e] + bv;
        if (a.relu) s = s > 0.f ? s : s * a.slope;
        y[h][e] = s;
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int x = ox + 4 * half;
        if (x + 3 < a.Wout) *reinterpret_cast<f32x4*>(orow + x) = f32x4{y[h][4 * half], y[h][4 * half + 1], y[h][4 * half + 2], y[h][4 * half + 3]};
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (x + e < a.Wout) orow[x + e] = y[h][4 * half + e];
        }
      }
    }
  }
  }
}

// Task list, cut into 8 contiguous ranges, one per XCD (block b runs on XCD b % 8): (group block, sample, tile row, tile column,
// group within the block of kGL), innermost fastest.  The kGL workgroups that share an input window run back to back on one XCD
// (the window is fetched once into its L2), and a contiguous range stays inside one block of kGL channel groups, whose U slabs
// (kGL x Cin x 64 B, e.g. 1.9 MB for conv3_1) then live in that XCD's 4 MiB L2 instead of streaming from the Infinity Cache for
// every workgroup (with the group index fastest over all 16 groups every XCD cycled through the whole 7.8 MB).
constexpr int kGL = 4;
__device__ __forceinline__ void decode_tile(const Args& a, unsigned t, int& g, int& bx, int& by, int& n) {
  const unsigned per_block = (unsigned)a.N * a.ty * a.tx * kGL;
  const int gh = t / per_block;
  t -= gh * per_block;
  const int gl = a.ng - gh * kGL < kGL ? a.ng - gh * kGL : kGL;      // the last block may hold fewer groups
  const int glo = t % gl; t /= gl;
  g = gh * kGL + glo;
  bx = t % a.tx; t /= a.tx;
  by = t % a.ty;
  n = t / a.ty;
}

template <class K>
__global__ void __launch_bounds__(K::THREADS, (K::NW == 4 && K::MT == 1 && K::MW == 1) ? 3 : 2)      // 3 workgroups of 4 waves per CU: <= 168 VGPRs
conv_wino(Args a) {
  const unsigned per_xcd = (a.total + 7) / 8;
  const unsigned t = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (blockIdx.x / 8 >= per_xcd || t >= a.total) return;
  int g, bx, by, n;
  decode_tile(a, t, g, bx, by, n);
  wino_body<K>(a, g, bx, by, n);
}

// Split tail (see conv_mfma.hip): whole rounds of tiles run as they are, every remaining tile as two workgroups of half the
// width (MT / 2 tile blocks per wave).  Same arithmetic per output element: same bits.
template <class K>
__global__ void __launch_bounds__(K::THREADS, 2)
conv_wino_tail(Args a) {
  using KH = Cfg<K::TGY, K::TGX, K::MT / 2, K::WNY, K::WNX, K::WM, K::MW>;
  int g, bx, by, n;
  if (blockIdx.x < a.nbig) {
    const unsigned t = (blockIdx.x % 8) * (a.nbig / 8) + blockIdx.x / 8;
    decode_tile(a, t, g, bx, by, n);
    wino_body<K>(a, g, bx, by, n);
  } else {
    const unsigned b = blockIdx.x - a.nbig, nsm = 2 * (a.total - a.nbig), per_xcd = (nsm + 7) / 8;
    const unsigned u = (b % 8) * per_xcd + b / 8;
    if (b / 8 >= per_xcd || u >= nsm) return;
    decode_tile(a, a.nbig + u / 2, g, bx, by, n);
    wino_body<KH>(a, g, 2 * bx + (int)(u & 1), by, n);
  }
}

// weight [Cout][Cin][3][3] -> U = G g G^T in MFMA operand order [Cout/16][quads][position quad][lane][4]
// (lane = 16 kq + co, element e of position quad pq = position 4 pq + e = (xi, nu) = (pq, e)), zero beyond Cin.
__device__ __host__ inline void wino_u(const float g[3][3], float U[4][4]) {
  float tmp[4][3];
  for (int k = 0; k < 3; ++k) {
    tmp[0][k] = g[0][k];
    tmp[1][k] = ((g[0][k] + g[1][k]) + g[2][k]) * 0.5f;
    tmp[2][k] = ((g[0][k] - g[1][k]) + g[2][k]) * 0.5f;
    tmp[3][k] = g[2][k];
  }
  for (int i = 0; i < 4; ++i) {
    U[i][0] = tmp[i][0];
    U[i][1] = ((tmp[i][0] + tmp[i][1]) + tmp[i][2]) * 0.5f;
    U[i][2] = ((tmp[i][0] - tmp[i][1]) + tmp[i][2]) * 0.5f;
    U[i][3] = tmp[i][2];
  }
}

__global__ void pack_u(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int kquads) {
  const long long total = (long long)(Cout / 16) * kquads * 64;       // one thread per (group, quad, lane)
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long r = i >> 6;
    const int cq = (int)(r % kquads), grp = (int)(r / kquads);
    const int co = 16 * grp + (lane & 15), ci = 4 * cq + (lane >> 4);
    float g[3][3], U[4][4];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) g[a][b] = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + a * 3 + b] : 0.f;
    wino_u(g, U);
    float* dst = up + ((size_t)grp * kquads + cq) * 1024 + lane * 4;
    for (int pq = 0; pq < 4; ++pq)
      for (int e = 0; e < 4; ++e) dst[pq * 256 + e] = U[pq][e];
  }
}

inline int kquads_for(int Cin) { return cdiv(cdiv(Cin, 4), kCQ) * kCQ; }

template <class K>
static void set_geometry(Args& a) {
  a.tx = cdiv(a.Wout, K::TW); a.ty = cdiv(a.Hout, K::TH);
  a.ng = a.Cout / (16 * K::WM * K::MW);
  a.nchunks = a.kquads / K::CQ;
}
inline long long tiles_of(const Args& a) { return (long long)a.N * a.tx * a.ty * a.ng; }

template <class K>
static int launch(const Args& base, hipStream_t st) {
  Args a = base;
  set_geometry<K>(a);
  if (tiles_of(a) > 0x3fffff00ll) return fail(FN2_ERR_UNSUPPORTED, "conv_wino: grid too large");
  a.total = (unsigned)tiles_of(a); a.nbig = a.total;
  constexpr size_t lds = sizeof(float) * K::LDS_FLOATS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wino<K>), dim3(8 * ((a.total + 7) / 8)), dim3(K::THREADS), lds, st, a);
  return check_launch("conv_wino_forward");
}

template <class K>
static int launch_tail(const Args& base, hipStream_t st) {
  if constexpr (K::MT < 2) {
    return fail(FN2_ERR_UNSUPPORTED, "conv_wino: variant has no split tail");
  } else {
    Args a = base;
    set_geometry<K>(a);
    if (tiles_of(a) > 0x1fffff00ll) return fail(FN2_ERR_UNSUPPORTED, "conv_wino: grid too large");
    a.total = (unsigned)tiles_of(a); a.nbig = (unsigned)(tiles_of(a) / 256 * 256);
    const unsigned nsm = 2 * (a.total - a.nbig);
    using KH = Cfg<K::TGY, K::TGX, K::MT / 2, K::WNY, K::WNX, K::WM, K::MW>;
    constexpr size_t lds = sizeof(float) * (K::LDS_FLOATS > KH::LDS_FLOATS ? K::LDS_FLOATS : KH::LDS_FLOATS);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_tail<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL((conv_wino_tail<K>), dim3(a.nbig + 8 * ((nsm + 7) / 8)), dim3(K::THREADS), lds, st, a);
    return check_launch("conv_wino_forward");
  }
}

struct Variant {
  int tgy, tgx, mt, wny, wnx, wm, mw;
  int (*fn)(const Args&, hipStream_t);
  int (*fn_tail)(const Args&, hipStream_t);
};
template <class K, bool TAIL> struct TailFn { static constexpr int (*fn)(const Args&, hipStream_t) = nullptr; };
template <class K> struct TailFn<K, true> { static constexpr int (*fn)(const Args&, hipStream_t) = &launch_tail<K>; };

// (TGY, TGX, MT, WNY, WNX, WM): tile block shape, blocks per wave (along x), waves of the workgroup along y / x / channel groups
// (TGY, TGX, MT, WNY, WNX, WM, MW)
#define FN2_WV_LIST(X) \
  X(4, 4, 2, 4, 1, 1, 1) X(4, 4, 2, 2, 2, 1, 1) X(4, 4, 2, 1, 4, 1, 1) X(2, 8, 2, 4, 1, 1, 1) X(2, 8, 2, 2, 2, 1, 1) X(2, 8, 2, 1, 4, 1, 1) \
  X(4, 4, 1, 4, 1, 1, 1) X(4, 4, 1, 2, 2, 1, 1) X(4, 4, 1, 1, 4, 1, 1) X(2, 8, 1, 4, 1, 1, 1) X(2, 8, 1, 2, 2, 1, 1) X(2, 8, 1, 1, 4, 1, 1) \
  /* two channel groups per WAVE: one transformed patch feeds 32 MFMAs */ \
  X(4, 4, 1, 4, 1, 1, 2) X(4, 4, 1, 2, 2, 1, 2) X(4, 4, 1, 1, 4, 1, 2) X(2, 8, 1, 4, 1, 1, 2) X(2, 8, 1, 2, 2, 1, 2) X(2, 8, 1, 1, 4, 1, 2) \
  /* two channel groups per workgroup on separate waves (small maps), eight waves on pixels only */ \
  X(4, 4, 1, 1, 2, 2, 1) X(4, 4, 1, 2, 1, 2, 1) X(2, 8, 1, 1, 2, 2, 1) X(2, 8, 1, 2, 1, 2, 1) X(4, 4, 1, 1, 8, 1, 1) X(2, 8, 1, 2, 4, 1, 1)
#define FN2_WV_ROW(TGY, TGX, MT, WNY, WNX, WM, MW) \
  {TGY, TGX, MT, WNY, WNX, WM, MW, &launch<Cfg<TGY, TGX, MT, WNY, WNX, WM, MW>>, TailFn<Cfg<TGY, TGX, MT, WNY, WNX, WM, MW>, (MT >= 2)>::fn},
static const Variant kVariants[] = {FN2_WV_LIST(FN2_WV_ROW)};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
int g_forced_variant = -1;

static bool variant_applies(const Variant& v, const Args& a) { return a.Cout % (16 * v.wm * v.mw) == 0; }

// Cost model (used when the candidates cannot be timed): tile times of the busiest CU x MFMA work of a workgroup per SIMD.
static double variant_cost(const Variant& v, const Args& a, bool tail) {
  if (!variant_applies(v, a)) return 1e30;
  const int th = 2 * v.tgy * v.wny, tw = 2 * v.tgx * v.mt * v.wnx;
  const long long wgs = (long long)a.N * cdiv(a.Hout, th) * cdiv(a.Wout, tw) * (a.Cout / (16 * v.wm * v.mw));
  double rounds = (double)((wgs + 255) / 256);
  if (tail) {
    const long long r = wgs % 256;
    if (wgs < 256 || r == 0 || r > 128) return 1e30;
    rounds = (double)(wgs / 256) + 0.56;
  }
  const double per_simd = v.mt * v.mw * (v.wny * v.wnx * v.wm / 4.0);      // accumulator blocks per SIMD and workgroup
  return rounds * per_simd * (v.mt == 2 ? 1.35 : 1.0);              // measured: the 1-block-per-wave variants (112 VGPRs) run the matrix pipes fuller
}

}  // namespace wino
}  // namespace fn2

using namespace fn2;

FN2_API int fn2_conv_wino_supported(int Cin, int Hin, int Win, int Cout, int pad) {
  if (Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || Cout % 16 != 0 || Win % 4 != 0 || pad != wino::kPad) return 0;
  return (long long)Cin * Hin * Win < (1ll << 28);
}

FN2_API size_t fn2_conv_wino_packed_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cout % 16 != 0 || Cin <= 0) return 0;
  return (size_t)(Cout / 16) * wino::kquads_for(Cin) * 1024;
}

FN2_API int fn2_conv_wino_pack_weights(const float* weight, float* packed, int Cout, int Cin, void* stream) {
  if (!weight || !packed) return fail(FN2_ERR_INVALID_ARG, "conv_wino_pack_weights: null blob");
  if (Cout <= 0 || Cout % 16 != 0 || Cin <= 0) return fail(FN2_ERR_UNSUPPORTED, "conv_wino_pack_weights: needs Cout %% 16 == 0 (got %d)", Cout);
  const int kquads = wino::kquads_for(Cin);
  const long long total = (long long)(Cout / 16) * kquads * 64;
  hipLaunchKernelGGL(wino::pack_u, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, as_stream(stream), weight, packed, Cout, Cin, kquads);
  return check_launch("conv_wino_pack_weights");
}

FN2_API int fn2_debug_set_wino_variant(int v) { wino::g_forced_variant = v; return FN2_OK; }
FN2_API int fn2_conv_wino_num_variants(void) { return wino::kNumVariants; }

FN2_API int fn2_conv_wino_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                                  int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                  int Cout, int top_channels, int top_c0, int pad, int relu, float negative_slope, void* stream) {
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "conv_wino: bad batch");
  if (N == 0) return FN2_OK;
  if (!bottom || !packed_weight || !top) return fail(FN2_ERR_INVALID_ARG, "conv_wino: null blob");
  if (!fn2_conv_wino_supported(Cin, Hin, Win, Cout, pad))
    return fail(FN2_ERR_UNSUPPORTED, "conv_wino: unsupported geometry (Cin %d, %dx%d, Cout %d, pad %d)", Cin, Hin, Win, Cout, pad);
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels)
    return fail(FN2_ERR_INVALID_ARG, "conv_wino: channel slice outside the blob");
  if (((reinterpret_cast<uintptr_t>(bottom) | reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(packed_weight)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "conv_wino: blobs must be 16-byte aligned");
  wino::Args a{};
  a.in = bottom; a.up = packed_weight; a.bias = bias; a.out = top;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.in_ctot = bottom_channels; a.in_c0 = bottom_c0;
  a.Cout = Cout; a.Hout = Hin + 2 * pad - 2; a.Wout = Win + 2 * pad - 2; a.out_ctot = top_channels; a.out_c0 = top_c0;
  a.kquads = wino::kquads_for(Cin);
  a.slope = negative_slope; a.relu = relu;
  int best = -1;
  bool tail = false;
  if (wino::g_forced_variant >= 0) {
    tail = wino::g_forced_variant >= 1000;
    best = wino::g_forced_variant % 1000;
    if (best >= wino::kNumVariants || !wino::variant_applies(wino::kVariants[best], a) || (tail && !wino::kVariants[best].fn_tail))
      return fail(FN2_ERR_UNSUPPORTED, "conv_wino: forced variant %d does not apply", wino::g_forced_variant);
  } else {
    hipStream_t st = as_stream(stream);
    int picked = -1;
    if (autotune_enabled(st)) {
      static TuneCache cache("conv_wino", wino::kNumVariants);
      auto usable = [&](int c) -> bool {
        const wino::Variant& v = wino::kVariants[c / 2];
        return wino::variant_applies(v, a) && (!(c & 1) || (v.fn_tail && wino::variant_cost(v, a, true) < 1e29));
      };
      const TuneKey key{N, Cin, Hin, Win, Cout, pad, bottom_channels == Cin, top_channels == Cout, 0, 0};
      picked = autotune_pick(cache, key, 2 * wino::kNumVariants, st, [&](int c) -> int {
        const wino::Variant& v = wino::kVariants[c / 2];
        if (!wino::variant_applies(v, a)) return FN2_ERR_UNSUPPORTED;
        if (c & 1) return (v.fn_tail && wino::variant_cost(v, a, true) < 1e29) ? v.fn_tail(a, st) : FN2_ERR_UNSUPPORTED;
        return v.fn(a, st);
      }, usable);
    }
    if (picked >= 0) { best = picked / 2; tail = (picked & 1) != 0; }
    else {
      double bc = 0;
      for (int i = 0; i < wino::kNumVariants; ++i)
        for (int t = 0; t < (wino::kVariants[i].fn_tail ? 2 : 1); ++t) {
          const double c = wino::variant_cost(wino::kVariants[i], a, t == 1);
          if (best < 0 || c < bc) { best = i; bc = c; tail = t == 1; }
        }
    }
  }
  return tail ? wino::kVariants[best].fn_tail(a, as_stream(stream)) : wino::kVariants[best].fn(a, as_stream(stream));
}
