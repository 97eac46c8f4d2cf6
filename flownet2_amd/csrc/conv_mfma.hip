// Direct convolution (Convolution{kernel_size KS, stride S, pad} + bias + ReLU{negative_slope}) of the FlowNet encoders on
// v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation in k order (bit-identical to an fmaf chain).
//
// Reference: ConvolutionLayer::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23: per SAMPLE im2col_gpu + cublasSgemm, then
// forward_gpu_bias -- base_conv_layer.cpp:326-348) followed by the in-place ReLU layer (relu_layer.cu:8-27).  Here: one
// launch per layer for the whole mini-batch, no column matrix, NCHW in and out (no layout transposes), bias and activation in
// the epilogue, optional channel slices on both blobs (a consumer's concat blob can be written in place).
//
// GEMM view.  Output pixels are cut into 4x4 PATCHES (one MFMA M tile: 16 pixels), output channels into groups of 16 (one N
// tile); K = (channel quad cq, ky, kx) k-steps, the 4 k values of a k-step being the 4 channels of the quad -- so that the
// pixel operand of tap (ky, kx) for lane (pixel p, kq) is ONE ds_read_b32 at  lane_base + immediate  from a window of the
// input staged in its natural [channel][row][column] order:
//     address = [kq * CS + (py * S) * RS + px * S]  +  [cq * 4 * CS + ky * RS + kx + 4 * S * patch]
//   * pixel operand: the input window of a workgroup tile, CQ channel quads at a time, arrives by 16-byte LDS-DMA
//     (buffer_load_dwordx4 ... lds) straight from NCHW; rows / columns outside the image and channels beyond Cin are
//     out-of-range for the buffer descriptor and come back 0.0f = the zero padding.  Two window buffers, one barrier per chunk.
//     RS == 4 (mod 16) and CS == 16 (mod 32) make the stride-1 reads conflict-free (stride 2: 2-way, the odd banks idle).
//   * weight operand: pre-packed once per weight blob (fn2_conv_mfma_pack_weights) as [Cout/64][k-step][lane][4]: lane
//     (co, kq) holds W[64 g + 16 j + co][4 cq + kq][ky][kx] for j = 0..3, so a wave's operand for a k-step is ONE coalesced
//     global_load_dwordx{MW} per lane straight into registers, prefetched NBUFA - 1 k-steps ahead; no LDS for weights.
//   * wave tile = MW channel groups x NP patches (MW * NP accumulator tiles), workgroup = WM x WNX x WNY waves.
//   * epilogue: the MFMA result layout hands every lane 4 consecutive x of one output channel: bias + leaky ReLU + 16-byte store.
#include "fn2_common.hpp"
#include "autotune.hpp"

#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace fn2 {
namespace cv {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int up_mod(int v, int r, int m) { return v + ((r - v % m) + m) % m; }   // smallest >= v with == r (mod m)

struct Args {
  const float* in; const float* wp; const float* bias; float* out;
  int N, Cin, Hin, Win, in_ctot, in_c0;
  int Cout, Hout, Wout, out_ctot, out_c0;
  int pad;
  int nchunks;        // chunks of CQ channel quads
  int ksteps;         // k-steps of the packed weights per 64-channel group (= quads * KS * KS), multiple of the chunk length
  int tx, ty;         // workgroup tiles per sample along x / y
  int ng;             // Cout / (16 * MW * WM)
  unsigned total;     // tiles (= workgroups of the plain launch)
  unsigned nbig;      // split-tail launch: tiles that run whole (a multiple of 256); the rest run as two half-channel workgroups
  float slope; int relu;
};

// P16 (1x1 kernels only): the M tile is 16 CONSECUTIVE pixels of one row instead of a 4x4 patch -- the launcher flattens every plane into
// one row of H * W pixels (a 1x1 convolution has no spatial window), so no tile hangs over an image edge: the plain-GEMM form.
template <int KS_, int S_, int MW_, int NP_, int WM_, int WNX_, int WNY_, int CQ_, int P16_ = 0, int PIN_ = 0>
struct Cfg {
  static constexpr int KS = KS_, S = S_, MW = MW_, NP = NP_, WM = WM_, WNX = WNX_, WNY = WNY_, CQ = CQ_, P16 = P16_;
  static constexpr int PIN = PIN_;      // 1: operand reads pinned one behind each patch's MFMAs, a k-step ahead (else the compiler's order)
  static constexpr int NW = WM * WNX * WNY, THREADS = 64 * NW;
  static constexpr int PADL = 4;                                     // window columns left of S * x0 (16-byte aligned start)
  static constexpr int TW = (P16 ? 16 : 4) * NP * WNX, TH = (P16 ? 1 : 4) * WNY;   // output pixels of a workgroup tile
  static_assert(!P16 || (KS == 1 && S == 1), "row tiles are for 1x1 kernels");
  static constexpr int WR = (TH - 1) * S + KS;                       // window rows per channel
  static constexpr int WC = (TW - 1) * S + KS + PADL;                // window columns incl. the left margin (pad <= PADL)
  static constexpr int RS = up_mod(cdiv(WC, 4) * 4, 4, 16);          // row stride (dwords)
  static constexpr int CS = up_mod(WR * RS, 16, 32);                 // channel stride
  static constexpr int SLOTS_C = CS / 4;                             // 16-byte slots per channel
  static constexpr int SLOTS = 4 * CQ * SLOTS_C;                     // per chunk
  static constexpr int NRUN = cdiv(SLOTS, 64);                       // 1 KiB LDS-DMA runs per chunk
  static constexpr int RPW = cdiv(NRUN, NW);                         // runs per wave
  static constexpr int BUF = NRUN * 256;                             // dwords per window buffer (whole runs)
  static constexpr int KSC = CQ * KS * KS;                           // k-steps per chunk
  static constexpr int NBUFA = (KSC % 6 == 0) ? 6 : (KSC % 5 == 0) ? 5 : (KSC % 7 == 0) ? 7 : (KSC % 4 == 0) ? 4 : 3;   // weight-operand ring
  static_assert(KSC % NBUFA == 0, "ring phase must repeat per chunk");
  static_assert(2 * BUF * 4 <= 160 * 1024, "LDS");
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

// LDS-DMA of one chunk's window: run r = i * NW + wave -> 1 KiB at dst + 1024 r (a __device__ function, not a lambda: the
// host pass of a __global__ template cannot see the amdgcn builtins inside a lambda body)
template <class K>
__device__ __forceinline__ void stage_chunk(__amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[K::RPW], unsigned dst, int wave, unsigned soff) {
#pragma unroll
  for (int i = 0; i < K::RPW; ++i) {
    const int r = i * K::NW + wave;
    if (r < K::NRUN)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)r), 16, voff[i], soff, 0, 0);
  }
}

// One workgroup tile: channel group g (of 16 * MW * WM channels), pixel tile (bx, by) of sample n.
template <class K>
__device__ __forceinline__ void conv_body(const Args& a, int g, int bx, int by, int n) {
  static_assert(K::THREADS == 256, "launch bounds");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KS = K::KS, S = K::S, MW = K::MW, NP = K::NP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % K::WM, wnx = (wave / K::WM) % K::WNX, wny = wave / (K::WM * K::WNX);
  const int x0 = bx * K::TW, y0 = by * K::TH;

  // ---- LDS-DMA plan: run r = i * NW + wave, slot s = 64 r + lane -> (channel, window row, group of 4 columns)
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
      const int yi = S * y0 - a.pad + row, xi = S * x0 - K::PADL + 4 * gq;
      if (row < K::WR && 4 * gq < K::WC && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win)
        voff[i] = 4u * (unsigned)(c * plane + (size_t)yi * a.Win + xi);
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const unsigned chunk_bytes = 4u * 4u * K::CQ * (unsigned)plane;
  auto stage = [&](int chunk, int buf) { stage_chunk<K>(rs, voff, lds_base + 4u * (unsigned)(buf * K::BUF), wave, (unsigned)chunk * chunk_bytes); };

  // ---- operands
  const int kq = lane >> 4, p16 = lane & 15, py = K::P16 ? 0 : p16 >> 2, px = K::P16 ? p16 : p16 & 3;
  constexpr int PW = K::P16 ? 16 : 4, PH = K::P16 ? 1 : 4;          // pixels of an M tile along x / y
  const int bbase = kq * K::CS + (S * (PH * wny + py)) * K::RS + S * (PW * NP * wnx + px) + K::PADL - a.pad;
  using WV = typename WVec<MW>::T;
  // packed weights: [Cout/64][ksteps][64 lanes][4]; this wave's channel groups: 16 * MW * (g * WM + wm) ...
  const int cg0 = (g * K::WM + wm) * MW;                      // first 16-channel group of this wave
  const float* wl = a.wp + ((size_t)(cg0 / 4) * a.ksteps * 64 + lane) * 4 + (cg0 % 4);
  auto wload = [&](int ks) -> WV { return *reinterpret_cast<const WV*>(wl + (size_t)ks * 256); };

  f32x4 acc[MW][NP];
#pragma unroll
  for (int j = 0; j < MW; ++j)
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[j][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  WV wreg[K::NBUFA];
  stage(0, 0);
#pragma unroll
  for (int i = 0; i < K::NBUFA - 1; ++i) wreg[i] = wload(i);

  for (int c = 0; c < a.nchunks; ++c) {
    const int buf = c & 1;
    // in flight: the window of chunk c (issued a whole chunk ago) and, younger than it, the weight prefetch of the last NBUFA - 1
    // k-steps; loads retire in order, so the window has landed once at most NBUFA - 1 loads remain -- no need to drain the ring
    wait_vmcnt<K::NBUFA - 1>();
    __builtin_amdgcn_s_barrier();
    if (c + 1 < a.nchunks) stage(c + 1, buf ^ 1);
    const float* win = smem + buf * K::BUF + bbase;
    const int ks0 = c * K::KSC;
    if constexpr (K::PIN) {
      // operand reads one k-step ahead, ONE read behind the MW MFMAs of every patch, order pinned (a block of NP reads in front of a
      // k-step's MFMAs leaves the matrix pipe a single queued instruction deep while it issues: csrc/conv_plane.hip, DESIGN 3.4 c).
      // A variant of its own: conv3 [16,128,80,112] gains (434 -> 412 us), conv2 [16,64,160,224] loses (428 -> 490): the autotuner
      // decides per layer, the bits are the same.
      float b[2][NP];
      auto tap_off = [](int ks) { const int cq = ks / (KS * KS), ky = (ks / KS) % KS, kx = ks % KS; return cq * 4 * K::CS + ky * K::RS + kx; };
#pragma unroll
      for (int p = 0; p < NP; ++p) b[0][p] = win[tap_off(0) + PW * S * p];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < K::KSC; ++ks) {
        wreg[(ks + K::NBUFA - 1) % K::NBUFA] = wload(ks0 + ks + K::NBUFA - 1);
        const WV w = wreg[ks % K::NBUFA];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
          for (int j = 0; j < MW; ++j) acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ks & 1][p], wget<MW>(w, j), acc[j][p], 0, 0, 0);
          if (ks + 1 < K::KSC) b[(ks + 1) & 1][p] = win[tap_off(ks + 1 < K::KSC ? ks + 1 : ks) + PW * S * p];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < K::KSC; ++ks) {
      wreg[(ks + K::NBUFA - 1) % K::NBUFA] = wload(ks0 + ks + K::NBUFA - 1);      // the packed array carries NBUFA spare k-steps
      const int cq = ks / (KS * KS), ky = (ks / KS) % KS, kx = ks % KS;
      float b[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) b[p] = win[cq * 4 * K::CS + ky * K::RS + kx + PW * S * p];
      const WV w = wreg[ks % K::NBUFA];
#pragma unroll
      for (int j = 0; j < MW; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[p], wget<MW>(w, j), acc[j][p], 0, 0, 0);
    }
    }
  }

  // ---- epilogue: lane (row block = lane >> 4, channel = lane & 15) holds 4 consecutive x of output row y0 + 4 wny + (lane >> 4)
  // (row tiles: pixels 4 (lane >> 4) .. + 3 of the 16-pixel tile, row y0 + wny)
  const int y = K::P16 ? y0 + wny : y0 + 4 * wny + (lane >> 4);
  if (y < a.Hout) {
#pragma unroll
    for (int j = 0; j < MW; ++j) {
      const int co = 16 * (cg0 + j) + (lane & 15);
      const float bv = a.bias ? a.bias[co] : 0.f;
      float* orow = a.out + (((size_t)n * a.out_ctot + a.out_c0 + co) * a.Hout + y) * a.Wout;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int x = K::P16 ? x0 + 16 * (NP * wnx + p) + 4 * (lane >> 4) : x0 + 4 * (NP * wnx + p);
        f32x4 v = acc[j][p];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = v[r] + bv;
          if (a.relu) s = s > 0.f ? s : s * a.slope;
          v[r] = s;
        }
        if (x + 3 < a.Wout) *reinterpret_cast<f32x4*>(orow + x) = v;
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (x + r < a.Wout) orow[x + r] = v[r];
        }
      }
    }
  }
}

// Task list: (sample, tile row, tile column, channel group), channel group fastest, cut into 8 contiguous ranges, one per XCD
// (block b runs on XCD b % 8): the workgroups that share an input window run next to each other on one XCD, whose L2 serves
// the re-reads.
__device__ __forceinline__ void decode_tile(const Args& a, unsigned t, int& g, int& bx, int& by, int& n) {
  g = t % a.ng; t /= a.ng;
  bx = t % a.tx; t /= a.tx;
  by = t % a.ty;
  n = t / a.ty;
}

template <class K>
__global__ void __launch_bounds__(256, 2)
conv_mfma(Args a) {
  const unsigned per_xcd = (a.total + 7) / 8;
  const unsigned t = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (blockIdx.x / 8 >= per_xcd || t >= a.total) return;
  int g, bx, by, n;
  decode_tile(a, t, g, bx, by, n);
  conv_body<K>(a, g, bx, by, n);
}

// The same launch with a split tail.  With W tiles over 256 CUs the last, partial round (W mod 256 tiles) leaves most CUs idle for
// a whole tile time; here the first a.nbig tiles (whole rounds) run as they are and each remaining tile is cut into two
// workgroups of half the channels (MW / 2 channel groups per wave): if they are at most 256 the tail costs half a tile time.
// The arithmetic per output element is the same k-ordered chain: same bits.
template <class K>
__global__ void __launch_bounds__(256, 2)
conv_mfma_tail(Args a) {
  using KH = Cfg<K::KS, K::S, K::MW / 2, K::NP, K::WM, K::WNX, K::WNY, K::CQ, K::P16>;
  int g, bx, by, n;
  if (blockIdx.x < a.nbig) {
    const unsigned t = (blockIdx.x % 8) * (a.nbig / 8) + blockIdx.x / 8;
    decode_tile(a, t, g, bx, by, n);
    conv_body<K>(a, g, bx, by, n);
  } else {
    const unsigned b = blockIdx.x - a.nbig, nsm = 2 * (a.total - a.nbig), per_xcd = (nsm + 7) / 8;
    const unsigned u = (b % 8) * per_xcd + b / 8;
    if (b / 8 >= per_xcd || u >= nsm) return;
    decode_tile(a, a.nbig + u / 2, g, bx, by, n);
    conv_body<KH>(a, 2 * g + (int)(u & 1), bx, by, n);
  }
}

// weight [Cout][Cin][KS][KS] -> packed [Cout/64][ksteps + spare][64][4]
__global__ void pack_weights(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KS, int ksteps, int kalloc) {
  const long long total = (long long)((Cout + 63) / 64) * kalloc * 256;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long long r = i >> 8;
    const int ks = (int)(r % kalloc), grp = (int)(r / kalloc);
    const int co = 64 * grp + 16 * j + (lane & 15), kq = lane >> 4;
    const int cq = ks / (KS * KS), tap = ks % (KS * KS);
    const int ci = 4 * cq + kq;
    float v = 0.f;
    if (ks < ksteps && ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * KS * KS + tap];
    wp[i] = v;
  }
}

// The same packing from a strided VIEW of a weight blob, through LDS: logical element (co, ci, tap) of a [Cout][Cin][KS][KS] operand is
//   src[co * s_co + ci * s_ci + (flip ? T - 1 - tap : tap)]   for co < src_co and ci < src_ci,   0 beyond (channel padding).
// That covers, without a layout pass in front: the blob as it is (s_co = Cin T, s_ci = T), its channel axes swapped (a Deconvolution's
// blob read as a Convolution's, the transposed convolutions of the data gradients: s_co = T, s_ci = src_co T), the 180-degree rotation of
// a 3x3 data gradient (flip), zero channels up to the kernels' group sizes, and the [Cout 16][Cin] GEMM operand of a Deconvolution
// (T = 1, s_co = 1, s_ci = Cout 16).  A training step repacks every weight after every update (39 M floats, two forms each): the
// gather form above reads 4 bytes per 64-byte line (17-56 us per blob, 0.53 ms per step plus 0.35 ms of torch transposes / cats / fills
// in front of it); here a workgroup reads runs of the source's fastest axis, turns them in LDS and writes whole 1 KiB k-steps.
// (KS and with it every divisor of the index decodes is a template parameter: with run-time divisors the kernel spent its time in
// 32-bit divisions -- 21 us per blob, slower than the gather it replaced.)
template <int KS> struct PackTile {
  static constexpr int T = KS * KS;
  static constexpr int QT = KS == 1 ? 8 : KS == 3 ? 3 : KS == 4 ? 2 : 1;      // channel quads per workgroup tile (<= 8192 floats; 7x7: 12544).  Twice the tile
                                                                                // (runs of 0.9-1.2 KB, 2 workgroups per CU) was 1.7x SLOWER: the kernel lives on workgroups in flight
  static constexpr int CIL = 4 * QT, ROW = CIL * T + 1;                          // + 1 float per output-channel row against bank conflicts
  static constexpr int LDS_FLOATS = 64 * ROW;
};

template <int KS, int ORDER>      // ORDER 0: source runs along (ci, tap) for a fixed co; 1: along (co, tap) for a fixed ci; 2: any strides
__global__ void __launch_bounds__(256) pack_weights_view(const float* __restrict__ w, float* __restrict__ wp, int ksteps, int kalloc,
                                                         int src_co, int src_ci, long long s_co, long long s_ci, int flip) {
  using P = PackTile<KS>;
  constexpr int T = P::T, QT = P::QT, CIL = P::CIL, ROW = P::ROW;
  __shared__ float tile[P::LDS_FLOATS];                   // [64][4 * QT][T]
  const int grp = blockIdx.x, cq0 = blockIdx.y * QT;
  const int co0 = 64 * grp, ci0 = 4 * cq0;
  const int nquads = ksteps / T;                          // channel quads of the packed operand (incl. zero padding quads)
  constexpr int TOTAL = 64 * CIL * T;
  if constexpr (ORDER == 0) {
    const float* src = w + (size_t)co0 * s_co + (size_t)ci0 * T;
#pragma unroll
    for (int i = threadIdx.x; i < TOTAL; i += 256) {      // (all loads of the thread in flight at once: TOTAL / 256 <= 49)
      const int col = i / (CIL * T), e = i - col * (CIL * T), c = e / T;
      tile[col * ROW + e] = (co0 + col < src_co && ci0 + c < src_ci) ? src[(size_t)col * s_co + e] : 0.f;
    }
  } else if constexpr (ORDER == 1) {
    const float* src = w + (size_t)ci0 * s_ci + (size_t)co0 * T;
#pragma unroll
    for (int i = threadIdx.x; i < TOTAL; i += 256) {      // (all loads of the thread in flight at once: TOTAL / 256 <= 49)
      const int c = i / (64 * T), e = i - c * (64 * T), col = e / T, t = e - col * T;
      tile[col * ROW + c * T + t] = (co0 + col < src_co && ci0 + c < src_ci) ? src[(size_t)c * s_ci + e] : 0.f;
    }
  } else {
#pragma unroll
    for (int i = threadIdx.x; i < TOTAL; i += 256) {      // (all loads of the thread in flight at once: TOTAL / 256 <= 49)
      const int col = i / (CIL * T), e = i - col * (CIL * T), c = e / T, t = e - c * T;
      tile[col * ROW + e] = (co0 + col < src_co && ci0 + c < src_ci) ? w[(size_t)(co0 + col) * s_co + (size_t)(ci0 + c) * s_ci + t] : 0.f;
    }
  }
  __syncthreads();
  // k-step ks = cq * T + tap: 256 floats [lane = kq * 16 + co % 16][j = co / 16]
  const int nq = min(QT, nquads - cq0);
  float* dst = wp + ((size_t)grp * kalloc + (size_t)cq0 * T) * 256;
  const int j = threadIdx.x & 3, lane = threadIdx.x >> 2;
  const float* trow = tile + (16 * j + (lane & 15)) * ROW + (lane >> 4) * T;
#pragma unroll
  for (int ks = 0; ks < QT * T; ++ks) {
    const int cq = ks / T, tap = ks - cq * T;
    if (cq < nq) dst[ks * 256 + threadIdx.x] = trow[4 * cq * T + (flip ? T - 1 - tap : tap)];
  }
  if (cq0 + QT >= nquads)                                 // the spare k-steps behind the group
    for (int i = threadIdx.x; i < (kalloc - ksteps) * 256; i += 256) wp[((size_t)grp * kalloc + ksteps) * 256 + i] = 0.f;
}

template <int KS>
static void launch_pack_view(const float* w, float* wp, int Cout, int ksteps, int kalloc, int src_co, int src_ci, long long s_co, long long s_ci, int flip,
                             hipStream_t st) {
  using P = PackTile<KS>;
  const dim3 grid((unsigned)((Cout + 63) / 64), (unsigned)cdiv(ksteps / P::T, P::QT));
  if (s_ci == (long long)P::T) hipLaunchKernelGGL((pack_weights_view<KS, 0>), grid, dim3(256), 0, st, w, wp, ksteps, kalloc, src_co, src_ci, s_co, s_ci, flip);
  else if (s_co == (long long)P::T) hipLaunchKernelGGL((pack_weights_view<KS, 1>), grid, dim3(256), 0, st, w, wp, ksteps, kalloc, src_co, src_ci, s_co, s_ci, flip);
  else hipLaunchKernelGGL((pack_weights_view<KS, 2>), grid, dim3(256), 0, st, w, wp, ksteps, kalloc, src_co, src_ci, s_co, s_ci, flip);
}

constexpr int kSpare = 8;          // spare (zero) k-steps behind every group: the weight prefetch runs NBUFA - 1 k-steps ahead
constexpr int kChunkQuads = 2;     // k-steps are padded to a whole number of chunks of at most this many channel quads (8 for 1x1 kernels)

inline int chunk_quads(int KS) { return KS == 1 ? 8 : kChunkQuads; }
inline int ksteps_for(int Cin, int KS) { return cdiv(cdiv(Cin, 4), chunk_quads(KS)) * chunk_quads(KS) * KS * KS; }

template <class K>
static void set_geometry(Args& a) {
  if constexpr (K::P16) {                 // every plane as ONE row of H * W pixels (contiguous in NCHW; a 1x1 kernel has no spatial window)
    a.Win *= a.Hin; a.Hin = 1; a.Wout *= a.Hout; a.Hout = 1;
  }
  a.tx = cdiv(a.Wout, K::TW); a.ty = cdiv(a.Hout, K::TH);
  a.ng = a.Cout / (16 * K::MW * K::WM);
  a.nchunks = cdiv(cdiv(a.Cin, 4), K::CQ);
}

inline long long tiles_of(const Args& a) { return (long long)a.N * a.tx * a.ty * a.ng; }

template <class K>
static int launch(const Args& base, hipStream_t st) {
  Args a = base;
  set_geometry<K>(a);
  if (tiles_of(a) > 0x3fffff00ll) return fail(FN2_ERR_UNSUPPORTED, "conv_mfma: grid too large");
  a.total = (unsigned)tiles_of(a); a.nbig = a.total;
  const unsigned grid = 8 * ((a.total + 7) / 8);
  constexpr size_t lds = sizeof(float) * 2 * K::BUF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_mfma<K>), dim3(grid), dim3(K::THREADS), lds, st, a);
  return check_launch("conv_mfma_forward");
}

// whole rounds of tiles, then the remainder as half-channel workgroups (conv_mfma_tail)
inline unsigned whole_rounds(long long tiles) { return (unsigned)(tiles / 256 * 256); }

template <class K>
static int launch_tail(const Args& base, hipStream_t st) {
  if constexpr (K::MW < 2) {
    return fail(FN2_ERR_UNSUPPORTED, "conv_mfma: variant has no split tail");
  } else {
    Args a = base;
    set_geometry<K>(a);
    if (tiles_of(a) > 0x1fffff00ll) return fail(FN2_ERR_UNSUPPORTED, "conv_mfma: grid too large");
    a.total = (unsigned)tiles_of(a); a.nbig = whole_rounds(a.total);
    const unsigned nsm = 2 * (a.total - a.nbig);
    const unsigned grid = a.nbig + 8 * ((nsm + 7) / 8);
    constexpr size_t lds = sizeof(float) * 2 * K::BUF;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_tail<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL((conv_mfma_tail<K>), dim3(grid), dim3(K::THREADS), lds, st, a);
    return check_launch("conv_mfma_forward");
  }
}

struct Variant {
  int ks, s, mw, np, wm, wnx, wny, p16;
  int (*fn)(const Args&, hipStream_t);
  int (*fn_tail)(const Args&, hipStream_t);     // nullptr: no split-tail form
};

// X-macro list of the tile variants: (KS, S, MW, NP, WM, WNX, WNY, CQ, split-tail form instantiated)
#define FN2_CV_LIST(X) \
  /* 1x1 stride 1: a plain GEMM over the channels (conv_redir; the weight^T x bottom product of the Deconvolution layers), 8 quads per chunk */ \
  X(1, 1, 2, 7, 2, 2, 1, 8, 1) X(1, 1, 2, 7, 2, 1, 2, 8, 1) X(1, 1, 4, 7, 1, 2, 2, 8, 0) X(1, 1, 4, 7, 1, 1, 4, 8, 0) X(1, 1, 2, 6, 2, 2, 1, 8, 1) \
  X(1, 1, 2, 4, 2, 2, 1, 8, 1) X(1, 1, 2, 7, 1, 2, 2, 8, 0) X(1, 1, 2, 7, 1, 4, 1, 8, 0) X(1, 1, 2, 4, 1, 2, 2, 8, 0) X(1, 1, 4, 4, 1, 2, 2, 8, 0) \
  /* 3x3 stride 1 */ \
  X(3, 1, 2, 7, 2, 2, 1, 2, 1) X(3, 1, 2, 7, 2, 1, 2, 2, 1) X(3, 1, 4, 7, 1, 2, 2, 2, 0) X(3, 1, 4, 7, 1, 1, 4, 2, 0) \
  X(3, 1, 2, 6, 2, 2, 1, 2, 1) X(3, 1, 4, 6, 1, 2, 2, 2, 0) X(3, 1, 4, 4, 1, 2, 2, 2, 0) X(3, 1, 2, 4, 2, 2, 1, 2, 1) \
  /* 3x3 stride 2 */ \
  X(3, 2, 2, 7, 2, 1, 2, 2, 1) X(3, 2, 4, 7, 1, 1, 4, 2, 0) X(3, 2, 2, 6, 2, 2, 1, 2, 1) X(3, 2, 4, 6, 1, 2, 2, 2, 0) X(3, 2, 2, 4, 2, 2, 1, 2, 1) \
  /* 4x4 stride 2 (the data gradient of the Deconvolution{4, 2, 1} layers: a convolution of top_diff with the weight blob as it is) */ \
  X(4, 2, 2, 7, 2, 2, 1, 2, 1) X(4, 2, 2, 7, 2, 1, 2, 2, 1) X(4, 2, 4, 7, 1, 2, 2, 2, 0) X(4, 2, 2, 6, 2, 2, 1, 2, 1) X(4, 2, 4, 4, 1, 2, 2, 2, 0) X(4, 2, 2, 4, 2, 2, 1, 2, 1) \
  /* 5x5 stride 2 */ \
  X(5, 2, 2, 7, 2, 2, 1, 1, 1) X(5, 2, 2, 7, 2, 1, 2, 1, 1) X(5, 2, 4, 7, 1, 2, 2, 1, 0) X(5, 2, 4, 7, 1, 1, 4, 1, 0) \
  X(5, 2, 2, 6, 2, 2, 1, 1, 1) X(5, 2, 4, 6, 1, 2, 2, 1, 0) X(5, 2, 4, 4, 1, 2, 2, 1, 0) \
  /* 7x7 stride 2 (the 12-channel stems of FlowNet2's stacked nets: whole channel quads) */ \
  X(7, 2, 2, 6, 2, 2, 1, 1, 1) X(7, 2, 2, 7, 2, 2, 1, 1, 1) X(7, 2, 4, 4, 1, 2, 2, 1, 0) X(7, 2, 4, 6, 1, 2, 2, 1, 0) X(7, 2, 2, 4, 2, 2, 1, 1, 1)

template <class K, int TAIL> struct TailFn { static constexpr int (*fn)(const Args&, hipStream_t) = nullptr; };
template <class K> struct TailFn<K, 1> { static constexpr int (*fn)(const Args&, hipStream_t) = &launch_tail<K>; };

#define FN2_CV_ROW(KS, S, MW, NP, WM, WNX, WNY, CQ, TAIL) \
  {KS, S, MW, NP, WM, WNX, WNY, 0, &launch<Cfg<KS, S, MW, NP, WM, WNX, WNY, CQ>>, TailFn<Cfg<KS, S, MW, NP, WM, WNX, WNY, CQ>, TAIL>::fn},
#define FN2_CV_ROW16(MW, NP, WM, WNX, WNY, TAIL) \
  {1, 1, MW, NP, WM, WNX, WNY, 1, &launch<Cfg<1, 1, MW, NP, WM, WNX, WNY, 8, 1>>, TailFn<Cfg<1, 1, MW, NP, WM, WNX, WNY, 8, 1>, TAIL>::fn},
#define FN2_CV_PIN(KS, S, MW, NP, WM, WNX, WNY, CQ, P16, TAIL) \
  {KS, S, MW, NP, WM, WNX, WNY, P16, &launch<Cfg<KS, S, MW, NP, WM, WNX, WNY, CQ, P16, 1>>, TailFn<Cfg<KS, S, MW, NP, WM, WNX, WNY, CQ, P16, 1>, TAIL>::fn},
static const Variant kVariants[] = {FN2_CV_LIST(FN2_CV_ROW)
  /* 1x1 on flattened planes (16-pixel row tiles): TW = 16 NP WNX pixels */
  /* (WNY = 1: a flattened plane has one row) */
  FN2_CV_ROW16(2, 7, 2, 2, 1, 1) FN2_CV_ROW16(2, 7, 4, 1, 1, 1) FN2_CV_ROW16(4, 7, 1, 4, 1, 0) FN2_CV_ROW16(4, 7, 2, 2, 1, 0) FN2_CV_ROW16(2, 5, 2, 2, 1, 1)
  FN2_CV_ROW16(2, 4, 2, 2, 1, 1) FN2_CV_ROW16(2, 7, 1, 4, 1, 0) FN2_CV_ROW16(2, 5, 1, 4, 1, 0) FN2_CV_ROW16(4, 5, 1, 4, 1, 0) FN2_CV_ROW16(2, 4, 1, 4, 1, 0)
  FN2_CV_ROW16(4, 5, 2, 2, 1, 0) FN2_CV_ROW16(4, 4, 2, 2, 1, 0)
  /* planes of 140 (10x14) / 288 (12x24) pixels: 9 x 16 = 144 */
  FN2_CV_ROW16(2, 9, 4, 1, 1, 1) FN2_CV_ROW16(2, 9, 2, 2, 1, 1) FN2_CV_ROW16(2, 3, 2, 2, 1, 1) FN2_CV_ROW16(4, 3, 1, 4, 1, 0)
  /* 256-channel workgroup tiles (36 / 28 accumulator tiles per wave): fewer re-reads of the pixel operand by the big-M deconvolution GEMMs */
  FN2_CV_ROW16(4, 9, 4, 1, 1, 0) FN2_CV_ROW16(4, 9, 2, 2, 1, 0) FN2_CV_ROW16(4, 7, 4, 1, 1, 0)
  /* pinned issue order (Cfg::PIN): the 5x5 / 2 tiles the encoders pick and the GEMM tiles of the deconvolutions */
  FN2_CV_PIN(5, 2, 2, 7, 2, 2, 1, 1, 0, 1) FN2_CV_PIN(5, 2, 2, 7, 2, 1, 2, 1, 0, 1)
  FN2_CV_PIN(1, 1, 4, 9, 4, 1, 1, 8, 1, 0) FN2_CV_PIN(1, 1, 2, 9, 4, 1, 1, 8, 1, 1) FN2_CV_PIN(1, 1, 2, 5, 2, 2, 1, 8, 1, 1) FN2_CV_PIN(1, 1, 4, 7, 4, 1, 1, 8, 1, 0)
  FN2_CV_PIN(1, 1, 2, 7, 2, 2, 1, 8, 1, 1) FN2_CV_PIN(1, 1, 2, 3, 2, 2, 1, 8, 1, 1)
  FN2_CV_PIN(3, 1, 2, 7, 2, 2, 1, 2, 0, 1) FN2_CV_PIN(3, 1, 2, 6, 2, 2, 1, 2, 0, 1) FN2_CV_PIN(3, 2, 2, 7, 2, 1, 2, 2, 0, 1) FN2_CV_PIN(3, 2, 2, 6, 2, 2, 1, 2, 0, 1)
  FN2_CV_PIN(4, 2, 2, 7, 2, 2, 1, 2, 0, 1) FN2_CV_PIN(4, 2, 2, 6, 2, 2, 1, 2, 0, 1) FN2_CV_PIN(5, 2, 2, 6, 2, 2, 1, 1, 0, 1)
  FN2_CV_PIN(7, 2, 2, 6, 2, 2, 1, 1, 0, 1) FN2_CV_PIN(7, 2, 4, 6, 1, 2, 2, 1, 0, 0)};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int g_forced_variant = -1;       // >= 0: plain launch of that variant; >= 1000: its split-tail launch

// Cost model: tile times of the busiest CU x accumulator tiles of a wave (tiles hanging over the image edge included), with a
// mild penalty for small wave tiles (more operand traffic per MFMA).  tail: the last partial round as half tiles.
static double variant_cost(const Variant& v, const Args& a, bool tail) {
  const int tw = (v.p16 ? 16 : 4) * v.np * v.wnx, th = (v.p16 ? 1 : 4) * v.wny;
  const long long wgs = v.p16 ? (long long)a.N * cdiv(a.Wout * a.Hout, tw) * (a.Cout / (16 * v.mw * v.wm))
                              : (long long)a.N * cdiv(a.Wout, tw) * cdiv(a.Hout, th) * (a.Cout / (16 * v.mw * v.wm));
  double rounds = (double)((wgs + 255) / 256);
  if (tail) {
    const long long r = wgs % 256;
    if (wgs < 256 || r == 0 || r > 128) return 1e30;               // nothing to gain
    rounds = (double)(wgs / 256) + 0.5 * 1.12;                      // half tiles run MW / 2: less efficient
  }
  const double eff = 1.0 / (1.0 + 0.08 * (4.0 / v.mw - 1.0) + 0.02 * (7.0 / v.np - 1.0));
  return rounds * v.mw * v.np / eff;
}

static bool variant_applies(const Variant& v, const Args& a, int KS, int S) {
  if (!v.p16 && a.Win % 4 != 0) return false;          // 4x4-patch tiles stage 16-byte pieces of image rows; row tiles only need H * W % 4 == 0
  return v.ks == KS && v.s == S && a.Cout % (16 * v.mw * v.wm) == 0;
}

}  // namespace cv
}  // namespace fn2

using namespace fn2;

FN2_API size_t fn2_conv_mfma_packed_floats(int Cout, int Cin, int kernel) {
  if (Cout <= 0 || Cout % 32 != 0 || Cin <= 0 || kernel <= 0) return 0;
  return (size_t)((Cout + 63) / 64) * (cv::ksteps_for(Cin, kernel) + cv::kSpare) * 256;
}

FN2_API int fn2_conv_mfma_pack_weights_view(const float* weight, float* packed, int Cout, int Cin, int kernel, int src_cout, int src_cin,
                                            long long stride_cout, long long stride_cin, int flip, void* stream);

FN2_API int fn2_conv_mfma_pack_weights(const float* weight, float* packed, int Cout, int Cin, int kernel, void* stream) {
  if (!weight || !packed) return fail(FN2_ERR_INVALID_ARG, "conv_mfma_pack_weights: null blob");
  if (Cout <= 0 || Cout % 32 != 0 || Cin <= 0 || (kernel != 1 && kernel != 3 && kernel != 4 && kernel != 5 && kernel != 7))
    return fail(FN2_ERR_UNSUPPORTED, "conv_mfma_pack_weights: needs Cout %% 32 == 0 and kernel_size 1, 3, 4, 5 or 7 (got Cout %d, kernel %d)", Cout, kernel);
  return fn2_conv_mfma_pack_weights_view(weight, packed, Cout, Cin, kernel, Cout, Cin, (long long)Cin * kernel * kernel, (long long)kernel * kernel, 0, stream);
}

FN2_API int fn2_conv_mfma_pack_weights_view(const float* weight, float* packed, int Cout, int Cin, int kernel, int src_cout, int src_cin,
                                            long long stride_cout, long long stride_cin, int flip, void* stream) {
  if (!weight || !packed) return fail(FN2_ERR_INVALID_ARG, "conv_mfma_pack_weights_view: null blob");
  if (Cout <= 0 || Cout % 32 != 0 || Cin <= 0 || (kernel != 1 && kernel != 3 && kernel != 4 && kernel != 5 && kernel != 7))
    return fail(FN2_ERR_UNSUPPORTED, "conv_mfma_pack_weights_view: needs Cout %% 32 == 0 and kernel_size 1, 3, 4, 5 or 7 (got Cout %d, kernel %d)", Cout, kernel);
  if (src_cout < 1 || src_cout > Cout || src_cin < 1 || src_cin > Cin || stride_cout < 1 || stride_cin < 1)
    return fail(FN2_ERR_INVALID_ARG, "conv_mfma_pack_weights_view: bad source view (%d of %d, %d of %d channels, strides %lld / %lld)", src_cout, Cout,
                src_cin, Cin, stride_cout, stride_cin);
  const int ksteps = cv::ksteps_for(Cin, kernel), kalloc = ksteps + cv::kSpare;
  hipStream_t st = as_stream(stream);
  switch (kernel) {
    case 1: cv::launch_pack_view<1>(weight, packed, Cout, ksteps, kalloc, src_cout, src_cin, stride_cout, stride_cin, flip ? 1 : 0, st); break;
    case 3: cv::launch_pack_view<3>(weight, packed, Cout, ksteps, kalloc, src_cout, src_cin, stride_cout, stride_cin, flip ? 1 : 0, st); break;
    case 4: cv::launch_pack_view<4>(weight, packed, Cout, ksteps, kalloc, src_cout, src_cin, stride_cout, stride_cin, flip ? 1 : 0, st); break;
    case 5: cv::launch_pack_view<5>(weight, packed, Cout, ksteps, kalloc, src_cout, src_cin, stride_cout, stride_cin, flip ? 1 : 0, st); break;
    default: cv::launch_pack_view<7>(weight, packed, Cout, ksteps, kalloc, src_cout, src_cin, stride_cout, stride_cin, flip ? 1 : 0, st); break;
  }
  return check_launch("conv_mfma_pack_weights_view");
}

FN2_API int fn2_conv_mfma_supported(int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad) {
  if (Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || Cout % 32 != 0) return 0;
  if (kernel == 1 ? ((long long)Hin * Win) % 4 != 0 : Win % 4 != 0) return 0;      // (1x1: the planes are flattened, only their size matters)
  if (!((kernel == 1 && stride == 1) || (kernel == 3 && (stride == 1 || stride == 2)) || (kernel == 4 && stride == 2) || (kernel == 5 && stride == 2) || (kernel == 7 && stride == 2))) return 0;
  if (pad < 0 || pad > 4 || pad > kernel - 1) return 0;
  if ((long long)Cin * Hin * Win >= (1ll << 28)) return 0;
  const int Hout = (Hin + 2 * pad - kernel) / stride + 1, Wout = (Win + 2 * pad - kernel) / stride + 1;
  return Hout >= 1 && Wout >= 1;
}

FN2_API int fn2_debug_set_conv_variant(int v) { cv::g_forced_variant = v; return FN2_OK; }
FN2_API int fn2_conv_mfma_num_variants(void) { return cv::kNumVariants; }

FN2_API int fn2_conv_mfma_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                                  int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                  int Cout, int top_channels, int top_c0, int kernel, int stride, int pad,
                                  int relu, float negative_slope, void* stream) {
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "conv_mfma: bad batch");
  if (N == 0) return FN2_OK;
  if (!bottom || !packed_weight || !top) return fail(FN2_ERR_INVALID_ARG, "conv_mfma: null blob");
  if (!fn2_conv_mfma_supported(Cin, Hin, Win, Cout, kernel, stride, pad))
    return fail(FN2_ERR_UNSUPPORTED, "conv_mfma: unsupported geometry (Cin %d, %dx%d, Cout %d, k %d s %d p %d)", Cin, Hin, Win, Cout, kernel, stride, pad);
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels)
    return fail(FN2_ERR_INVALID_ARG, "conv_mfma: channel slice outside the blob");
  if (((reinterpret_cast<uintptr_t>(bottom) | reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(packed_weight)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "conv_mfma: blobs must be 16-byte aligned");
  cv::Args a{};
  a.in = bottom; a.wp = packed_weight; a.bias = bias; a.out = top;
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.in_ctot = bottom_channels; a.in_c0 = bottom_c0;
  a.Cout = Cout; a.Hout = (Hin + 2 * pad - kernel) / stride + 1; a.Wout = (Win + 2 * pad - kernel) / stride + 1;
  a.out_ctot = top_channels; a.out_c0 = top_c0; a.pad = pad;
  a.ksteps = cv::ksteps_for(Cin, kernel) + cv::kSpare;
  a.slope = negative_slope; a.relu = relu;
  if ((a.Wout % 4) != 0 && ((size_t)a.Wout * sizeof(float)) % 16 != 0) { /* scalar tail stores handle it */ }
  int best = -1;
  bool tail = false;
  if (cv::g_forced_variant >= 0) {
    tail = cv::g_forced_variant >= 1000;
    best = cv::g_forced_variant % 1000;
    if (best >= cv::kNumVariants || !cv::variant_applies(cv::kVariants[best], a, kernel, stride) || (tail && !cv::kVariants[best].fn_tail))
      return fail(FN2_ERR_UNSUPPORTED, "conv_mfma: forced variant %d does not apply", cv::g_forced_variant);
  } else {
    hipStream_t st = as_stream(stream);
    int picked = -1;
    if (autotune_enabled(st)) {
      // candidates 2 i / 2 i + 1 = plain / split-tail launch of variant i; all of them write the same bits
      static TuneCache cache("conv_mfma", cv::kNumVariants);
      auto usable = [&](int c) -> bool {
        const cv::Variant& v = cv::kVariants[c / 2];
        return cv::variant_applies(v, a, kernel, stride) && (!(c & 1) || (v.fn_tail && cv::variant_cost(v, a, true) < 1e29));
      };
      const TuneKey key{N, Cin, Hin, Win, Cout, kernel, stride, pad, bottom_channels == Cin, top_channels == Cout};
      picked = autotune_pick(cache, key, 2 * cv::kNumVariants, st, [&](int c) -> int {
        const cv::Variant& v = cv::kVariants[c / 2];
        if (!cv::variant_applies(v, a, kernel, stride)) return FN2_ERR_UNSUPPORTED;
        if (c & 1) return (v.fn_tail && cv::variant_cost(v, a, true) < 1e29) ? v.fn_tail(a, st) : FN2_ERR_UNSUPPORTED;
        return v.fn(a, st);
      }, usable);
    }
    if (picked >= 0) { best = picked / 2; tail = (picked & 1) != 0; }
    else {
      double bc = 0;
      for (int i = 0; i < cv::kNumVariants; ++i) {
        if (!cv::variant_applies(cv::kVariants[i], a, kernel, stride)) continue;
        for (int t = 0; t < (cv::kVariants[i].fn_tail ? 2 : 1); ++t) {
          const double c = cv::variant_cost(cv::kVariants[i], a, t == 1);
          if (best < 0 || c < bc) { best = i; bc = c; tail = t == 1; }
        }
      }
    }
  }
  if (best < 0) return fail(FN2_ERR_UNSUPPORTED, "conv_mfma: no kernel variant for this geometry");
  return tail ? cv::kVariants[best].fn_tail(a, as_stream(stream)) : cv::kVariants[best].fn(a, as_stream(stream));
}
