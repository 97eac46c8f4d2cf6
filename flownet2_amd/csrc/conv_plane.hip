// Convolutions of the SMALL feature maps of the FlowNet encoders / decoders on v_mfma_f32_16x16x4_f32, whole input planes in LDS:
//   * MODE 0: Convolution{3x3, stride 1 or 2, pad 0 or 1} + bias + ReLU (conv5 .. conv6_1 and friends: 20x28 down to 5x7 pixels,
//     512 .. 1024 channels);
//   * MODE 1: Deconvolution{4x4, stride 2, pad 1} + bias + ReLU (deconv5 .. deconv2: the refinement stages), written straight into a
//     channel slice of the consumer's Concat blob.
//
// Reference: ConvolutionLayer::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23: per sample im2col_gpu + cublasSgemm, then
// forward_gpu_bias -- base_conv_layer.cpp:326-348), DeconvolutionLayer::Forward_gpu (deconv_layer.cu:8-26: per sample
// backward_gpu_gemm = weight^T x bottom + col2im_gpu -- base_conv_layer.cpp:375-393, im2col.cu:246-318 --, then forward_gpu_bias)
// and the in-place ReLU behind both (relu_layer.cu:8-27).
//
// Why its own kernel.  csrc/conv_mfma.hip cuts the output into 4x4 pixel patches and one workgroup tile per (sample, patch block):
// on a 5x7 or 10x14 map most patch slots hang over the edge and there are far fewer tiles than CUs.  Here
//   * the pixels of a GROUP of samples are flattened (sample, y, x) into one pixel axis and cut into MFMA M tiles of 16 consecutive
//     pixel slots: 8 samples of 5x7 are 280 slots = 17.5 tiles instead of 32 patches;
//   * the whole input planes (with their zero border) of the group, CQ channel quads at a time, are staged in LDS in natural
//     [sample][channel][row][column] order by LDS-DMA: one buffer_load ... lds per 64-slot run of a plane, the plane's (sample,
//     channel) address in the scalar offset, so a lane keeps only NPR row/column offsets; border elements are out of range for
//     the buffer descriptor and come back 0.0f.  16-byte DMA when Win % 4 == 0, dword DMA otherwise (rows are not 16-byte aligned);
//   * the pixel operand of tap (ky, kx) for lane (pixel slot, kq) is one ds_read_b32 at  base[pixel tile] + (cq, ky) offset + kx;
//   * the weight operand is packed once per weight blob in MFMA operand order, global -> VGPR with a prefetch ring;
//   * the deconvolution is 4 stride-1 convolutions with 2x2 taps, one per output parity class (Y % 2, X % 2):
//         out[2m + py][2l + px] = sum_{a', b' in {0, 1}} in[m + py - 1 + a'][l + px - 1 + b'] * W[ky(py, a')][kx(px, b')],
//         ky(0, .) = (3, 1), ky(1, .) = (2, 0)  (same for kx);
//     wave w of a workgroup computes class w on the SAME staged window (the class only shifts the window origin by (py, px) and
//     selects its own packed weights), so the window traffic per MFMA is a quarter of a convolution's;
//   * K (the channel quads) is SPLIT over `ksplit` workgroups so that >= 256 workgroups exist; every part writes its partial sums
//     and a second small kernel adds the parts in part order, then bias and ReLU.  ksplit is a function of the layer geometry alone
//     (fn2_conv_plane_ksplit / fn2_deconv_plane_ksplit), never of the tile variant, so all variants produce the same bits: per part
//     a k-ordered fma chain (channel quad, tap row, tap column, channel within the quad), the parts added in part order; part p
//     covers the 2-quad units [p * U / ksplit, (p + 1) * U / ksplit), U = ceil(quads / 2) -- restated by the oracle twins
//     fn2_conv_plane_forward_cpu / fn2_deconv_plane_forward_cpu.
#include "fn2_common.hpp"
#include "autotune.hpp"

namespace fn2 {
namespace cp {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int up_mod(int v, int r, int m) { return v + ((r - v % m) + m) % m; }

constexpr int kNPR = 12;            // at most this many 64-slot DMA runs per (sample, channel) plane
constexpr int kLdsBytes = 160 * 1024;
constexpr int kSpare = 8;           // spare (zero) k-steps behind every packed 64-channel group: the weight prefetch runs ahead

struct Args {
  const float* in; const float* wp; const float* bias; float* out; float* part;
  int N, Cin, Hin, Win, in_ctot, in_c0;
  int Cout, Hout, Wout, out_ctot, out_c0;
  int pad;
  int P;              // pixel slots per sample: Hout * Wout (convolution), Hin * Win (deconvolution: input-resolution pixels per class)
  int Wp;             // pixels per row of that pixel axis
  int img;            // samples per workgroup (their pixels share the M tiles)
  int npb;            // pixel blocks per sample group (> 1 only when img == 1)
  int nig;            // sample groups
  int ng;             // channel blocks: Cout / (16 * MW * WM)
  int ksplit;         // K parts
  int units;          // 2-quad units of the channel axis: ceil(ceil(Cin / 4) / 2)
  int ksteps;         // k-steps of the packed weights per 64-channel group incl. the spare ones
  size_t class_stride;  // floats between the packed weights of two parity classes (deconvolution)
  int rs, cs, wr;     // LDS row stride, channel stride (dwords), window rows
  int band;           // 1: one sample per workgroup and several pixel blocks: the window holds only the rows the block needs
  int cap;            // pixel slots per workgroup (16 * NP * WN)
  int slots_c, npr;   // DMA slots per plane, 64-slot runs per plane
  int buf;            // dwords per window buffer
  unsigned mP, mW, mrs;   // ceil(2^32 / d) for d = P, Wp, rs: q / d == umulhi(q, m) for q * d < 2^32 (fastdiv below)
  unsigned total;
  float slope; int relu;
};

template <int MODE_, int S_, int MW_, int NP_, int WM_, int WN_, int CQ_, int VEC_, int KS_ = 3>
struct Cfg {
  static constexpr int MODE = MODE_, S = S_, MW = MW_, NP = NP_, WM = WM_, WN = WN_, CQ = CQ_, VEC = VEC_, KS = KS_;
  static constexpr int NCLS = MODE == 1 ? 4 : 1;      // parity classes = waves sharing a pixel block
  static constexpr int NW = NCLS * WM * WN, THREADS = 64 * NW;
  static constexpr int TAP = MODE == 1 ? 2 : KS;      // taps per axis (convolution: 3, or 4 for the 4x4 / 2 data gradient of a Deconvolution)
  static constexpr int PADL = VEC == 4 ? 4 : (KS_ == 5 ? 2 : 1);       // window columns left of x = 0 (>= pad)
  static constexpr int KSC = CQ * TAP * TAP;
  static constexpr int NBUFA = KS_ == 5 ? 5 : (KSC % 9 == 0 && KSC > 9) ? 9 : (KSC % 8 == 0) ? 8 : (KSC % 4 == 0) ? 4 : 3;   // weight-operand ring (k-steps)
  static constexpr int CAP = 16 * NP * WN;            // pixel slots of a workgroup
  static_assert(KSC % NBUFA == 0, "ring phase must repeat per chunk");
  static_assert(NBUFA - 1 <= kSpare, "prefetch distance");
  static_assert(MODE == 0 || S == 1, "the deconvolution reads its input at stride 1");
  static_assert(KS == 3 || ((KS == 4 || KS == 5) && MODE == 0 && S == 2), "tap classes: 3x3 / 1, 3x3 / 2, 4x4 / 2, 5x5 / 2");
};

// q / d for the small non-negative values of the index decodes (q < 2^16, d < 2^16); m = ceil(2^32 / d), d == 1 has no 32-bit m
__device__ __forceinline__ int fastdiv(int q, int d, unsigned m) { return d == 1 ? q : (int)__umulhi((unsigned)q, m); }
inline unsigned magic_for(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

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

// LDS-DMA of one chunk: planes (sample il, channel ch of the chunk) wave, wave + NW, ..., every plane as npr runs of 64 slots
// (a __device__ function: the host pass of a __global__ template cannot see amdgcn builtins inside a lambda)
template <class K>
__device__ __forceinline__ void stage_chunk(__amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[kNPR], const Args& a, unsigned dst, int wave,
                                            unsigned chunk_off, unsigned plane_bytes, unsigned img_bytes, int nimg) {
  const int planes = nimg * 4 * K::CQ;                              // samples beyond the batch are not staged (their pixel slots are dropped)
  const int last = a.slots_c - 64;
  for (int pl = wave; pl < planes; pl += K::NW) {
    const int il = pl / (4 * K::CQ), ch = pl % (4 * K::CQ);
    const unsigned soff = __builtin_amdgcn_readfirstlane(chunk_off + (unsigned)il * img_bytes + (unsigned)ch * plane_bytes);
    const unsigned d0 = __builtin_amdgcn_readfirstlane(dst + 4u * (unsigned)(pl * a.cs));
#pragma unroll
    for (int j = 0; j < kNPR; ++j) {
      if (j < a.npr) {
        const int s0 = 64 * j < last ? 64 * j : last;              // the last run overlaps its predecessor instead of spilling
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(d0 + 4u * K::VEC * (unsigned)s0), 4 * K::VEC, voff[j], soff, 0, 0);
      }
    }
  }
}

template <class K>
__global__ void __launch_bounds__(K::THREADS)
conv_plane(Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int S = K::S, MW = K::MW, NP = K::NP, TAP = K::TAP;
  // ---- task: channel block fastest (the blocks that share an input window are neighbours on one XCD: block b runs on XCD b % 8)
  const unsigned per_xcd = (a.total + 7) / 8;
  unsigned t = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (blockIdx.x / 8 >= per_xcd || t >= a.total) return;
  // (readfirstlane: the quotients are wave-uniform, but the compiler divides on the vector ALU; without it the buffer descriptor
  // built from them lives in VGPRs and every LDS-DMA becomes a waterfall loop)
  const int g = __builtin_amdgcn_readfirstlane((int)(t % a.ng)); t /= a.ng;
  const int pb = __builtin_amdgcn_readfirstlane((int)(t % a.npb)); t /= a.npb;
  const int ig = __builtin_amdgcn_readfirstlane((int)(t % a.nig));
  const int kp = __builtin_amdgcn_readfirstlane((int)(t / a.nig));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cls = wave % K::NCLS, w2 = wave / K::NCLS;            // parity class (deconvolution), wave within the class
  const int py = cls >> 1, px = cls & 1;
  const int wm = w2 % K::WM, wn = w2 / K::WM;
  const int n0 = ig * a.img;
  const int nimg = a.N - n0 < a.img ? a.N - n0 : a.img;          // samples really present in this group

  // ---- this part's chunks: 2-quad units [kp * U / ksplit, (kp + 1) * U / ksplit) -> chunks of CQ quads
  const int u0 = __builtin_amdgcn_readfirstlane((int)((long long)kp * a.units / a.ksplit));
  const int u1 = __builtin_amdgcn_readfirstlane((int)((long long)(kp + 1) * a.units / a.ksplit));
  const int chunk0 = u0 * (2 / K::CQ), nchunks = (u1 - u0) * (2 / K::CQ);

  // ---- window rows: the whole plane, or (band) the rows from the block's first pixel row on
  const int yblk = a.band ? fastdiv(pb * a.cap, a.Wp, a.mW) : 0;     // first pixel row of this block
  const int yorg = (K::MODE == 0 ? S * yblk : yblk) - a.pad;          // input row of window row 0

  // ---- LDS-DMA plan
  const size_t plane = (size_t)a.Hin * a.Win;
  const unsigned plane_bytes = 4u * (unsigned)plane, img_bytes = plane_bytes * (unsigned)a.in_ctot;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in + ((size_t)n0 * a.in_ctot + a.in_c0) * plane), 0,
      (unsigned)(img_bytes * (unsigned)(nimg - 1) + plane_bytes * (unsigned)a.Cin), 0x00020000);
  constexpr unsigned OOB = 0x7ffffff0u;
  unsigned voff[kNPR];
#pragma unroll
  for (int j = 0; j < kNPR; ++j) {
    voff[j] = OOB;
    if (j < a.npr) {
      const int last = a.slots_c - 64;
      const int d = K::VEC * ((64 * j < last ? 64 * j : last) + lane);
      const int row = fastdiv(d, a.rs, a.mrs), col = d - row * a.rs;
      const int yi = yorg + row, xi = col - K::PADL;
      if (row < a.wr && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win) voff[j] = 4u * (unsigned)(yi * a.Win + xi);
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const unsigned chunk_bytes = 4u * K::CQ * plane_bytes;

  // ---- operands: pixel slot q = 16 * tile + (lane & 15) -> (sample il, y, x) -> window offset of its first tap
  const int kq = lane >> 4;
  const int tile0 = NP * (pb * K::WN + wn);
  const int lim = nimg * a.P;
  int base[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    int q = 16 * (tile0 + p) + (lane & 15);
    if (q >= lim) q = 16 * tile0 < lim ? 16 * tile0 : 0;          // a slot beyond the group: any address inside the window (its row is dropped)
    const int il = fastdiv(q, a.P, a.mP), rem = q - il * a.P;
    const int y = fastdiv(rem, a.Wp, a.mW), x = rem - y * a.Wp;
    if constexpr (K::MODE == 0) base[p] = (il * 4 * K::CQ + kq) * a.cs + (S * (y - yblk)) * a.rs + S * x + K::PADL - a.pad;
    else base[p] = (il * 4 * K::CQ + kq) * a.cs + (y - yblk + py) * a.rs + x + K::PADL - 1 + px;      // window row of input row r is r + 1 - yblk
  }
  using WV = typename WVec<MW>::T;
  const int cg0 = (g * K::WM + wm) * MW;
  const float* wl = a.wp + (size_t)cls * a.class_stride + ((size_t)(cg0 / 4) * a.ksteps * 64 + lane) * 4 + (cg0 % 4) + (size_t)chunk0 * K::KSC * 256;
  auto wload = [&](int ks) -> WV { return *reinterpret_cast<const WV*>(wl + (size_t)ks * 256); };

  f32x4 acc[MW][NP];
#pragma unroll
  for (int j = 0; j < MW; ++j)
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[j][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  WV wreg[K::NBUFA];
  stage_chunk<K>(rs, voff, a, lds_base, wave, (unsigned)chunk0 * chunk_bytes, plane_bytes, img_bytes, nimg);
#pragma unroll
  for (int i = 0; i < K::NBUFA - 1; ++i) wreg[i] = wload(i);

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    // in flight: this chunk's window (issued a whole chunk ago) and, younger than it, the weight prefetch of the last NBUFA - 1
    // k-steps; loads retire in order, so the window is complete once at most NBUFA - 1 loads remain
    wait_vmcnt<K::NBUFA - 1>();
    __builtin_amdgcn_s_barrier();
    if (c + 1 < nchunks)
      stage_chunk<K>(rs, voff, a, lds_base + 4u * (unsigned)((buf ^ 1) * a.buf), wave, (unsigned)(chunk0 + c + 1) * chunk_bytes, plane_bytes, img_bytes, nimg);
    const float* win = smem + buf * a.buf;
    const int ks0 = c * K::KSC;
    // software pipeline over the (cq, ky) steps of the chunk: the TAP * NP pixel operands of step st + 1 are read from LDS before
    // the TAP * MW * NP MFMAs of step st issue (one wave per SIMD has nobody else to hide the LDS latency behind)
    constexpr int NSTEP = K::CQ * TAP;
    float bb[2][TAP][NP];
    auto lds_step = [&](int st, float (&b)[TAP][NP]) {
      const float* wk = win + ((st / TAP) * 4 * a.cs + (st % TAP) * a.rs);
#pragma unroll
      for (int kx = 0; kx < TAP; ++kx)
#pragma unroll
        for (int p = 0; p < NP; ++p) b[kx][p] = wk[base[p] + kx];
    };
    lds_step(0, bb[0]);
    __builtin_amdgcn_sched_barrier(0);
    // Issue order pinned group by group: the MW MFMAs that share a pixel operand, then ONE LDS read of the next step's operands.  A read
    // issued between MFMAs is free (scripts/probes/mfma_32x32_probe.hip: 153.9 TFLOP/s with one ds_read_b32 per MFMA, 154.1 without); the
    // block of TAP * NP reads the compiler forms when left alone leaves the matrix pipe one queued instruction deep for ~170 cycles a step.
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const float* wkn = win + (((st + 1) / TAP) * 4 * a.cs + ((st + 1) % TAP) * a.rs);
#pragma unroll
      for (int kx = 0; kx < TAP; ++kx) {
        const int ks = st * TAP + kx;
        wreg[(ks + K::NBUFA - 1) % K::NBUFA] = wload(ks0 + ks + K::NBUFA - 1);
        const WV w = wreg[ks % K::NBUFA];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
          for (int j = 0; j < MW; ++j)
            acc[j][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[st & 1][kx][p], wget<MW>(w, j), acc[j][p], 0, 0, 0);
          if (st + 1 < NSTEP) bb[(st + 1) & 1][kx][p] = wkn[base[p] + kx];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  // ---- epilogue: lane (slot block = lane >> 4, channel = lane & 15) holds 4 consecutive pixel slots of tile p
  const bool final_pass = a.ksplit == 1;
  const int Po = a.Hout * a.Wout;
  float* dst = final_pass ? a.out : a.part + (size_t)kp * a.N * a.Cout * Po;
  const int ctot = final_pass ? a.out_ctot : a.Cout, c0 = final_pass ? a.out_c0 : 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int q0 = 16 * (tile0 + p) + 4 * (lane >> 4);
    if (q0 >= lim) continue;
    const int il = fastdiv(q0, a.P, a.mP), pix = q0 - il * a.P;
#pragma unroll
    for (int j = 0; j < MW; ++j) {
      const int co = 16 * (cg0 + j) + (lane & 15);
      const float bv = (final_pass && a.bias) ? a.bias[co] : 0.f;
      f32x4 v = acc[j][p];
      if (final_pass) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = v[r] + bv;
          if (a.relu) s = s > 0.f ? s : s * a.slope;
          v[r] = s;
        }
      }
      if constexpr (K::MODE == 0) {
        float* o = dst + ((size_t)(n0 + il) * ctot + c0 + co) * a.P + pix;
        if (pix + 3 < a.P && (a.P & 3) == 0) *reinterpret_cast<f32x4*>(o) = v;
        else {
          int il2 = il, pix2 = pix;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (q0 + r < lim) dst[((size_t)(n0 + il2) * ctot + c0 + co) * a.P + pix2] = v[r];
            if (++pix2 == a.P) { pix2 = 0; ++il2; }
          }
        }
      } else {
        // input pixel (m, l) of class (py, px) -> output pixel (2 m + py, 2 l + px)
        int il2 = il, m = fastdiv(pix, a.Wp, a.mW), l = pix - m * a.Wp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (q0 + r < lim) dst[(((size_t)(n0 + il2) * ctot + c0 + co) * a.Hout + 2 * m + py) * a.Wout + 2 * l + px] = v[r];
          if (++l == a.Wp) { l = 0; if (++m == a.Hin) { m = 0; ++il2; } }
        }
      }
    }
  }
}

// out[n][c0 + co][pix] = act(bias[co] + ((part0 + part1) + part2) + ...)
// KS > 0: the number of parts at compile time -- all KS loads of an element are issued before the first add (a loop over a run-time part
// count is a chain of memory round trips: load, wait, add, 16 times for conv6_1), the adds keep the part order: the same bits.  KS = 0: any
// part count, four loads at a time.
template <int KS>
__global__ void __launch_bounds__(256) plane_reduce(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
                                                    int N, int Cout, int P, int out_ctot, int out_c0, int ksplit, int relu, float slope) {
  const unsigned total = (unsigned)N * (unsigned)Cout * (unsigned)P;      // (< 2^31: checked by the launcher)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float s;
    if constexpr (KS > 0) {
      float v[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) v[k] = part[(size_t)k * total + i];
      s = v[0];
#pragma unroll
      for (int k = 1; k < KS; ++k) s += v[k];
    } else {
      s = part[i];
      int k = 1;
      for (; k + 3 < ksplit; k += 4) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = part[(size_t)(k + j) * total + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[j];
      }
      for (; k < ksplit; ++k) s += part[(size_t)k * total + i];
    }
    const unsigned r = i / (unsigned)P, pix = i - r * (unsigned)P;
    const unsigned n = r / (unsigned)Cout, co = r - n * (unsigned)Cout;
    s += bias ? bias[co] : 0.f;
    if (relu) s = s > 0.f ? s : s * slope;
    out[((size_t)n * out_ctot + out_c0 + co) * P + pix] = s;
  }
}

// weight [Cin][Cout][SK][SK] (Caffe's deconvolution blob; SK = 3: a 3x3 blob read as the 4x4 one whose fourth tap row and column are zero --
// the transposed 3x3 / 2 / 1 convolution of a data gradient) -> packed [class][Cout/64][k-steps + spare][64][4]:
// lane (co, kq), element j <-> W[4 cq + kq][64 g + 16 j + co][ky(py, a')][kx(px, b')], k-step = (cq * 2 + a') * 2 + b'.
// A workgroup turns 64 output channels x 2 channel quads through LDS: per input channel ONE contiguous run of 64 SK^2 floats in, whole
// 1 KiB k-steps out (the gather form of rounds 2-3 read 4 bytes per 64-byte line: 56 us for deconv5's blob, once per training step).
template <int SK>
__global__ void __launch_bounds__(256) pack_deconv_weights(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int ksteps, int kalloc) {
  constexpr int T = SK * SK, ROW = 8 * T + 1;             // tile [64 co][8 ci][T]
  __shared__ float tile[64 * ROW];
  const int grp = blockIdx.x, u = blockIdx.y;             // unit u = channel quads 2u, 2u + 1
  const int co0 = 64 * grp, ci0 = 8 * u, ngrp = Cout / 64;
#pragma unroll
  for (int i = threadIdx.x; i < 8 * 64 * T; i += 256) {
    const int c = i / (64 * T), e = i - c * (64 * T), col = e / T, t = e - col * T;
    tile[col * ROW + c * T + t] = (ci0 + c < Cin) ? w[((size_t)(ci0 + c) * Cout + co0) * T + e] : 0.f;
  }
  __syncthreads();
  const int j = threadIdx.x & 3, lane = threadIdx.x >> 2;
  const float* trow = tile + (16 * j + (lane & 15)) * ROW + (lane >> 4) * T;
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    float* dst = wp + (((size_t)cls * ngrp + grp) * kalloc + (size_t)8 * u) * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                         // k-steps 8 u + k: cq = 2 u + k / 4
      const int q = k >> 2, ta = (k >> 1) & 1, tb = k & 1;
      const int ky = py == 0 ? (ta == 0 ? 3 : 1) : (ta == 0 ? 2 : 0), kx = px == 0 ? (tb == 0 ? 3 : 1) : (tb == 0 ? 2 : 0);
      dst[k * 256] = (ky < SK && kx < SK) ? trow[4 * q * T + ky * SK + kx] : 0.f;
    }
    if (8 * (u + 1) >= ksteps)                            // the spare k-steps behind the group
      for (int i = threadIdx.x; i < (kalloc - ksteps) * 256; i += 256) wp[(((size_t)cls * ngrp + grp) * kalloc + ksteps) * 256 + i] = 0.f;
  }
}

struct Variant {
  int mode, s, mw, np, wm, wn, cq, vec, ks;
  int (*fn)(const Args&, hipStream_t);
};

// geometry of a variant on a layer; false if it does not apply.  a.P / a.Wp / a.ksplit / a.units are set by the caller.
static bool plan(const Variant& v, Args& a) {
  if (a.Cout % (16 * v.mw * v.wm) != 0) return false;
  if (v.vec == 4 && a.Win % 4 != 0) return false;
  const int padl = v.vec == 4 ? 4 : (v.ks == 5 ? 2 : 1);
  const int cap = 16 * v.np * v.wn;
  a.cap = cap;
  int prow = v.mode == 0 ? a.Hout : a.Hin;                            // pixel rows a workgroup touches: all of them, or (several
  a.band = 0;                                                          // blocks per sample) at most the span of `cap` consecutive pixels
  if (a.P > cap) {
    const int span = (cap + a.Wp - 2) / a.Wp + 1;
    if (span < prow) { prow = span; a.band = 1; }
  }
  if (v.mode == 0) {
    const int wr_need = (prow - 1) * v.s + v.ks;
    a.wr = (!a.band && a.Hin + 2 * a.pad < wr_need) ? a.Hin + 2 * a.pad : wr_need;
  } else {
    a.wr = prow + 2;
  }
  const int wc = padl + a.Win + (v.ks == 5 ? 2 : 1);                    // columns right of the map a tap can reach
  a.rs = v.vec == 4 ? cdiv(wc, 4) * 4 : wc;
  const int plane_dw = a.wr * a.rs < 64 * v.vec ? 64 * v.vec : a.wr * a.rs;     // at least one whole 64-slot DMA run (a 2x3 map has 20 dwords)
  a.cs = up_mod(plane_dw, 16, 32);
  a.slots_c = a.cs / v.vec;
  a.npr = cdiv(a.slots_c, 64);
  if (a.npr > kNPR) return false;
  int img = 1;
  if (a.P < cap && a.Cin % 8 == 0) {      // several samples per workgroup need whole 2-quad units of real channels (no per-sample descriptor range)
    img = cap / a.P;
    if (img > a.N) img = a.N;
    while (img > 1 && 2ll * img * 4 * v.cq * a.cs * 4 > kLdsBytes) --img;
  }
  if (2ll * img * 4 * v.cq * a.cs * 4 > kLdsBytes) return false;
  a.img = img;
  a.nig = cdiv(a.N, img);
  a.npb = img == 1 ? cdiv(a.P, cap) : 1;
  a.ng = a.Cout / (16 * v.mw * v.wm);
  a.buf = img * 4 * v.cq * a.cs;
  if ((long long)a.N * a.P >= 65536 || a.cs >= 65536) return false;          // fastdiv range
  a.mP = magic_for(a.P); a.mW = magic_for(a.Wp); a.mrs = magic_for(a.rs);
  const long long total = (long long)a.ng * a.npb * a.nig * a.ksplit;
  if (total > 0x3fffff00ll) return false;
  a.total = (unsigned)total;
  return true;
}

template <class K>
static int launch(const Args& a, hipStream_t st) {
  const size_t lds = sizeof(float) * 2 * (size_t)a.buf;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_plane<K>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_plane<K>), dim3(8 * ((a.total + 7) / 8)), dim3(K::THREADS), lds, st, a);
  return check_launch("conv_plane_forward");
}

// (MODE, S, MW, NP, WM, WN, CQ, VEC)
#define FN2_CP_TILES(X, S, VEC) \
  X(0, S, 2, 9, 2, 2, 2, VEC) X(0, S, 2, 5, 2, 2, 2, VEC) X(0, S, 2, 7, 2, 2, 2, VEC) X(0, S, 4, 5, 1, 4, 2, VEC) X(0, S, 2, 9, 1, 4, 1, VEC) \
  X(0, S, 4, 9, 1, 4, 1, VEC) X(0, S, 2, 9, 1, 4, 2, VEC) X(0, S, 4, 4, 2, 2, 2, VEC) \
  /* eight waves on one staged window (two per SIMD where LDS leaves room for one workgroup only) */ \
  X(0, S, 2, 9, 2, 4, 2, VEC) X(0, S, 2, 5, 2, 4, 2, VEC) X(0, S, 2, 5, 4, 2, 2, VEC)
#define FN2_DP_TILES(X, VEC) \
  X(1, 1, 2, 9, 1, 1, 2, VEC) X(1, 1, 2, 9, 1, 1, 1, VEC) X(1, 1, 4, 9, 1, 1, 1, VEC) X(1, 1, 4, 9, 1, 1, 2, VEC) X(1, 1, 2, 5, 1, 1, 2, VEC) \
  X(1, 1, 4, 5, 1, 1, 2, VEC) X(1, 1, 2, 7, 1, 1, 2, VEC) X(1, 1, 4, 7, 1, 1, 1, VEC)
#define FN2_CP_LIST(X) FN2_CP_TILES(X, 1, 1) FN2_CP_TILES(X, 1, 4) FN2_CP_TILES(X, 2, 1) FN2_CP_TILES(X, 2, 4) FN2_DP_TILES(X, 1) FN2_DP_TILES(X, 4)
#define FN2_CP_ROW(MODE, S, MW, NP, WM, WN, CQ, VEC) {MODE, S, MW, NP, WM, WN, CQ, VEC, MODE == 1 ? 4 : 3, &launch<Cfg<MODE, S, MW, NP, WM, WN, CQ, VEC>>},
// 4x4 taps at stride 2 (the data gradient of a Deconvolution{4, 2, 1} on a small map: deconv_layer.cu:52-56 = forward_gpu_gemm of top_diff)
#define FN2_CP4_TILES(X, VEC) X(2, 9, 2, 2, 2, VEC) X(2, 5, 2, 2, 2, VEC) X(4, 5, 1, 4, 2, VEC) X(2, 5, 2, 4, 2, VEC)
#define FN2_CP4_ROW(MW, NP, WM, WN, CQ, VEC) {0, 2, MW, NP, WM, WN, CQ, VEC, 4, &launch<Cfg<0, 2, MW, NP, WM, WN, CQ, VEC, 4>>},
// 5x5 taps at stride 2 / pad 2 (conv2 / conv3 of the encoders when ONE sample has to fill the chip: FlowNet2 at batch 1 -- the direct kernel
// of conv_mfma.hip has no K split and ran conv3 [1,128,112,256] -> 256 at 66 TFLOP/s, behind the library's 81); one channel quad per chunk: a
// row band of a 256-pixel-wide map is 12 KB per channel
#define FN2_CP5_TILES(X, VEC) X(2, 9, 2, 2, 1, VEC) X(2, 5, 2, 2, 1, VEC) X(4, 5, 1, 4, 1, VEC) X(2, 9, 1, 4, 1, VEC) X(2, 7, 2, 2, 1, VEC) X(4, 4, 2, 2, 1, VEC)
#define FN2_CP5_ROW(MW, NP, WM, WN, CQ, VEC) {0, 2, MW, NP, WM, WN, CQ, VEC, 5, &launch<Cfg<0, 2, MW, NP, WM, WN, CQ, VEC, 5>>},
static const Variant kVariants[] = {FN2_CP_LIST(FN2_CP_ROW) FN2_CP4_TILES(FN2_CP4_ROW, 1) FN2_CP4_TILES(FN2_CP4_ROW, 4)
                                    FN2_CP5_TILES(FN2_CP5_ROW, 1) FN2_CP5_TILES(FN2_CP5_ROW, 4)};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int g_forced_variant = -1;
int g_forced_ksplit = 0;

// K parts: enough workgroups for 256 CUs at about 56 accumulator tiles per workgroup, a power of two, at most one part per
// 2-quad unit.  A function of the layer geometry ONLY: it fixes the summation order.
static int ksplit_for(long long mfma_tiles, int units) {
  int k = 1;
  if (g_forced_ksplit > 0) k = g_forced_ksplit;
  else {
    const double want = 14336.0 / (double)mfma_tiles;
    while (k < 16 && (double)k < want) k *= 2;
    if ((double)k > 1.5 * want && k > 1) k /= 2;                  // round to the nearer power of two
  }
  while (k > 1 && k > units) k /= 2;
  return k;
}

static double variant_cost(const Variant& v, const Args& a) {
  const double rounds = (double)((a.total + 255) / 256);
  const double eff = 1.0 / (1.0 + 0.08 * (4.0 / v.mw - 1.0) + 0.02 * (9.0 / v.np - 1.0));
  return rounds * v.mw * v.np / eff;
}

// One forward call of either mode: pick the variant (forced / autotuned / cost model), launch, reduce the K parts.
static int forward(Args a, int mode, int stride, int ks, void* workspace, size_t workspace_bytes, const char* what, hipStream_t st) {
  const int Po = a.Hout * a.Wout;
  if (a.ksplit > 1) {
    const size_t need = sizeof(float) * (size_t)a.ksplit * a.N * a.Cout * Po;
    if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "%s: workspace of %zu bytes needed", what, need);
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return fail(FN2_ERR_UNSUPPORTED, "%s: workspace must be 16-byte aligned", what);
    a.part = static_cast<float*>(workspace);
  }
  auto run = [&](int i) -> int {
    const Variant& v = kVariants[i];
    Args t = a;
    if (v.mode != mode || v.s != stride || v.ks != ks || !plan(v, t)) return FN2_ERR_UNSUPPORTED;
    return v.fn(t, st);
  };
  int best = -1;
  if (g_forced_variant >= 0) {
    best = g_forced_variant;
    if (best >= kNumVariants) return fail(FN2_ERR_UNSUPPORTED, "%s: no variant %d", what, best);
  } else {
    if (autotune_enabled(st)) {
      static TuneCache cache_conv("conv_plane", kNumVariants), cache_deconv("deconv_plane", kNumVariants);
      auto usable = [&](int i) -> bool {
        Args t = a;
        return kVariants[i].mode == mode && kVariants[i].s == stride && kVariants[i].ks == ks && plan(kVariants[i], t);
      };
      const TuneKey key{a.N, a.Cin, a.Hin, a.Win, a.Cout, stride + 16 * ks, a.pad, a.ksplit, a.in_ctot == a.Cin, a.out_ctot == a.Cout};
      best = autotune_pick(mode == 0 ? cache_conv : cache_deconv, key, kNumVariants, st, run, usable);
    }
    if (best < 0) {
      double bc = 0;
      for (int i = 0; i < kNumVariants; ++i) {
        Args t = a;
        if (kVariants[i].mode != mode || kVariants[i].s != stride || kVariants[i].ks != ks || !plan(kVariants[i], t)) continue;
        const double c = variant_cost(kVariants[i], t);
        if (best < 0 || c < bc) { best = i; bc = c; }
      }
    }
  }
  if (best < 0) return fail(FN2_ERR_UNSUPPORTED, "%s: no kernel variant for this geometry", what);
  const int rc = run(best);
  if (rc == FN2_ERR_UNSUPPORTED) return fail(FN2_ERR_UNSUPPORTED, "%s: variant %d does not apply to this geometry", what, best);
  if (rc != FN2_OK) return rc;
  if (a.ksplit > 1) {
    const long long total = (long long)a.N * a.Cout * Po;
    if (total >= (1ll << 31)) return fail(FN2_ERR_UNSUPPORTED, "%s: blob too large for the part reduction", what);
    const dim3 rg(blocks_for(total, 256, 4096));
#define FN2_PR(KS_) hipLaunchKernelGGL(plane_reduce<KS_>, rg, dim3(256), 0, st, a.part, a.bias, a.out, a.N, a.Cout, Po, a.out_ctot, a.out_c0, a.ksplit, a.relu, a.slope)
    switch (a.ksplit) {
      case 2: FN2_PR(2); break;
      case 3: FN2_PR(3); break;
      case 4: FN2_PR(4); break;
      case 6: FN2_PR(6); break;
      case 8: FN2_PR(8); break;
      case 12: FN2_PR(12); break;
      case 16: FN2_PR(16); break;
      default: FN2_PR(0); break;
    }
#undef FN2_PR
    return check_launch(what);
  }
  return FN2_OK;
}

static bool supported(const Args& a, int mode, int stride, int ks) {
  for (int i = 0; i < kNumVariants; ++i) {
    Args t = a;
    if (kVariants[i].mode == mode && kVariants[i].s == stride && kVariants[i].ks == ks && plan(kVariants[i], t)) return true;
  }
  return false;
}

}  // namespace cp
}  // namespace fn2

using namespace fn2;

// ------------------------------------------------------------------------------------------------ convolution 3x3 (and 4x4 / 2)
static bool plane_geometry_ok(int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad) {
  if (N <= 0 || Cin <= 0 || Cin % 8 != 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || Cout % 64 != 0) return false;
  if ((stride != 1 && stride != 2) || pad < 0 || pad > 2) return false;
  if (!((kernel == 3 && pad <= 1) || (kernel == 4 && stride == 2 && pad == 1) || (kernel == 5 && stride == 2 && pad == 2))) return false;
  if (Hin + 2 * pad < kernel || Win + 2 * pad < kernel) return false;
  if ((long long)N * Cin * Hin * Win >= (1ll << 28)) return false;
  return true;
}

static void fill_args(cp::Args& a, int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad) {
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.pad = pad;
  a.Hout = (Hin + 2 * pad - kernel) / stride + 1; a.Wout = (Win + 2 * pad - kernel) / stride + 1;
  a.P = a.Hout * a.Wout; a.Wp = a.Wout;
  a.units = ((Cin + 3) / 4 + 1) / 2;
  a.ksplit = cp::ksplit_for((long long)cp::cdiv(order_batch(N) * a.P, 16) * (Cout / 16), a.units);
  a.ksteps = a.units * 2 * kernel * kernel + cp::kSpare;          // fn2_conv_mfma_pack_weights: whole chunks of 2 quads + 8 spare k-steps
  a.class_stride = 0;
}

FN2_API int fn2_conv_plane_k_supported(int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad) {
  if (!plane_geometry_ok(N, Cin, Hin, Win, Cout, kernel, stride, pad)) return 0;
  cp::Args a{};
  fill_args(a, N, Cin, Hin, Win, Cout, kernel, stride, pad);
  return cp::supported(a, 0, stride, kernel) ? 1 : 0;
}

FN2_API int fn2_conv_plane_k_ksplit(int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad) {
  if (!plane_geometry_ok(N, Cin, Hin, Win, Cout, kernel, stride, pad)) return 0;
  cp::Args a{};
  fill_args(a, N, Cin, Hin, Win, Cout, kernel, stride, pad);
  return a.ksplit;
}

FN2_API size_t fn2_conv_plane_k_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad) {
  if (!plane_geometry_ok(N, Cin, Hin, Win, Cout, kernel, stride, pad)) return 0;
  cp::Args a{};
  fill_args(a, N, Cin, Hin, Win, Cout, kernel, stride, pad);
  return a.ksplit > 1 ? sizeof(float) * (size_t)a.ksplit * N * Cout * a.P : 0;
}

FN2_API int fn2_conv_plane_supported(int N, int Cin, int Hin, int Win, int Cout, int stride, int pad) {
  return fn2_conv_plane_k_supported(N, Cin, Hin, Win, Cout, 3, stride, pad);
}
FN2_API int fn2_conv_plane_ksplit(int N, int Cin, int Hin, int Win, int Cout, int stride, int pad) {
  return fn2_conv_plane_k_ksplit(N, Cin, Hin, Win, Cout, 3, stride, pad);
}
FN2_API size_t fn2_conv_plane_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout, int stride, int pad) {
  return fn2_conv_plane_k_workspace_bytes(N, Cin, Hin, Win, Cout, 3, stride, pad);
}

FN2_API int fn2_debug_set_plane_variant(int v) { cp::g_forced_variant = v; return FN2_OK; }
FN2_API int fn2_debug_set_plane_ksplit(int k) { cp::g_forced_ksplit = k; return FN2_OK; }
FN2_API int fn2_conv_plane_num_variants(void) { return cp::kNumVariants; }

FN2_API int fn2_conv_plane_k_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                                     int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                     int Cout, int top_channels, int top_c0, int kernel, int stride, int pad,
                                     int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "conv_plane: bad batch");
  if (N == 0) return FN2_OK;
  if (!bottom || !packed_weight || !top) return fail(FN2_ERR_INVALID_ARG, "conv_plane: null blob");
  if (!plane_geometry_ok(N, Cin, Hin, Win, Cout, kernel, stride, pad))
    return fail(FN2_ERR_UNSUPPORTED, "conv_plane: unsupported geometry (N %d, Cin %d, %dx%d, Cout %d, k %d s %d p %d)", N, Cin, Hin, Win, Cout, kernel, stride, pad);
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels)
    return fail(FN2_ERR_INVALID_ARG, "conv_plane: channel slice outside the blob");
  if (((reinterpret_cast<uintptr_t>(bottom) | reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(packed_weight)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "conv_plane: blobs must be 16-byte aligned");
  if ((long long)bottom_channels * Hin * Win * 4 * N >= 0x7ffffff0ll) return fail(FN2_ERR_UNSUPPORTED, "conv_plane: bottom blob too large");
  cp::Args a{};
  fill_args(a, N, Cin, Hin, Win, Cout, kernel, stride, pad);
  a.in = bottom; a.wp = packed_weight; a.bias = bias; a.out = top;
  a.in_ctot = bottom_channels; a.in_c0 = bottom_c0; a.out_ctot = top_channels; a.out_c0 = top_c0;
  a.slope = negative_slope; a.relu = relu;
  return cp::forward(a, 0, stride, kernel, workspace, workspace_bytes, "conv_plane", as_stream(stream));
}

FN2_API int fn2_conv_plane_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                                   int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                   int Cout, int top_channels, int top_c0, int stride, int pad,
                                   int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream) {
  return fn2_conv_plane_k_forward(bottom, packed_weight, bias, top, N, Cin, Hin, Win, bottom_channels, bottom_c0, Cout, top_channels, top_c0,
                                  3, stride, pad, relu, negative_slope, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ deconvolution 4x4 / 2
static bool deconv_geometry_ok(int N, int Cin, int Hin, int Win, int Cout) {
  if (N <= 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0 || Cout % 64 != 0) return false;
  if ((long long)N * Cin * Hin * Win >= (1ll << 28) || (long long)N * Cout * Hin * Win >= (1ll << 27)) return false;
  return true;
}

static void fill_deconv_args(cp::Args& a, int N, int Cin, int Hin, int Win, int Cout) {
  a.N = N; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.pad = 1;
  a.Hout = 2 * Hin; a.Wout = 2 * Win;
  a.P = Hin * Win; a.Wp = Win;
  a.units = ((Cin + 3) / 4 + 1) / 2;
  a.ksplit = cp::ksplit_for(4ll * cp::cdiv(order_batch(N) * a.P, 16) * (Cout / 16), a.units);
  a.ksteps = a.units * 2 * 4 + cp::kSpare;
  a.class_stride = (size_t)(Cout / 64) * a.ksteps * 256;
}

FN2_API int fn2_deconv_plane_supported(int N, int Cin, int Hin, int Win, int Cout) {
  if (!deconv_geometry_ok(N, Cin, Hin, Win, Cout)) return 0;
  cp::Args a{};
  fill_deconv_args(a, N, Cin, Hin, Win, Cout);
  return cp::supported(a, 1, 1, 4) ? 1 : 0;
}

FN2_API int fn2_deconv_plane_ksplit(int N, int Cin, int Hin, int Win, int Cout) {
  if (!deconv_geometry_ok(N, Cin, Hin, Win, Cout)) return 0;
  cp::Args a{};
  fill_deconv_args(a, N, Cin, Hin, Win, Cout);
  return a.ksplit;
}

FN2_API size_t fn2_deconv_plane_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout) {
  if (!deconv_geometry_ok(N, Cin, Hin, Win, Cout)) return 0;
  cp::Args a{};
  fill_deconv_args(a, N, Cin, Hin, Win, Cout);
  return a.ksplit > 1 ? sizeof(float) * (size_t)a.ksplit * N * Cout * a.Hout * a.Wout : 0;
}

FN2_API size_t fn2_deconv_plane_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cout % 64 != 0) return 0;
  return 4 * (size_t)(Cout / 64) * ((((Cin + 3) / 4 + 1) / 2) * 2 * 4 + cp::kSpare) * 256;
}

FN2_API int fn2_deconv_plane_pack_weights_k(const float* weight, float* packed, int Cin, int Cout, int src_kernel, void* stream) {
  if (!weight || !packed) return fail(FN2_ERR_INVALID_ARG, "deconv_plane_pack_weights: null blob");
  if (Cin <= 0 || Cout <= 0 || Cout % 64 != 0) return fail(FN2_ERR_UNSUPPORTED, "deconv_plane_pack_weights: needs Cout %% 64 == 0 (got %d)", Cout);
  if (src_kernel != 3 && src_kernel != 4) return fail(FN2_ERR_UNSUPPORTED, "deconv_plane_pack_weights: the blob has 4x4 taps, or 3x3 (zero-padded to 4x4)");
  const int units = ((Cin + 3) / 4 + 1) / 2, ksteps = units * 2 * 4, kalloc = ksteps + cp::kSpare;
  const dim3 grid((unsigned)(Cout / 64), (unsigned)units);
  if (src_kernel == 4) hipLaunchKernelGGL((cp::pack_deconv_weights<4>), grid, dim3(256), 0, as_stream(stream), weight, packed, Cin, Cout, ksteps, kalloc);
  else hipLaunchKernelGGL((cp::pack_deconv_weights<3>), grid, dim3(256), 0, as_stream(stream), weight, packed, Cin, Cout, ksteps, kalloc);
  return check_launch("deconv_plane_pack_weights");
}

FN2_API int fn2_deconv_plane_pack_weights(const float* weight, float* packed, int Cin, int Cout, void* stream) {
  return fn2_deconv_plane_pack_weights_k(weight, packed, Cin, Cout, 4, stream);
}

FN2_API int fn2_deconv_plane_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                                     int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                                     int Cout, int top_channels, int top_c0, int relu, float negative_slope,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "deconv_plane: bad batch");
  if (N == 0) return FN2_OK;
  if (!bottom || !packed_weight || !top) return fail(FN2_ERR_INVALID_ARG, "deconv_plane: null blob");
  if (!deconv_geometry_ok(N, Cin, Hin, Win, Cout))
    return fail(FN2_ERR_UNSUPPORTED, "deconv_plane: unsupported geometry (N %d, Cin %d, %dx%d, Cout %d)", N, Cin, Hin, Win, Cout);
  if (bottom_c0 < 0 || bottom_c0 + Cin > bottom_channels || top_c0 < 0 || top_c0 + Cout > top_channels)
    return fail(FN2_ERR_INVALID_ARG, "deconv_plane: channel slice outside the blob");
  if (((reinterpret_cast<uintptr_t>(bottom) | reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(packed_weight)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "deconv_plane: blobs must be 16-byte aligned");
  if ((long long)bottom_channels * Hin * Win * 4 * N >= 0x7ffffff0ll) return fail(FN2_ERR_UNSUPPORTED, "deconv_plane: bottom blob too large");
  cp::Args a{};
  fill_deconv_args(a, N, Cin, Hin, Win, Cout);
  a.in = bottom; a.wp = packed_weight; a.bias = bias; a.out = top;
  a.in_ctot = bottom_channels; a.in_c0 = bottom_c0; a.out_ctot = top_channels; a.out_c0 = top_c0;
  a.slope = negative_slope; a.relu = relu;
  return cp::forward(a, 1, 1, 4, workspace, workspace_bytes, "deconv_plane", as_stream(stream));
}
