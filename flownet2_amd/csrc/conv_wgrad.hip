// Weight gradient of Convolution / Deconvolution layers on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chains), NCHW in,
// Caffe weight layout out, no layout transposes, deterministic.
//
//     dw[ca][cb][ky][kx] (+)= sum_{n, y, x}  a[n][ca][y][x] * b[n][cb][S*y + ky - pad][S*x + kx - pad]        (zero outside b)
//
// Reference: ConvolutionLayer::Backward_gpu -> weight_gpu_gemm (src/caffe/layers/conv_layer.cu:40-52, base_conv_layer.cpp:368-384:
// per SAMPLE im2col_gpu + cublasSgemm(top_diff x col^T) accumulated into weight_diff with beta = 1), a = top_diff, b = bottom, dw =
// weight_diff [Cout][Cin][k][k]; DeconvolutionLayer::Backward_gpu (deconv_layer.cu:36-50: weight_gpu_gemm(top_diff, bottom) -- the roles
// swapped), a = bottom (the small map), b = top_diff, dw = weight_diff [Cin][Cout][k][k].  The bias gradient stays in bias_act.hip.
//
// GEMM view: M = 16 `a` channels, N = 16 `b` channels, K = pixels -- 4 consecutive x of one row per MFMA k-step.  A wave owns MA x NB
// channel-group pairs x ALL KS*KS taps (one accumulator tile each): per k-step it reads MA `a` operands and NB*KS*KS `b` operands (one
// ds_read_b32 at lane base + immediate each) for MA*NB*KS*KS MFMAs.  A workgroup (4 waves, one per SIMD) walks a contiguous range of
// rows of the `a` map ("K part"), chunk by chunk (R rows x XT pixels): the chunk's `a` pixels [CA][R][XT] and the window of `b` it needs
// [CB][S(R-1)+KS][S(XT-1)+KS+4] arrive by 16-byte LDS-DMA straight from NCHW (rows / columns outside the map, channels beyond the blob
// and pixels beyond the row are out of range for the buffer descriptor: 0.0f = the zero padding), two buffers, one barrier per chunk.
// Every part writes its accumulator tiles as they are (16-byte stores, MFMA layout); wgrad_finalize adds the parts in part order and
// scatters into the weight layout.  Summation order (restated by the oracle twin fn2_conv_wgrad_cpu): per part one fma chain over the
// part's pixels in (n, y, x) order, parts added in part order; ksplit is a function of the layer geometry only (conv_wgrad_geom.hpp).
#include "fn2_common.hpp"
#include "conv_wgrad_geom.hpp"

namespace fn2 {
namespace wg {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;

struct Args {
  const float* a; const float* b; float* slab;
  int N, Ca, Ha, Wa, a_ctot, a_c0;
  int Cb, Hb, Wb, b_ctot, b_c0;
  int pad, ksplit, nblk_a, nblk_b, nxb, U;
  unsigned total;            // workgroups: nblk_a * nblk_b * ksplit
  long long slab_part;       // floats per part
};

constexpr unsigned kOOB = 0x7ffffff0u;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Chunk { int n, y, rvalid, x0; };

// LDS-DMA of one chunk into the buffer at LDS byte address `dst` (a __device__ function, not a lambda: the host pass of a __global__
// template cannot see the amdgcn builtins inside a lambda body)
template <class K>
__device__ __forceinline__ void stage_chunk(const Args& a, const Chunk& c, int ca0, int cb0, unsigned dst, int wave,
                                            const unsigned (&pa_off)[K::RPW_A], const unsigned (&pa_rc)[K::RPW_A],
                                            const unsigned (&pb_off)[K::RPW_B], const unsigned (&pb_rc)[K::RPW_B]) {
  const size_t planeA = (size_t)a.Ha * a.Wa, planeB = (size_t)a.Hb * a.Wb;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.a + ((size_t)c.n * a.a_ctot + a.a_c0 + ca0) * planeA), 0, (unsigned)(4u * K::CA * planeA), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.b + ((size_t)c.n * a.b_ctot + a.b_c0 + cb0) * planeB), 0, (unsigned)(4u * K::CB * planeB), 0x00020000);
  const unsigned originA = (unsigned)(c.y * a.Wa + c.x0);
#pragma unroll
  for (int i = 0; i < K::RPW_A; ++i) {
    const int run = i * K::NW + wave;
    if (run < K::NRUN_A) {
      const unsigned r = pa_rc[i] & 0xffu, x = pa_rc[i] >> 8;
      const bool ok = (int)r < c.rvalid && c.x0 + (int)x < a.Wa;          // (an entry outside the image / blob has r = 255)
      const unsigned voff = ok ? 4u * (pa_off[i] + originA) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)run), 16, voff, 0, 0, 0);
    }
  }
  const int gy0 = K::S * c.y - a.pad, gx0 = K::S * c.x0 - K::PADL;
#pragma unroll
  for (int i = 0; i < K::RPW_B; ++i) {
    const int run = i * K::NW + wave;
    if (run < K::NRUN_B) {
      const int gy = gy0 + (int)(pb_rc[i] & 0xffffu), gx = gx0 + (int)(pb_rc[i] >> 16);
      const bool ok = (unsigned)gy < (unsigned)a.Hb && (unsigned)gx < (unsigned)a.Wb;
      const unsigned voff = ok ? 4u * (pb_off[i] + (unsigned)(gy * a.Wb + gx)) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(uintptr_t)(dst + 4u * K::A_DW + 1024u * (unsigned)run), 16, voff, 0, 0, 0);
    }
  }
}

// LDS operand reads are VOLATILE address-space-3 loads: the compiler keeps them as single ds_read_b32 at  lane base + immediate  (its
// load/store optimizer otherwise pairs neighbours into ds_read2_b32, whose 8-bit offsets need a VALU add per pair -- 0.5 VALU
// instructions per MFMA, which cost matrix-pipe time) and in program order, which is the software pipeline written below.
using lds_vf = const volatile __attribute__((address_space(3))) float*;

// One chunk's MFMAs from the staging buffer whose lane bases are the LDS BYTE addresses a_addr / b_addr: rows r < rvalid, every k-step
// of a row (pixels beyond the row end are zeros in the `a` image: fma(0, b, acc) == acc).  Per row a software pipeline: the operands of
// k-step xq + 1 are read one by one between the MFMAs of k-step xq (one ds_read per MA MFMAs), so every read has a whole k-step of
// MFMAs to land; only the first k-step of a row waits for its reads.
template <class K>
__device__ __forceinline__ void compute_chunk(unsigned a_addr, unsigned b_addr, int rvalid, f32x4 (&acc)[K::TILES]) {
  constexpr int MA = K::MA, NB = K::NB, KS = K::KS, T = K::T, NBT = NB * T, NXQ = K::XT / 4;
  static_assert(NBT >= MA, "the a operands ride along with the first b operands");
  lds_vf Ab = (lds_vf)(uintptr_t)a_addr;
  lds_vf Bb = (lds_vf)(uintptr_t)b_addr;
#pragma unroll
  for (int r = 0; r < K::R; ++r) {
    if (r < rvalid) {
      float av[2][MA], bv[2][NBT];
#pragma unroll
      for (int ma = 0; ma < MA; ++ma) av[0][ma] = Ab[a_step_off<K>(ma, r, 0)];
#pragma unroll
      for (int j = 0; j < NBT; ++j) bv[0][j] = Bb[b_step_off<K>(j / T, r, 0, (j % T) / KS, j % KS)];
#pragma unroll
      for (int xq = 0; xq < NXQ; ++xq) {
        const int cur = xq & 1, nxt = cur ^ 1;
#pragma unroll
        for (int j = 0; j < NBT; ++j) {
          if (xq + 1 < NXQ) {
            if (j < MA) av[nxt][j] = Ab[a_step_off<K>(j, r, xq + 1)];
            bv[nxt][j] = Bb[b_step_off<K>(j / T, r, xq + 1, (j % T) / KS, j % KS)];
            __builtin_amdgcn_sched_barrier(0);       // hipcc otherwise sinks every read to its first use: read, wait, two MFMAs, read, wait ...
          }
#pragma unroll
          for (int ma = 0; ma < MA; ++ma) {
            const int ti = tile_index<K>(ma, j / T, j % T);
            acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][ma], bv[cur][j], acc[ti], 0, 0, 0);
          }
          if (xq + 1 < NXQ) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}

template <class K>
__global__ void __launch_bounds__(256, K::WG_PER_CU) conv_wgrad(Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % K::WM, wn = wave / K::WM;

  // ---- task: (K part, channel block); the blocks of one part share its pixels -> contiguous on one XCD (block b runs on XCD b % 8)
  const unsigned per_xcd = (a.total + 7) / 8;
  const unsigned t = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (blockIdx.x / 8 >= per_xcd || t >= a.total) return;
  const int nblk = a.nblk_a * a.nblk_b;
  const int part = (int)(t / (unsigned)nblk), blk = (int)(t % (unsigned)nblk);
  const int ca0 = (blk / a.nblk_b) * K::CA, cb0 = (blk % a.nblk_b) * K::CB;
  const int u0 = part_begin(part, a.ksplit, a.U), u1 = part_begin(part + 1, a.ksplit, a.U);

  // ---- LDS-DMA plan: run = i * NW + wave, slot = 64 * run + lane
  const unsigned planeA = (unsigned)(a.Ha * a.Wa), planeB = (unsigned)(a.Hb * a.Wb);
  unsigned pa_off[K::RPW_A], pa_rc[K::RPW_A], pb_off[K::RPW_B], pb_rc[K::RPW_B];
#pragma unroll
  for (int i = 0; i < K::RPW_A; ++i) {
    const SlotA s = slot_a<K>((i * K::NW + wave) * 64 + lane);
    const bool ok = s.in_image && ca0 + s.ch < a.Ca;
    pa_off[i] = ok ? (unsigned)s.ch * planeA + (unsigned)(s.r * a.Wa + s.x) : 0u;
    pa_rc[i] = ok ? ((unsigned)s.r | ((unsigned)s.x << 8)) : 0xffffffffu;
  }
#pragma unroll
  for (int i = 0; i < K::RPW_B; ++i) {
    const SlotB s = slot_b<K>((i * K::NW + wave) * 64 + lane);
    const bool ok = s.in_image && cb0 + s.ch < a.Cb;
    pb_off[i] = ok ? (unsigned)s.ch * planeB : 0u;
    pb_rc[i] = ok ? ((unsigned)s.wr | ((unsigned)s.wc << 16)) : 0x7fffu;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

  // ---- chunk cursor: rows u of the part, x blocks of a row (R > 1 only with one x block per row: pixel order stays (n, y, x))
  auto chunk_at = [&](int u, int xb) -> Chunk {
    const int n = u / a.Ha, y = u - n * a.Ha;
    int rv = a.Ha - y;
    if (rv > u1 - u) rv = u1 - u;
    if (rv > K::R) rv = K::R;
    return Chunk{n, y, rv, xb * K::XT};
  };

  f32x4 acc[K::TILES];
#pragma unroll
  for (int i = 0; i < K::TILES; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int abase = a_lane_base<K>(wm, lane), bbase = b_lane_base<K>(wn, lane, a.pad);
  int u = u0, xb = 0;
  Chunk c = chunk_at(u, xb);
  if constexpr (K::DB) {
    int cur = 0;
    stage_chunk<K>(a, c, ca0, cb0, lds_base, wave, pa_off, pa_rc, pb_off, pb_rc);
    while (true) {
      int un = u, xbn = xb + 1;
      if (xbn >= a.nxb) { xbn = 0; un = u + c.rvalid; }
      const bool has_next = un < u1;
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      Chunk cn = c;
      if (has_next) {
        cn = chunk_at(un, xbn);
        stage_chunk<K>(a, cn, ca0, cb0, lds_base + 4u * (unsigned)((cur ^ 1) * K::BUF), wave, pa_off, pa_rc, pb_off, pb_rc);
      }
      compute_chunk<K>(lds_base + 4u * (unsigned)(cur * K::BUF + abase), lds_base + 4u * (unsigned)(cur * K::BUF + bbase), c.rvalid, acc);
      if (!has_next) break;
      cur ^= 1; u = un; xb = xbn; c = cn;
    }
  } else {
    while (true) {
      stage_chunk<K>(a, c, ca0, cb0, lds_base, wave, pa_off, pa_rc, pb_off, pb_rc);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      compute_chunk<K>(lds_base + 4u * (unsigned)abase, lds_base + 4u * (unsigned)bbase, c.rvalid, acc);
      ++xb;
      if (xb >= a.nxb) { xb = 0; u += c.rvalid; }
      if (u >= u1) break;
      c = chunk_at(u, xb);
      __builtin_amdgcn_s_barrier();          // every wave has read this chunk before the next one lands
    }
  }

  // ---- epilogue: the accumulator tiles as they are (lane l holds D[4 (l >> 4) + j][l & 15], j = 0..3), 16-byte stores
  float* dst = a.slab + (size_t)part * (size_t)a.slab_part + (((size_t)blk * 4 + wave) * K::TILES) * 256 + lane * 4;
#pragma unroll
  for (int i = 0; i < K::TILES; ++i) *reinterpret_cast<f32x4*>(dst + (size_t)i * 256) = acc[i];
}

// Adds the K parts in part order and scatters into the weight layout [Ca][Cb][T]: one thread per 4 consecutive slab floats
// (= 4 consecutive `a` channels of one (`b` channel, tap)), reads coalesced over the slab, 4-byte scattered writes.
__global__ void __launch_bounds__(256) wgrad_finalize(const float* __restrict__ slab, float* __restrict__ dw, SlabMap m, int Ca, int Cb, int nblk_b,
                                                      long long quads, int ksplit, long long slab_part, int accumulate) {
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < quads; q += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(q & 63);
    long long r = q >> 6;
    const int tile = (int)(r % m.TILES); r /= m.TILES;
    const int wave = (int)(r & 3);
    const int blk = (int)(r >> 2);
    const int wmi = wave % m.WM, wni = wave / m.WM;
    const int tap = tile % m.T, nb = (tile / m.T) % m.NB, ma = tile / (m.T * m.NB);
    const int ca = (blk / nblk_b) * m.CA + (wmi * m.MA + ma) * 16 + 4 * (lane >> 4);
    const int cb = (blk % nblk_b) * m.CB + (wni * m.NB + nb) * 16 + (lane & 15);
    if (cb >= Cb || ca >= Ca) continue;
    const float4* p = reinterpret_cast<const float4*>(slab) + q;
    float4 s = *p;
    for (int k = 1; k < ksplit; ++k) {
      const float4 v = *(p + (size_t)k * (size_t)(slab_part / 4));
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ca + j < Ca) {
        float* o = dw + ((size_t)(ca + j) * Cb + cb) * m.T + tap;
        *o = accumulate ? *o + sv[j] : sv[j];
      }
  }
}

// The same sums through LDS, one workgroup per (channel block, wave, ma, nb) group of T tap tiles: the slab hands a thread element
// (m, n) of every tap tile -- weight elements T floats apart per lane and Cb * T apart per register, i.e. one 4-byte write per 32-byte
// sector in the kernel above (0.28 ms per training step for 157 MB of gradients: 8x write amplification) -- while the weight layout is
// contiguous over (cb, tap) for a fixed ca: 16 runs of 16 * T floats per group.  Parts are added in part order as above: same bits.
template <int T>
__global__ void __launch_bounds__(256) wgrad_finalize_tiled(const float* __restrict__ slab, float* __restrict__ dw, SlabMap m, int Ca, int Cb, int nblk_b,
                                                            int ksplit, long long slab_part, int accumulate) {
  __shared__ float tile[256 * T + 16];                    // [m][n][T], rows of 16 * T + 1 floats
  constexpr int ROW = 16 * T + 1;
  unsigned g = blockIdx.x;
  const int nb = (int)(g % m.NB); g /= m.NB;
  const int ma = (int)(g % m.MA); g /= m.MA;
  const int wave = (int)(g & 3);
  const int blk = (int)(g >> 2);
  const int wmi = wave % m.WM, wni = wave / m.WM;
  const int ca0 = (blk / nblk_b) * m.CA + (wmi * m.MA + ma) * 16;
  const int cb0 = (blk % nblk_b) * m.CB + (wni * m.NB + nb) * 16;
  if (ca0 >= Ca || cb0 >= Cb) return;                     // a group of padding channels only
  const int lane = threadIdx.x >> 2, reg = threadIdx.x & 3;
  const int mm = 4 * (lane >> 4) + reg, nn = lane & 15;
  const float* p = slab + (((size_t)blk * 4 + wave) * m.TILES + (size_t)(ma * m.NB + nb) * T) * 256 + threadIdx.x;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float s = p[(size_t)t * 256];
    for (int k = 1; k < ksplit; ++k) s += p[(size_t)k * (size_t)slab_part + (size_t)t * 256];
    tile[mm * ROW + nn * T + t] = s;
  }
  __syncthreads();
  const int ncb = min(16, Cb - cb0);                      // real `b` channels of the group: a run of ncb * T floats per `a` channel
#pragma unroll 1
  for (int row = 0; row < 16; ++row) {
    if (ca0 + row >= Ca) break;
    float* o = dw + ((size_t)(ca0 + row) * Cb + cb0) * T;
    for (int r = threadIdx.x; r < ncb * T; r += 256) o[r] = accumulate ? o[r] + tile[row * ROW + r] : tile[row * ROW + r];
  }
}

// [planes][H][W] -> [planes][H][Wp] (Wp = W rounded up to 4, zeros behind the row): rows of odd / narrow maps become 16-byte aligned
__global__ void __launch_bounds__(256) pad_width(const float* __restrict__ in, float* __restrict__ out, long long rows, int W, int Wp,
                                                 int C, int ctot, int c0, int H) {
  const long long total = rows * Wp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long long row = i / Wp;                       // (n * C + c) * H + y
    const int y = (int)(row % H);
    const long long pc = row / H;
    const int c = (int)(pc % C);
    const long long n = pc / C;
    out[i] = x < W ? in[(((size_t)n * ctot + c0 + c) * H + y) * W + x] : 0.f;
  }
}

template <class K>
static int launch(const Args& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<K>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad<K>), dim3(8 * ((a.total + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, st, a);
  return check_launch("conv_wgrad");
}

struct Variant {
  int ks, s, xt, r, db;
  int ca, cb, tiles;
  SlabMap map;
  int (*fn)(const Args&, hipStream_t);
};

// (KS, S, MA, NB, WM, WN, XT, R, DB)
#define FN2_WG_CLASS(X, KS, S, MA, NB, WM, WN) \
  X(KS, S, MA, NB, WM, WN, 8, 4, 1) X(KS, S, MA, NB, WM, WN, 16, 2, 1) X(KS, S, MA, NB, WM, WN, 28, 1, 1) \
  X(KS, S, MA, NB, WM, WN, 8, 4, 0) X(KS, S, MA, NB, WM, WN, 16, 2, 0) X(KS, S, MA, NB, WM, WN, 28, 1, 0)
#define FN2_WG_LIST(X) \
  FN2_WG_CLASS(X, 1, 1, 2, 4, 1, 4) FN2_WG_CLASS(X, 3, 1, 2, 2, 2, 2) FN2_WG_CLASS(X, 3, 2, 2, 2, 2, 2) \
  FN2_WG_CLASS(X, 4, 2, 2, 1, 2, 2) FN2_WG_CLASS(X, 5, 2, 2, 1, 2, 2) \
  X(3, 1, 2, 2, 2, 2, 56, 1, 1) X(3, 1, 2, 2, 2, 2, 56, 1, 0) \
  /* chunks of 5 rows x 8 pixels: one whole 5x7 sample (conv6_1, deconv5: the 4-row chunks spend 8 row slots on 5 rows) */ \
  X(3, 1, 2, 2, 2, 2, 8, 5, 1) X(4, 2, 2, 1, 2, 2, 8, 5, 1)        /* (3x3 / 2: its 11-row `b` window does not fit twice into the LDS) */
#define FN2_WG_ROW(KS, S, MA, NB, WM, WN, XT, R, DB) \
  {KS, S, XT, R, DB, Cfg<KS, S, MA, NB, WM, WN, XT, R, DB>::CA, Cfg<KS, S, MA, NB, WM, WN, XT, R, DB>::CB, Cfg<KS, S, MA, NB, WM, WN, XT, R, DB>::TILES, \
   slab_map<Cfg<KS, S, MA, NB, WM, WN, XT, R, DB>>(), &launch<Cfg<KS, S, MA, NB, WM, WN, XT, R, DB>>},
static const Variant kVariants[] = {FN2_WG_LIST(FN2_WG_ROW)};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int g_forced_db = -1;          // fn2_debug_set_wgrad_buffers: 0 / 1 forces the one- / two-buffer kernels (same bits either way)
int g_forced_xt = 0;           // fn2_debug_set_wgrad_chunk: forces the chunk width where a kernel of that width exists and covers the row rule

inline int round4(int v) { return (v + 3) / 4 * 4; }

// chunk width for a (padded) row width: whole small rows, 28-pixel pieces of wide ones (56 for the 3x3 / 1 class when it divides better)
static int pick_variant(int KS, int S, int Wa, int Ha) {
  int xt = Wa <= 8 ? 8 : Wa <= 16 ? 16 : 28;
  // rows per chunk: the class default (4 / 2 / 1 for 8 / 16 / 28+ pixels), or 5 where the map has 5 rows (a sample = one chunk).  The
  // accumulation order is (n, y, x) whatever the chunk shape: the padding slots add exact zeros
  const int want_r = (xt == 8 && Ha % 5 == 0 && g_forced_xt == 0) ? 5 : 0;
  if (KS == 3 && S == 1 && Wa >= 56 && cdiv_c(Wa, 56) * 56 <= cdiv_c(Wa, 28) * 28) xt = 56;
  if (g_forced_xt > 0 && (g_forced_xt >= 28 || g_forced_xt >= Wa)) xt = g_forced_xt;     // (chunks of several rows need whole rows)
  const int db = g_forced_db >= 0 ? g_forced_db : 1;      // two buffers: measured faster on every FlowNetC layer (and the 5x5 one-buffer kernel spills)
  if (want_r)
    for (int i = 0; i < kNumVariants; ++i)
      if (kVariants[i].ks == KS && kVariants[i].s == S && kVariants[i].xt == xt && kVariants[i].db == db && kVariants[i].r == want_r) return i;
  for (int i = 0; i < kNumVariants; ++i)
    if (kVariants[i].ks == KS && kVariants[i].s == S && kVariants[i].xt == xt && kVariants[i].db == db && kVariants[i].r != 5) return i;
  return -1;
}

struct Plan {
  int variant, Wap, Wbp, ksplit, nblk_a, nblk_b;
  size_t slab_floats, apad_floats, bpad_floats;
};

static bool make_plan(Plan& p, int N, int Ca, int Ha, int Wa, int Cb, int Hb, int Wb, int KS, int S, int pad) {
  if (N <= 0 || Ca <= 0 || Ha <= 0 || Wa <= 0 || Cb <= 0 || Hb <= 0 || Wb <= 0) return false;
  if (pad < 0 || pad > 4 || pad > KS - 1) return false;
  p.Wap = round4(Wa); p.Wbp = round4(Wb);
  p.variant = pick_variant(KS, S, p.Wap, Ha);
  if (p.variant < 0) return false;
  const Variant& v = kVariants[p.variant];
  if ((long long)v.ca * Ha * p.Wap >= (1ll << 28) || (long long)v.cb * Hb * p.Wbp >= (1ll << 28)) return false;   // descriptor range
  if ((long long)N * Ha >= (1ll << 30)) return false;
  p.ksplit = ksplit_for(N, Ca, Ha, Wa, Cb, KS);
  p.nblk_a = cdiv_c(Ca, v.ca); p.nblk_b = cdiv_c(Cb, v.cb);
  p.slab_floats = (size_t)p.nblk_a * p.nblk_b * 4 * v.tiles * 256;
  p.apad_floats = p.Wap != Wa ? (size_t)N * Ca * Ha * p.Wap : 0;
  p.bpad_floats = p.Wbp != Wb ? (size_t)N * Cb * Hb * p.Wbp : 0;
  return true;
}

}  // namespace wg
}  // namespace fn2

using namespace fn2;

FN2_API int fn2_debug_set_wgrad_buffers(int db) { wg::g_forced_db = db; return FN2_OK; }
FN2_API int fn2_debug_set_wgrad_chunk(int xt) { wg::g_forced_xt = xt; return FN2_OK; }

FN2_API int fn2_conv_wgrad_supported(int N, int Ca, int Ha, int Wa, int Cb, int Hb, int Wb, int kernel, int stride, int pad) {
  wg::Plan p;
  return wg::make_plan(p, N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad) ? 1 : 0;
}

FN2_API int fn2_conv_wgrad_ksplit(int N, int Ca, int Ha, int Wa, int Cb, int Hb, int Wb, int kernel, int stride, int pad) {
  wg::Plan p;
  return wg::make_plan(p, N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad) ? p.ksplit : 0;
}

FN2_API size_t fn2_conv_wgrad_workspace_bytes(int N, int Ca, int Ha, int Wa, int Cb, int Hb, int Wb, int kernel, int stride, int pad) {
  wg::Plan p;
  if (!wg::make_plan(p, N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad)) return 0;
  return sizeof(float) * ((size_t)p.ksplit * p.slab_floats + p.apad_floats + p.bpad_floats);
}

FN2_API int fn2_conv_wgrad(const float* a, const float* b, float* dw,
                           int N, int Ca, int Ha, int Wa, int a_channels, int a_c0,
                           int Cb, int Hb, int Wb, int b_channels, int b_c0,
                           int kernel, int stride, int pad, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "conv_wgrad: bad batch");
  if (!a || !b || !dw) return fail(FN2_ERR_INVALID_ARG, "conv_wgrad: null blob");
  if (a_c0 < 0 || a_c0 + Ca > a_channels || b_c0 < 0 || b_c0 + Cb > b_channels) return fail(FN2_ERR_INVALID_ARG, "conv_wgrad: channel slice outside the blob");
  hipStream_t st = as_stream(stream);
  if (N == 0) {
    if (!accumulate) (void)hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Ca * Cb * kernel * kernel, st);
    return FN2_OK;
  }
  // the taps must reach every pixel of `a`: S * (Ha - 1) + kernel - pad <= Hb + pad is NOT required (rows beyond b read zeros)
  wg::Plan p;
  if (!wg::make_plan(p, N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad))
    return fail(FN2_ERR_UNSUPPORTED, "conv_wgrad: unsupported geometry (a [%d,%d,%d,%d], b [.,%d,%d,%d], k %d s %d p %d)", N, Ca, Ha, Wa, Cb, Hb, Wb, kernel, stride, pad);
  const size_t need = sizeof(float) * ((size_t)p.ksplit * p.slab_floats + p.apad_floats + p.bpad_floats);
  if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "conv_wgrad: workspace of %zu bytes needed", need);
  if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "conv_wgrad: blobs and workspace must be 16-byte aligned");
  const wg::Variant& v = wg::kVariants[p.variant];
  float* slab = static_cast<float*>(workspace);
  float* apad = slab + (size_t)p.ksplit * p.slab_floats;
  float* bpad = apad + p.apad_floats;
  wg::Args g{};
  g.a = a; g.b = b; g.slab = slab;
  g.N = N; g.Ca = Ca; g.Ha = Ha; g.Wa = Wa; g.a_ctot = a_channels; g.a_c0 = a_c0;
  g.Cb = Cb; g.Hb = Hb; g.Wb = Wb; g.b_ctot = b_channels; g.b_c0 = b_c0;
  if (p.apad_floats) {
    const long long rows = (long long)N * Ca * Ha;
    hipLaunchKernelGGL(wg::pad_width, dim3(blocks_for(rows * p.Wap, 256, 4096)), dim3(256), 0, st, a, apad, rows, Wa, p.Wap, Ca, a_channels, a_c0, Ha);
    g.a = apad; g.Wa = p.Wap; g.a_ctot = Ca; g.a_c0 = 0;
  }
  if (p.bpad_floats) {
    const long long rows = (long long)N * Cb * Hb;
    hipLaunchKernelGGL(wg::pad_width, dim3(blocks_for(rows * p.Wbp, 256, 4096)), dim3(256), 0, st, b, bpad, rows, Wb, p.Wbp, Cb, b_channels, b_c0, Hb);
    g.b = bpad; g.Wb = p.Wbp; g.b_ctot = Cb; g.b_c0 = 0;
  }
  g.pad = pad; g.ksplit = p.ksplit; g.nblk_a = p.nblk_a; g.nblk_b = p.nblk_b;
  g.nxb = wg::cdiv_c(g.Wa, v.xt); g.U = N * Ha;
  g.total = (unsigned)((long long)p.nblk_a * p.nblk_b * p.ksplit);
  g.slab_part = (long long)p.slab_floats;
  const int rc = v.fn(g, st);
  if (rc != FN2_OK) return rc;
  const long long quads = (long long)p.slab_floats / 4;
  const long long groups = (long long)p.nblk_a * p.nblk_b * 4 * v.map.MA * v.map.NB;
  static const bool tiled = [] { const char* e = getenv("FN2_WGRAD_FINALIZE"); return !(e && e[0] == 'g'); }();      // "gather": the per-element kernel
  static const long long g_tiled_min = [] { const char* e = getenv("FN2_WGRAD_TILED_MIN"); return e ? atoll(e) : 1024ll; }();
  // (a group is T * 256 floats summed over ksplit parts by one workgroup: only where that still leaves >= 4 workgroups per CU (measured: 512 loses on conv4 / conv3_1, 1024 wins on deconv4) -- the layers
  // with megabytes of weights and few parts; conv2's 64 parts x 32 groups are the per-element kernel's)
  if (tiled && groups >= g_tiled_min && groups < (1ll << 31) && (v.map.T == 1 || v.map.T == 9 || v.map.T == 16 || v.map.T == 25)) {
    const dim3 grid((unsigned)groups);
    switch (v.map.T) {
      case 1: hipLaunchKernelGGL((wg::wgrad_finalize_tiled<1>), grid, dim3(256), 0, st, slab, dw, v.map, Ca, Cb, p.nblk_b, p.ksplit, g.slab_part, accumulate ? 1 : 0); break;
      case 9: hipLaunchKernelGGL((wg::wgrad_finalize_tiled<9>), grid, dim3(256), 0, st, slab, dw, v.map, Ca, Cb, p.nblk_b, p.ksplit, g.slab_part, accumulate ? 1 : 0); break;
      case 16: hipLaunchKernelGGL((wg::wgrad_finalize_tiled<16>), grid, dim3(256), 0, st, slab, dw, v.map, Ca, Cb, p.nblk_b, p.ksplit, g.slab_part, accumulate ? 1 : 0); break;
      default: hipLaunchKernelGGL((wg::wgrad_finalize_tiled<25>), grid, dim3(256), 0, st, slab, dw, v.map, Ca, Cb, p.nblk_b, p.ksplit, g.slab_part, accumulate ? 1 : 0); break;
    }
    return check_launch("conv_wgrad_finalize");
  }
  hipLaunchKernelGGL(wg::wgrad_finalize, dim3(blocks_for(quads, 256, 8192)), dim3(256), 0, st, slab, dw, v.map, Ca, Cb, p.nblk_b, quads, p.ksplit,
                     g.slab_part, accumulate ? 1 : 0);
  return check_launch("conv_wgrad_finalize");
}
