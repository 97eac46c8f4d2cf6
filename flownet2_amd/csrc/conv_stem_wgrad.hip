// Weight gradient of the 7x7 / stride 2 / pad 3 stem convolution (conv1 of every FlowNet: 3, 6 or 12 bottom channels -> 64) on
// v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chains), NCHW in, Caffe weight layout out, deterministic.
//
//     dw[co][ci][ky][kx] (+)= sum_{n, y, x}  top_diff[n][co][y][x] * bottom[n][ci][2 y + ky - 3][2 x + kx - 3]        (zero outside bottom)
//
// Reference: ConvolutionLayer::Backward_gpu -> weight_gpu_gemm (src/caffe/layers/conv_layer.cu:40-52, base_conv_layer.cpp:368-384: per
// SAMPLE im2col_gpu + cublasSgemm(top_diff x col^T) accumulated into weight_diff with beta = 1).
//
// Why its own kernel: csrc/conv_wgrad.hip tiles (top channels) x (BOTTOM CHANNELS) with one accumulator tile per tap -- 3 bottom channels
// fill 3 / 16 of a tile and 49 taps are 49 tiles.  Here the N axis of the GEMM is the TAP axis: M = 16 top channels, N = 16 consecutive
// taps t = (ci * 7 + ky) * 7 + kx (147 taps = 10 tiles at 3 channels: 92 % filled), K = pixels, 4 consecutive x of one row per k-step.
//   * wave w of a workgroup owns top channels 16 w .. 16 w + 15 and ALL tap tiles: per k-step one `top_diff` operand and NT `bottom`
//     operands (one ds_read_b32 at lane base + immediate each: lane (tap, k) reads window[ci][2 r + ky][2 (4 xq + k) + kx + 1]) for NT MFMAs;
//   * a workgroup walks a contiguous range of UNITS (sample, pair of output rows, 32-pixel x segment), unit by unit: the unit's top_diff tile
//     [64][2 rows x 32 px, padded to 68] and the bottom window [ci][9 rows][80 columns] arrive by 16-byte LDS-DMA straight from NCHW (rows /
//     columns outside the maps are out of range for the buffer descriptor: 0.0f = the zero padding; the 68-dword channel stride keeps the 16
//     channels of an operand read on 16 different banks), two buffers, one barrier per unit;
//   * every workgroup ("part") writes its accumulator tiles as they are; stem_wgrad_finalize adds the parts in part order into the weight
//     layout.  Summation order (restated by the oracle twin fn2_conv_k7s2_wgrad_cpu): per part one fma chain over the part's pixels in
//     (unit, row, x) order; the parts are added in 16 contiguous segments (part by part inside a segment, then the segment sums in order); the
//     number of parts is a function of the geometry only (fn2_conv_k7s2_wgrad_ksplit).
#include "fn2_common.hpp"

namespace fn2 {
namespace sw {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;
using lds_vf = const volatile __attribute__((address_space(3))) float*;

constexpr int kCout = 64, kR = 2, kXT = 32;
constexpr int kDS = kR * kXT + 4;                 // top_diff channel stride in LDS (dwords): 64 data + 4 padding
constexpr int kDSlots = kCout * (kDS / 4);        // 16-byte slots of the top_diff tile: 64 x 17
constexpr int kWC = 2 * kXT + 16;                 // window columns: bottom x = 2 x0 - 4 .. 2 x0 + 75 (70 are read; 80 == 16 (mod 64 banks): the tap
                                                  // rows ky, ky + 1, ky + 2 of one operand read sit on disjoint banks)
constexpr int kWR = 2 * (kR - 1) + 7;             // window rows: bottom y = 2 y0 - 3 .. 2 y0 + 5
constexpr unsigned kOOB = 0x7ffffff0u;

template <int CIN> struct Geo {
  static constexpr int TAPS = CIN * 49, NT = (TAPS + 15) / 16;
  static constexpr int CSW = kWR * kWC;                                  // window channel stride (dwords)
  static constexpr int WSlots = CIN * kWR * (kWC / 4);
  static constexpr int D_RUNS = (kDSlots + 63) / 64, W_RUNS = (WSlots + 63) / 64;
  static constexpr int D_DW = D_RUNS * 256, W_DW = W_RUNS * 256;         // whole 1 KiB runs
  static constexpr int BUF = D_DW + W_DW;
  static constexpr int RPW_D = (D_RUNS + 3) / 4, RPW_W = (W_RUNS + 3) / 4;
};

struct Args {
  const float* d; const float* b; float* slab;
  int N, H, W, Ho, Wo, nyb, nsx, units, parts;
};

__device__ __forceinline__ void unit_decode(const Args& a, int u, int& n, int& y0, int& x0) {
  const int sx = u % a.nsx; u /= a.nsx;
  const int yb = u % a.nyb;
  n = u / a.nyb; y0 = kR * yb; x0 = kXT * sx;
}

// per-lane plan of the LDS-DMA runs this wave issues for every unit (the same slots every time; only the unit's origin moves)
template <int CIN> struct Plan {
  unsigned d_off[Geo<CIN>::RPW_D];     // dword offset of the slot inside the sample's top_diff block relative to (y0, x0); ~0u: padding slot
  unsigned d_rx[Geo<CIN>::RPW_D];      // r | dx << 8
  unsigned w_off[Geo<CIN>::RPW_W];     // ci * H * W; ~0u: no slot
  unsigned w_rc[Geo<CIN>::RPW_W];      // window row | window column << 8
};

template <int CIN>
__device__ __forceinline__ void make_plan(const Args& a, int wave, int lane, Plan<CIN>& p) {
  using G = Geo<CIN>;
  const unsigned planeD = (unsigned)(a.Ho * a.Wo), planeB = (unsigned)(a.H * a.W);
#pragma unroll
  for (int i = 0; i < G::RPW_D; ++i) {
    const int s = (i * 4 + wave) * 64 + lane, co = s / (kDS / 4), q = s % (kDS / 4);      // slot q of channel co: q = 8 r + x / 4, q == 16: padding
    const int r = q >> 3, dx = 4 * (q & 7);
    const bool ok = i * 4 + wave < G::D_RUNS && s < kDSlots && q < 16;
    p.d_off[i] = ok ? (unsigned)co * planeD + (unsigned)(r * a.Wo + dx) : ~0u;
    p.d_rx[i] = (unsigned)r | ((unsigned)dx << 8);
  }
#pragma unroll
  for (int i = 0; i < G::RPW_W; ++i) {
    const int s = (i * 4 + wave) * 64 + lane, ci = s / (kWR * (kWC / 4)), rem = s % (kWR * (kWC / 4));
    const bool ok = i * 4 + wave < G::W_RUNS && s < G::WSlots;
    p.w_off[i] = ok ? (unsigned)ci * planeB : ~0u;
    p.w_rc[i] = (unsigned)(rem / (kWC / 4)) | ((unsigned)(4 * (rem % (kWC / 4))) << 8);
  }
}

template <int CIN>
__device__ __forceinline__ void stage_unit(const Args& a, int u, unsigned dst, int wave, const Plan<CIN>& p) {
  using G = Geo<CIN>;
  int n, y0, x0;
  unit_decode(a, u, n, y0, x0);
  const size_t planeD = (size_t)a.Ho * a.Wo, planeB = (size_t)a.H * a.W;
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d + (size_t)n * kCout * planeD), 0,
                                                                        (unsigned)(4u * kCout * planeD), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b + (size_t)n * CIN * planeB), 0,
                                                                        (unsigned)(4u * CIN * planeB), 0x00020000);
  const unsigned originD = (unsigned)(y0 * a.Wo + x0);
#pragma unroll
  for (int i = 0; i < G::RPW_D; ++i) {
    if (i * 4 + wave < G::D_RUNS) {
      const int r = (int)(p.d_rx[i] & 0xffu), dx = (int)(p.d_rx[i] >> 8);
      const bool ok = p.d_off[i] != ~0u && y0 + r < a.Ho && x0 + dx < a.Wo;
      const unsigned voff = ok ? 4u * (p.d_off[i] + originD) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_ptr_t)(uintptr_t)(dst + 1024u * (unsigned)(i * 4 + wave)), 16, voff, 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < G::RPW_W; ++i) {
    if (i * 4 + wave < G::W_RUNS) {
      const int gy = 2 * y0 - 3 + (int)(p.w_rc[i] & 0xffu), gx = 2 * x0 - 4 + (int)(p.w_rc[i] >> 8);
      const bool ok = p.w_off[i] != ~0u && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const unsigned voff = ok ? 4u * (p.w_off[i] + (unsigned)(gy * a.W + gx)) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(uintptr_t)(dst + 4u * G::D_DW + 1024u * (unsigned)(i * 4 + wave)), 16, voff, 0, 0, 0);
    }
  }
}

// BIAS (round 6): the A operand IS top_diff -- every lane also sums the values it feeds to the matrix pipe (one v_add per k-step), the four
// pixel lanes of a channel are combined at the end and the part's 64 bias sums travel in the first padding tap column of its slab:
// backward_gpu_bias (base_conv_layer.cpp:389-393) without a pass of its own over the blob.  Order: per lane its pixels in (unit, row, x)
// order, then lanes k = 0 .. 3 as ((k0 + k1) + (k2 + k3)), then the parts like the weights.
template <int CIN, bool BIAS>
__global__ void __launch_bounds__(256) stem_wgrad(Args a) {
  using G = Geo<CIN>;
  constexpr int NT = G::NT, NSTEP = kR * (kXT / 4);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = blockIdx.x;
  const int u0 = (int)((long long)part * a.units / a.parts), u1 = (int)((long long)(part + 1) * a.units / a.parts);
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  Plan<CIN> plan;
  make_plan<CIN>(a, wave, lane, plan);

  // operand lane bases (LDS byte addresses within a buffer): A = top_diff, lane (channel m, pixel k); B = bottom, lane (tap n, pixel k)
  const int m = lane & 15, k = lane >> 4;
  const unsigned a_base = 4u * (unsigned)((16 * wave + m) * kDS + k);
  unsigned b_base[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int t = 16 * nt + m;
    if (t >= G::TAPS) t = G::TAPS - 1;                         // padding taps read any valid address; their columns are never stored
    const int ci = t / 49, ky = (t % 49) / 7, kx = t % 7;
    b_base[nt] = 4u * (unsigned)(G::D_DW + ci * G::CSW + ky * kWC + kx + 1 + 2 * k);
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  if (u0 < u1) stage_unit<CIN>(a, u0, lds_base, wave, plan);
  for (int u = u0; u < u1; ++u) {
    const int buf = (u - u0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of unit u has landed ...
    __builtin_amdgcn_s_barrier();                             // ... and everybody's; everybody is done reading the other buffer
    if (u + 1 < u1) stage_unit<CIN>(a, u + 1, lds_base + 4u * (unsigned)((buf ^ 1) * G::BUF), wave, plan);
    const unsigned ab = lds_base + 4u * (unsigned)(buf * G::BUF) + a_base;
    const unsigned bb = lds_base + 4u * (unsigned)(buf * G::BUF);
    // k-steps st = 8 r + xq.  Software pipeline: the operands of step st + 1 are read one by one between the MFMAs of step st (volatile
    // LDS reads stay single ds_read_b32 at lane base + immediate, in program order), so every read has a whole step of MFMAs to land.
    auto a_at = [&](int st) { return ((lds_vf)(uintptr_t)ab)[(st / (kXT / 4)) * kXT + 4 * (st % (kXT / 4))]; };
    auto b_at = [&](int nt, int st) { return ((lds_vf)(uintptr_t)(bb + b_base[nt]))[2 * (st / (kXT / 4)) * kWC + 8 * (st % (kXT / 4))]; };
    float av[2], bv[2][NT];
    av[0] = a_at(0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[0][nt] = b_at(nt, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int cur = st & 1, nxt = cur ^ 1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (st + 1 < NSTEP) {
          if (nt == 0) av[nxt] = a_at(st + 1);
          bv[nxt][nt] = b_at(nt, st + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur], bv[cur][nt], acc[nt], 0, 0, 0);
        if (st + 1 < NSTEP) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (BIAS) bsum += av[cur];
    }
  }
  // slab[part][co][NT * 16]: lane (row block rb = lane >> 4, tap column lane & 15) holds rows 4 rb .. 4 rb + 3 of every tile
  float* out = a.slab + (size_t)part * kCout * (G::NT * 16);
  static_assert(G::TAPS < G::NT * 16, "a padding tap column carries the bias sums");
#pragma unroll
  for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (!(BIAS && 16 * nt + (lane & 15) == G::TAPS)) out[(size_t)(16 * wave + 4 * (lane >> 4) + r) * (G::NT * 16) + 16 * nt + (lane & 15)] = acc[nt][r];
  if constexpr (BIAS) {
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 16) out[(size_t)(16 * wave + lane) * (G::NT * 16) + G::TAPS] = bsum;
  }
}

// dw[co][t] (+)= sum over the parts, in a fixed two-level order: the parts are cut into kSeg contiguous segments [s P / kSeg, (s + 1) P / kSeg);
// a segment is summed part by part, the segment sums are added in segment order (empty segments skipped).  Workgroup = 16 consecutive
// elements (co, t) x 16 segments: every thread walks P / 16 parts (64-byte coalesced rows) instead of one thread walking all 768 -- the
// one-level form of the first version spent ~100 us of latency on 148 waves.
constexpr int kSeg = 16;
__global__ void __launch_bounds__(256) stem_wgrad_finalize(const float* __restrict__ slab, float* __restrict__ dw, int taps, int ntw, int parts, int accumulate) {
  __shared__ float seg_sum[kSeg][16];
  const int e = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;
  const bool live = i < kCout * taps;
  const int co = live ? i / taps : 0, t = live ? i % taps : 0;
  const float* p = slab + (size_t)co * ntw + t;
  const size_t stride = (size_t)kCout * ntw;
  const int k0 = (int)((long long)sg * parts / kSeg), k1 = (int)((long long)(sg + 1) * parts / kSeg);
  float s = 0.f;
  if (live && k0 < k1) {
    s = p[(size_t)k0 * stride];
    for (int k = k0 + 1; k < k1; ++k) s += p[(size_t)k * stride];
  }
  seg_sum[sg][e] = s;
  __syncthreads();
  if (sg == 0 && live) {
    float tot = 0.f;
    bool first = true;
    for (int q = 0; q < kSeg; ++q) {
      const int q0 = (int)((long long)q * parts / kSeg), q1 = (int)((long long)(q + 1) * parts / kSeg);
      if (q0 >= q1) continue;
      tot = first ? seg_sum[q][e] : tot + seg_sum[q][e];
      first = false;
    }
    dw[i] = accumulate ? dw[i] + tot : tot;
  }
}

static bool geometry_ok(int N, int Cin, int H, int W, int Cout) {
  return N > 0 && (Cin == 3 || Cin == 6 || Cin == 12) && Cout == kCout && H >= 1 && W >= 8 && W % 8 == 0 &&
         (long long)N * Cin * H * W < (1ll << 28) && (long long)N * Cout * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) < (1ll << 28);
}

static void fill(Args& a, int N, int H, int W) {
  a.N = N; a.H = H; a.W = W;
  a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.nyb = (a.Ho + kR - 1) / kR; a.nsx = (a.Wo + kXT - 1) / kXT;
  a.units = N * a.nyb * a.nsx;
  // parts: three workgroups per CU when there is that much work, at least 4 units each; a function of the geometry only
  const int n1 = order_batch(N) * a.nyb * a.nsx;
  int parts = 768;
  if (parts > n1 / 4) parts = n1 / 4;
  if (parts < 1) parts = 1;
  a.parts = parts > a.units ? a.units : parts;
}

template <int CIN>
static int launch(Args a, float* dw, float* db, int accumulate, hipStream_t st) {
  using G = Geo<CIN>;
  constexpr size_t lds = sizeof(float) * 2 * G::BUF;
  if (db) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad<CIN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return fail(FN2_ERR_UNSUPPORTED, "conv_k7s2_wgrad: %zu bytes of dynamic LDS refused by the runtime", lds);
    }
    hipLaunchKernelGGL((stem_wgrad<CIN, true>), dim3((unsigned)a.parts), dim3(256), lds, st, a);
    hipLaunchKernelGGL(stem_wgrad_finalize, dim3((kCout * G::TAPS + 15) / 16), dim3(256), 0, st, a.slab, dw, G::TAPS, G::NT * 16, a.parts, accumulate);
    // the bias sums: the same two-level order over the parts, "one tap" wide, from the padding column
    hipLaunchKernelGGL(stem_wgrad_finalize, dim3((kCout + 15) / 16), dim3(256), 0, st, a.slab + G::TAPS, db, 1, G::NT * 16, a.parts, accumulate);
    return check_launch("conv_k7s2_wgrad (+ bias)");
  }
  // 70 / 104 KB of dynamic LDS (above the 64 KB default): set per launch -- the attribute belongs to the CURRENT device's copy of the
  // function, a per-process latch would leave every device after the first without it -- and checked
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad<CIN, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return fail(FN2_ERR_UNSUPPORTED, "conv_k7s2_wgrad: %zu bytes of dynamic LDS refused by the runtime", lds);
  }
  hipLaunchKernelGGL((stem_wgrad<CIN, false>), dim3((unsigned)a.parts), dim3(256), lds, st, a);
  hipLaunchKernelGGL(stem_wgrad_finalize, dim3((kCout * G::TAPS + 15) / 16), dim3(256), 0, st, a.slab, dw, G::TAPS, G::NT * 16, a.parts, accumulate);
  return check_launch("conv_k7s2_wgrad");
}

}  // namespace sw
}  // namespace fn2

using namespace fn2;

FN2_API int fn2_conv_k7s2_wgrad_supported(int N, int Cin, int Hin, int Win, int Cout) {
  return sw::geometry_ok(N, Cin, Hin, Win, Cout) ? 1 : 0;
}

FN2_API int fn2_conv_k7s2_wgrad_ksplit(int N, int Cin, int Hin, int Win, int Cout) {
  if (!sw::geometry_ok(N, Cin, Hin, Win, Cout)) return 0;
  sw::Args a{};
  sw::fill(a, N, Hin, Win);
  return a.parts;
}

FN2_API size_t fn2_conv_k7s2_wgrad_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout) {
  if (!sw::geometry_ok(N, Cin, Hin, Win, Cout)) return 0;
  sw::Args a{};
  sw::fill(a, N, Hin, Win);
  return sizeof(float) * (size_t)a.parts * sw::kCout * (((Cin * 49 + 15) / 16) * 16);
}

namespace fn2 {
int conv_k7s2_wgrad_bias(const float* top_diff, const float* bottom, float* weight_diff, float* bias_diff, int N, int Cin, int Hin, int Win, int Cout,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream);
}

FN2_API int fn2_conv_k7s2_wgrad(const float* top_diff, const float* bottom, float* weight_diff, int N, int Cin, int Hin, int Win, int Cout,
                                int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  return fn2::conv_k7s2_wgrad_bias(top_diff, bottom, weight_diff, nullptr, N, Cin, Hin, Win, Cout, accumulate, workspace, workspace_bytes, stream);
}

// bias_diff != NULL: the bias gradient of the same layer comes out of the same pass (stem_wgrad<.., true>)
int fn2::conv_k7s2_wgrad_bias(const float* top_diff, const float* bottom, float* weight_diff, float* bias_diff, int N, int Cin, int Hin, int Win, int Cout,
                              int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (N < 0) return fail(FN2_ERR_INVALID_ARG, "conv_k7s2_wgrad: bad batch");
  if (!top_diff || !bottom || !weight_diff) return fail(FN2_ERR_INVALID_ARG, "conv_k7s2_wgrad: null blob");
  if (N == 0) {
    if (!accumulate) (void)hipMemsetAsync(weight_diff, 0, sizeof(float) * (size_t)Cout * Cin * 49, as_stream(stream));
    if (!accumulate && bias_diff) (void)hipMemsetAsync(bias_diff, 0, sizeof(float) * (size_t)Cout, as_stream(stream));
    return FN2_OK;
  }
  if (!sw::geometry_ok(N, Cin, Hin, Win, Cout))
    return fail(FN2_ERR_UNSUPPORTED, "conv_k7s2_wgrad: needs Cin in {3,6,12}, Cout == 64, width %% 8 == 0 (got Cin %d, Cout %d, %dx%d)", Cin, Cout, Hin, Win);
  if (((reinterpret_cast<uintptr_t>(top_diff) | reinterpret_cast<uintptr_t>(bottom)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "conv_k7s2_wgrad: blobs must be 16-byte aligned");
  const size_t need = fn2_conv_k7s2_wgrad_workspace_bytes(N, Cin, Hin, Win, Cout);
  if (!workspace || workspace_bytes < need) return fail(FN2_ERR_WORKSPACE, "conv_k7s2_wgrad: workspace of %zu bytes needed", need);
  sw::Args a{};
  sw::fill(a, N, Hin, Win);
  a.d = top_diff; a.b = bottom; a.slab = static_cast<float*>(workspace);
  hipStream_t st = as_stream(stream);
  if (Cin == 3) return sw::launch<3>(a, weight_diff, bias_diff, accumulate, st);
  if (Cin == 6) return sw::launch<6>(a, weight_diff, bias_diff, accumulate, st);
  return sw::launch<12>(a, weight_diff, bias_diff, accumulate, st);
}
