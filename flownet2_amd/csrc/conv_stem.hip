// Stem convolution of the FlowNet encoders: Convolution{kernel_size 7, stride 2, pad 3} + bias + ReLU{negative_slope}
// fused, direct (no im2col), on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).
//
// Reference: conv1 + ReLU1 of FlowNetC/S -- ConvolutionLayer::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23:
// per-sample im2col + SGEMM, then forward_gpu_bias, base_conv_layer.cpp:325-348) followed by the in-place ReLU layer
// (relu_layer.cu:8-27).  With 3 (or 6) input channels the contraction is only 147 (294) long: the library kernels run
// this layer at 36 TFLOP/s and the bias / activation passes re-read the 147 MB output twice.  Here
//   * GEMM view: M = output positions (16 consecutive x of one row), N = 16 output channels, K = (c, ky, kx);
//     a wave owns ONE N tile and keeps its whole weight slice in registers (14 VGPRs per input channel);
//   * K order (c, ky, kx padded to 8): k-step = (c, ky, half h), lane group kk <-> kx = 4h + kk, so every A operand
//     is one ds_read_b32 at  lane base + immediate;  the 8th tap has weight 0;
//   * the input rows of a workgroup (2 RB + 5 rows x CIN channels x (32 tiles + 8) columns) are staged once by 16-byte
//     LDS-DMA in natural pixel order; out-of-image rows / columns come back 0 from the buffer descriptor = zero padding;
//   * the MFMA result layout gives every lane 4 consecutive x of one output channel: bias + leaky ReLU + one 16-byte store;
//   * 12 input channels (the stems of FlowNet2's stacked nets) run as two 6-channel passes over the same output.
#include "fn2_common.hpp"

#include <type_traits>

namespace fn2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using lds_ptr_t = __attribute__((address_space(3))) void*;

struct StemArgs {
  int N, Hin, Win, Hout, Wout, Cout;
  int Ctot, c0;     // channels of the bottom blob / first channel of this pass (CIN channels per pass)
  int ntx;          // x blocks per output row
  int nyb;          // row blocks
  float slope;
};

// PASS: 0 = the whole contraction in one launch; 1 = first half of a two-pass contraction (raw partial sums are stored);
// 2 = second half (adds the stored partial sums, then bias and ReLU).  Two passes serve the 12-channel stems of FlowNet2's
// stacked FlowNetS nets: their weight slice (168 values per lane) does not fit the register file, 84 per pass do.
template <int CIN, int RB, int XT, int PASS>     // XT: 16-pixel tiles per workgroup row
__global__ void __launch_bounds__(256, 2)
conv_k7s2_relu(const float* __restrict__ in, const float* __restrict__ weight, const float* __restrict__ bias,
               float* __restrict__ out, StemArgs a) {
  constexpr int ROWS = 2 * RB + 5;               // staged input rows per channel
  constexpr int RS = 32 * XT + 8;                // staged columns per row (multiple of 4): input cols [2 x0 - 4, 2 x0 + 32 XT + 4)
  constexpr int CS = ROWS * RS;
  constexpr int SLOTS = CIN * ROWS * (RS / 4);   // 16-byte slots
  constexpr int NRUN = (SLOTS + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // N tile of this wave within the group of 4
  int b = blockIdx.x;
  const int bx = b % a.ntx; b /= a.ntx;
  const int by = b % a.nyb; b /= a.nyb;
  const int cog = b % (a.Cout / 64), n = b / (a.Cout / 64);
  const int x0 = bx * 16 * XT, y0 = by * RB;
  const int co0 = cog * 64 + wave * 16;

  // ---- stage the input window: slot s -> (c, row, group of 4 columns) ----
  const size_t plane = (size_t)a.Hin * a.Win;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in + ((size_t)n * a.Ctot + a.c0) * plane), 0, (unsigned)(4u * CIN * plane), 0x00020000);
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;
#pragma unroll
  for (int i = 0; i < (NRUN + 3) / 4; ++i) {
    const int run = i * 4 + wave;
    if (run < NRUN) {
      const int s = run * 64 + lane;
      unsigned voff = 0x7ffffff0u;
      if (s < SLOTS) {
        const int c = s / (ROWS * (RS / 4)), rem = s % (ROWS * (RS / 4));
        const int row = rem / (RS / 4), gq = rem % (RS / 4);
        const int yi = 2 * y0 - 3 + row, xi = 2 * x0 - 4 + 4 * gq;
        if (yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win) voff = 4u * (unsigned)(c * plane + (size_t)yi * a.Win + xi);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(uintptr_t)(lds_base + run * 1024u), 16, voff, 0, 0, 0);
    }
  }

  // ---- weights of this wave's 16 output channels: lane (kk, nn) holds W[co0 + nn][c][ky][4h + kk] ----
  const int kk = lane >> 4, nn = lane & 15;
  float w[CIN][7][2];
  {
    const float* wp = weight + ((size_t)(co0 + nn) * a.Ctot + a.c0) * 49;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        w[c][ky][0] = wp[(c * 7 + ky) * 7 + kk];
        w[c][ky][1] = kk < 3 ? wp[(c * 7 + ky) * 7 + 4 + kk] : 0.f;
      }
  }
  const float bv = bias ? bias[co0 + nn] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- A operand: lane (kk, m) reads column 32 t + 2 m + kx + 1 (kx = 4h + kk) of row 2 rb + ky ----
  const int abase = 2 * nn + kk + 1;
  const int xq = 4 * (lane >> 4);                  // this lane's 4 consecutive output pixels inside a tile
  auto tiles = [&](auto ntag, int rb, int y, int t) {          // NTL = 1 or 2 tiles in flight (independent accumulators)
    constexpr int NTL = decltype(ntag)::value;
    f32x4 acc[NTL];
#pragma unroll
    for (int u = 0; u < NTL; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* r0 = smem + (2 * rb) * RS + 32 * t + abase;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int u = 0; u < NTL; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(r0[c * CS + ky * RS + 4 * h + 32 * u], w[c][ky][h], acc[u], 0, 0, 0);
    float* orow = out + (((size_t)n * a.Cout + co0 + nn) * a.Hout + y) * a.Wout;
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
      const int x = x0 + 16 * (t + u) + xq;
      if (x < a.Wout) {                            // Wout % 4 == 0: a quad is inside or outside as a whole
        f32x4 v = acc[u];
        if constexpr (PASS == 2) v += *reinterpret_cast<const f32x4*>(orow + x);
        if constexpr (PASS != 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float s = v[j] + bv; v[j] = s > 0.f ? s : s * a.slope; }
        }
        *reinterpret_cast<f32x4*>(orow + x) = v;
      }
    }
  };
  for (int rb = 0; rb < RB; ++rb) {
    const int y = y0 + rb;
    if (y >= a.Hout) break;
    int t = 0;
#pragma unroll 1
    for (; t + 1 < XT; t += 2) {
      if (x0 + 16 * t >= a.Wout) break;            // ragged last x block
      tiles(std::integral_constant<int, 2>{}, rb, y, t);
    }
    if ((XT & 1) && x0 + 16 * t < a.Wout) tiles(std::integral_constant<int, 1>{}, rb, y, t);
  }
}

template <int CIN, int PASS>
static int launch_stem(const float* in, const float* weight, const float* bias, float* out, int N, int Ctot, int c0, int Hin, int Win,
                       int Cout, float slope, hipStream_t st) {
  constexpr int RB = 2, XT = 7;
  StemArgs a;
  a.N = N; a.Hin = Hin; a.Win = Win; a.Cout = Cout; a.slope = slope; a.Ctot = Ctot; a.c0 = c0;
  a.Hout = (Hin + 6 - 7) / 2 + 1; a.Wout = (Win + 6 - 7) / 2 + 1;
  a.ntx = (a.Wout + 16 * XT - 1) / (16 * XT);
  a.nyb = (a.Hout + RB - 1) / RB;
  const long long grid = (long long)N * (Cout / 64) * a.nyb * a.ntx;
  if (grid > 0x7fffffffll) return fail(FN2_ERR_UNSUPPORTED, "conv_k7s2_relu: grid too large");
  constexpr size_t lds = sizeof(float) * CIN * (2 * RB + 5) * (32 * XT + 8) + 1024;     // + the tail of the last (partial) run
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_k7s2_relu<CIN, RB, XT, PASS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_k7s2_relu<CIN, RB, XT, PASS>), dim3((unsigned)grid), dim3(256), lds, st, in, weight, bias, out, a);
  return check_launch("conv_k7s2_relu_forward");
}

}  // namespace fn2

using namespace fn2;

FN2_API int fn2_conv_k7s2_relu_supported(int Cin, int Hin, int Win, int Cout) {
  return (Cin == 3 || Cin == 6 || Cin == 12) && Cout % 64 == 0 && Cout > 0 && Hin >= 1 && Win >= 8 && Win % 8 == 0 &&
         (long long)Cin * Hin * Win < (1ll << 28);
}

FN2_API int fn2_conv_k7s2_relu_forward(const float* in, const float* weight, const float* bias, float* out,
                                       int N, int Cin, int Hin, int Win, int Cout, float negative_slope, void* stream) {
  if (N < 0 || Cin <= 0 || Hin <= 0 || Win <= 0 || Cout <= 0) return fail(FN2_ERR_INVALID_ARG, "conv_k7s2_relu: bad shape");
  if (N == 0) return FN2_OK;
  if (!in || !weight || !out) return fail(FN2_ERR_INVALID_ARG, "conv_k7s2_relu: null blob");
  if (!fn2_conv_k7s2_relu_supported(Cin, Hin, Win, Cout))
    return fail(FN2_ERR_UNSUPPORTED, "conv_k7s2_relu: needs Cin in {3,6,12}, Cout %% 64 == 0, width %% 8 == 0 (got Cin %d, Cout %d, W %d)", Cin, Cout, Win);
  if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) != 0)
    return fail(FN2_ERR_UNSUPPORTED, "conv_k7s2_relu: blobs must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (Cin == 3) return launch_stem<3, 0>(in, weight, bias, out, N, 3, 0, Hin, Win, Cout, negative_slope, st);
  if (Cin == 6) return launch_stem<6, 0>(in, weight, bias, out, N, 6, 0, Hin, Win, Cout, negative_slope, st);
  const int rc = launch_stem<6, 1>(in, weight, bias, out, N, 12, 0, Hin, Win, Cout, negative_slope, st);     // channels 0-5: partial sums
  if (rc) return rc;
  return launch_stem<6, 2>(in, weight, bias, out, N, 12, 6, Hin, Win, Cout, negative_slope, st);            // channels 6-11, bias, ReLU
}
