// Convolution / Deconvolution behind ONE geometry descriptor: the library picks its own kernel.
//
// The reference has one ConvolutionLayer / DeconvolutionLayer for every geometry (im2col + GEMM per sample: conv_layer.cu:8-23,
// deconv_layer.cu:8-26, base_conv_layer.cpp:255-396); this library has a kernel family per geometry class (direct 5x5/2 .. 7x7/2 and 1x1,
// Winograd F(2x2,3x3), the small-map kernel with its deterministic K split, the Deconvolution as GEMM + col2im or as parity classes).
// Which one serves a layer is decided HERE, from the descriptor alone -- the Python mirror (flownet2_amd/functional.py) and the Caffe
// adapter (csrc/caffe_adapter) ask the same function, so a Caffe user of libflownet2_hip.so gets the routing the benchmarks ran with.
// The thresholds are measurements: profiles/r02_conv_bench_*.txt, r04_conv_plane_bench_flownetc.txt, scripts/probes/small_layer_routes.py.
#include "fn2_common.hpp"

namespace {

struct Out { int H, W; };
inline Out conv_out(const fn2_conv_desc* d) {
  return {(d->Hin + 2 * d->pad - d->kernel) / d->stride + 1, (d->Win + 2 * d->pad - d->kernel) / d->stride + 1};
}

bool valid(const fn2_conv_desc* d) {
  return d && d->N >= 1 && d->Cin >= 1 && d->Cout >= 1 && d->Hin >= 1 && d->Win >= 1 && d->kernel >= 1 && d->stride >= 1 && d->pad >= 0 &&
         d->Hin + 2 * d->pad >= d->kernel && d->Win + 2 * d->pad >= d->kernel;
}

}  // namespace

FN2_API int fn2_conv_route(const fn2_conv_desc* d, int flags) {
  if (!valid(d)) return FN2_CONV_ROUTE_NONE;
  const bool force = (flags & FN2_ROUTE_FORCE) != 0;
  const int k = d->kernel, s = d->stride, p = d->pad, Cin = d->Cin, Cout = d->Cout, H = d->Hin, W = d->Win;
  // batch-invariant mode (fn2_set_batch_invariant): the route -- and with it the arithmetic -- must not depend on the batch: decide as for
  // one sample, and require the small-map kernel's plan for one sample AND for the real batch
  const bool inv = fn2_get_batch_invariant() != 0;
  const int N = inv ? 1 : d->N;
  const bool wino_first = force || inv;
  const Out o = conv_out(d);
  // the two geometry classes with kernels of their own (round 6: reachable by descriptor, so that the Caffe adapter's Convolution plug-in
  // serves EVERY layer of the FlowNet graphs): the 7x7 / 2 stem on 3 / 6 / 12 input channels, and the 2-channel predict_flow heads
  // (the 12-channel stems of FlowNet2's stacked nets are whole channel quads: the direct kernel does them in one pass -- 174 us against 277 for
  // the register-resident stem kernel's two passes at batch 4 @768x384 -- so the stem route is for the channel counts the direct kernel does not take)
  if (k == 7 && s == 2 && p == 3 && !(Cin % 4 == 0 && fn2_conv_mfma_supported(Cin, H, W, Cout, k, s, p)) && fn2_conv_k7s2_relu_supported(Cin, H, W, Cout))
    return FN2_CONV_ROUTE_STEM;
  if (k == 3 && s == 1 && p == 1 && Cout == 2) return FN2_CONV_ROUTE_HEAD;
  const bool wino_ok = k == 3 && s == 1 && fn2_conv_wino_supported(Cin, H, W, Cout, p) != 0;
  // accumulator blocks of the Winograd kernel (16 channels x an 8x8-pixel block of tiles): from ~1000 on the launch fills the 1024 SIMDs and
  // it is the fastest kernel of a 3x3 / 1 layer (20x28 maps win by 1.6x, 12x24 maps lose)
  if (wino_ok && (wino_first || (long long)N * ((o.H + 7) / 8) * ((o.W + 7) / 8) * (Cout / 16) >= 1000)) return FN2_CONV_ROUTE_WINOGRAD;
  const int maxpix = 8000;
  if (k == 3 && (force || o.H * o.W <= maxpix) && fn2_conv_plane_supported(N, Cin, H, W, Cout, s, p) &&
      (N == d->N || fn2_conv_plane_supported(d->N, Cin, H, W, Cout, s, p)))
    return FN2_CONV_ROUTE_PLANE;       // the encoder layers from 1/16 resolution down: whole planes in LDS, pixels of several samples per tile, split K
  if (k == 5 && s == 2 && p == 2 && (long long)N * ((o.H + 3) / 4) * ((o.W + 3) / 4) * (Cout / 16) < 64 * 256 && o.H * o.W <= maxpix &&
      fn2_conv_plane_k_supported(d->N, Cin, H, W, Cout, 5, 2, 2) && (N == d->N || fn2_conv_plane_k_supported(N, Cin, H, W, Cout, 5, 2, 2)))
    return FN2_CONV_ROUTE_PLANE;       // conv3 of the encoders when one sample is the whole batch: the direct kernel has no K split to fill the chip with
  if (wino_ok) return FN2_CONV_ROUTE_WINOGRAD;      // too large for the small-map kernel, too small to fill the chip: still 2.25x fewer multiplies
  if (fn2_conv_mfma_supported(Cin, H, W, Cout, k, s, p)) return FN2_CONV_ROUTE_DIRECT;
  return FN2_CONV_ROUTE_NONE;
}

FN2_API size_t fn2_conv_packed_weight_floats(const fn2_conv_desc* d, int route) {
  if (!valid(d)) return 0;
  if (route == FN2_CONV_ROUTE_WINOGRAD) return fn2_conv_wino_packed_floats(d->Cout, d->Cin);
  if (route == FN2_CONV_ROUTE_DIRECT || route == FN2_CONV_ROUTE_PLANE) return fn2_conv_mfma_packed_floats(d->Cout, d->Cin, d->kernel);
  if (route == FN2_CONV_ROUTE_STEM || route == FN2_CONV_ROUTE_HEAD) return (size_t)d->Cout * d->Cin * d->kernel * d->kernel;    // these kernels read the blob as it is
  return 0;
}

FN2_API int fn2_conv_pack_weights(const fn2_conv_desc* d, int route, const float* weight, float* packed, void* stream) {
  if (!valid(d) || !weight || !packed) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_pack_weights: bad descriptor or NULL blob");
  if (route == FN2_CONV_ROUTE_WINOGRAD) return fn2_conv_wino_pack_weights(weight, packed, d->Cout, d->Cin, stream);
  if (route == FN2_CONV_ROUTE_DIRECT || route == FN2_CONV_ROUTE_PLANE) return fn2_conv_mfma_pack_weights(weight, packed, d->Cout, d->Cin, d->kernel, stream);
  if (route == FN2_CONV_ROUTE_STEM || route == FN2_CONV_ROUTE_HEAD) {
    if (weight != packed &&
        hipMemcpyAsync(packed, weight, sizeof(float) * (size_t)d->Cout * d->Cin * d->kernel * d->kernel, hipMemcpyDeviceToDevice, fn2::as_stream(stream)) != hipSuccess)
      return fn2::fail(FN2_ERR_LAUNCH, "conv_pack_weights: device copy of the weight blob failed");
    return FN2_OK;
  }
  return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_pack_weights: no own kernel for this layer (route %d)", route);
}

FN2_API size_t fn2_conv_workspace_bytes(const fn2_conv_desc* d, int route) {
  if (valid(d) && route == FN2_CONV_ROUTE_HEAD) return fn2_predict_flow_conv_workspace_bytes(d->N, d->Cin, d->Hin, d->Win);
  if (!valid(d) || route != FN2_CONV_ROUTE_PLANE) return 0;
  return d->kernel == 3 ? fn2_conv_plane_workspace_bytes(d->N, d->Cin, d->Hin, d->Win, d->Cout, d->stride, d->pad)
                        : fn2_conv_plane_k_workspace_bytes(d->N, d->Cin, d->Hin, d->Win, d->Cout, d->kernel, d->stride, d->pad);
}

FN2_API int fn2_conv_forward(const fn2_conv_desc* d, int route, const float* bottom, int bottom_channels, int bottom_c0,
                             const float* packed_weight, const float* bias, float* top, int top_channels, int top_c0,
                             int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream) {
  if (!valid(d)) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_forward: bad descriptor");
  switch (route) {
    case FN2_CONV_ROUTE_WINOGRAD:
      return fn2_conv_wino_forward(bottom, packed_weight, bias, top, d->N, d->Cin, d->Hin, d->Win, bottom_channels, bottom_c0, d->Cout, top_channels,
                                   top_c0, d->pad, relu, negative_slope, stream);
    case FN2_CONV_ROUTE_PLANE:
      if (d->kernel == 3)
        return fn2_conv_plane_forward(bottom, packed_weight, bias, top, d->N, d->Cin, d->Hin, d->Win, bottom_channels, bottom_c0, d->Cout,
                                      top_channels, top_c0, d->stride, d->pad, relu, negative_slope, workspace, workspace_bytes, stream);
      return fn2_conv_plane_k_forward(bottom, packed_weight, bias, top, d->N, d->Cin, d->Hin, d->Win, bottom_channels, bottom_c0, d->Cout,
                                      top_channels, top_c0, d->kernel, d->stride, d->pad, relu, negative_slope, workspace, workspace_bytes, stream);
    case FN2_CONV_ROUTE_DIRECT:
      return fn2_conv_mfma_forward(bottom, packed_weight, bias, top, d->N, d->Cin, d->Hin, d->Win, bottom_channels, bottom_c0, d->Cout, top_channels,
                                   top_c0, d->kernel, d->stride, d->pad, relu, negative_slope, stream);
    case FN2_CONV_ROUTE_STEM:
      if (bottom_channels != d->Cin || bottom_c0 != 0 || top_channels != d->Cout || top_c0 != 0)
        return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_forward: the stem kernel reads and writes whole blobs, not channel slices");
      // the kernel always applies t > 0 ? t : t * slope: slope 1 is the identity (exactly: t * 1.0f == t) for a layer without a fused ReLU
      return fn2_conv_k7s2_relu_forward(bottom, packed_weight, bias, top, d->N, d->Cin, d->Hin, d->Win, d->Cout, relu ? negative_slope : 1.0f, stream);
    case FN2_CONV_ROUTE_HEAD: {
      if (bottom_channels != d->Cin || bottom_c0 != 0 || top_channels != d->Cout || top_c0 != 0)
        return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_forward: the flow-head kernel reads and writes whole blobs, not channel slices");
      const int rc = fn2_predict_flow_conv_forward(bottom, packed_weight, bias, top, d->N, d->Cin, d->Hin, d->Win, workspace, workspace_bytes, stream);
      if (rc != FN2_OK || !relu) return rc;
      return fn2_bias_leaky_relu_forward(top, nullptr, d->N, d->Cout, d->Hin, d->Win, negative_slope, stream);
    }
    default:
      return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_forward: no own kernel for Convolution{kernel %d, stride %d, pad %d} %d -> %d on %d x %d",
                       d->kernel, d->stride, d->pad, d->Cin, d->Cout, d->Hin, d->Win);
  }
}

// ---- Deconvolution{4, 2, 1}: Cin = bottom channels, Cout = top channels, top is [N, Cout, 2 Hin, 2 Win]; weight blob [Cin][Cout][4][4] ----
FN2_API int fn2_deconv_route(const fn2_conv_desc* d, int flags) {
  (void)flags;
  if (!valid(d) || d->kernel != 4 || d->stride != 2 || d->pad != 1) return FN2_DECONV_ROUTE_NONE;
  if (d->Cin == 2 && d->Cout == 2) return FN2_DECONV_ROUTE_HEAD;       // upsample_flow*: the 2-channel kernel (csrc/flow_head.hip)
  const int M = d->Cout * 16;
  // weight^T x bottom as the 1x1 / GEMM form of the direct kernel + our col2im / bias / ReLU pass: 5-25 % faster than the parity-class kernel
  // on every FlowNet map it takes (profiles/r02_deconv_bench_flownetc.txt); planes whose size is no multiple of 4 (deconv5: 5x7) are not its
  const bool gemm_ok = M % 32 == 0 && (d->Hin * d->Win) % 4 == 0 && fn2_conv_mfma_supported(d->Cin, d->Hin, d->Win, M, 1, 1, 0) != 0;
  if (gemm_ok) return FN2_DECONV_ROUTE_GEMM;
  if (fn2_deconv_plane_supported(d->N, d->Cin, d->Hin, d->Win, d->Cout)) return FN2_DECONV_ROUTE_PLANE;
  return FN2_DECONV_ROUTE_NONE;
}

FN2_API size_t fn2_deconv_packed_weight_floats(const fn2_conv_desc* d, int route) {
  if (!valid(d)) return 0;
  if (route == FN2_DECONV_ROUTE_GEMM) return fn2_conv_mfma_packed_floats(d->Cout * 16, d->Cin, 1);
  if (route == FN2_DECONV_ROUTE_PLANE) return fn2_deconv_plane_packed_floats(d->Cin, d->Cout);
  if (route == FN2_DECONV_ROUTE_HEAD) return (size_t)d->Cin * d->Cout * 16;
  return 0;
}

FN2_API int fn2_deconv_pack_weights(const fn2_conv_desc* d, int route, const float* weight, float* packed, void* stream) {
  if (!valid(d) || !weight || !packed) return fn2::fail(FN2_ERR_INVALID_ARG, "deconv_pack_weights: bad descriptor or NULL blob");
  const int M = d->Cout * 16;
  // GEMM operand [M = Cout 16][Cin] straight from the [Cin][Cout][4][4] blob through the strided view (base_conv_layer.cpp:375-384's weight^T)
  if (route == FN2_DECONV_ROUTE_GEMM) return fn2_conv_mfma_pack_weights_view(weight, packed, M, d->Cin, 1, M, d->Cin, 1, M, 0, stream);
  if (route == FN2_DECONV_ROUTE_PLANE) return fn2_deconv_plane_pack_weights(weight, packed, d->Cin, d->Cout, stream);
  if (route == FN2_DECONV_ROUTE_HEAD) {
    if (weight != packed && hipMemcpyAsync(packed, weight, sizeof(float) * (size_t)d->Cin * d->Cout * 16, hipMemcpyDeviceToDevice, fn2::as_stream(stream)) != hipSuccess)
      return fn2::fail(FN2_ERR_LAUNCH, "deconv_pack_weights: device copy of the weight blob failed");
    return FN2_OK;
  }
  return fn2::fail(FN2_ERR_UNSUPPORTED, "deconv_pack_weights: no own kernel for this layer (route %d)", route);
}

FN2_API size_t fn2_deconv_workspace_bytes(const fn2_conv_desc* d, int route) {
  if (!valid(d)) return 0;
  if (route == FN2_DECONV_ROUTE_GEMM) return sizeof(float) * (size_t)d->N * d->Cout * 16 * d->Hin * d->Win;      // the column matrix
  if (route == FN2_DECONV_ROUTE_PLANE) return fn2_deconv_plane_workspace_bytes(d->N, d->Cin, d->Hin, d->Win, d->Cout);
  return 0;
}

FN2_API int fn2_deconv_forward(const fn2_conv_desc* d, int route, const float* bottom, int bottom_channels, int bottom_c0,
                               const float* packed_weight, const float* bias, float* top, int top_channels, int top_c0,
                               int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream) {
  if (!valid(d)) return fn2::fail(FN2_ERR_INVALID_ARG, "deconv_forward: bad descriptor");
  if (route == FN2_DECONV_ROUTE_GEMM) {
    if (!workspace || workspace_bytes < fn2_deconv_workspace_bytes(d, route))
      return fn2::fail(FN2_ERR_INVALID_ARG, "deconv_forward: workspace of %zu bytes needed for the column matrix", fn2_deconv_workspace_bytes(d, route));
    float* col = static_cast<float*>(workspace);
    int rc = fn2_conv_mfma_forward(bottom, packed_weight, nullptr, col, d->N, d->Cin, d->Hin, d->Win, bottom_channels, bottom_c0, d->Cout * 16,
                                   d->Cout * 16, 0, 1, 1, 0, 0, 0.f, stream);
    if (rc) return rc;
    return fn2_col2im_bias_relu_forward_into(col, bias, top, d->N, d->Cout, 2 * d->Hin, 2 * d->Win, 4, 1, 2, relu, negative_slope, top_channels, top_c0, stream);
  }
  if (route == FN2_DECONV_ROUTE_PLANE)
    return fn2_deconv_plane_forward(bottom, packed_weight, bias, top, d->N, d->Cin, d->Hin, d->Win, bottom_channels, bottom_c0, d->Cout, top_channels,
                                    top_c0, relu, negative_slope, workspace, workspace_bytes, stream);
  if (route == FN2_DECONV_ROUTE_HEAD) {
    if (bottom_channels != 2 || bottom_c0 != 0) return fn2::fail(FN2_ERR_UNSUPPORTED, "deconv_forward: the flow-head kernel reads a whole 2-channel blob");
    if (relu) return fn2::fail(FN2_ERR_UNSUPPORTED, "deconv_forward: the 2-channel flow-head kernel has no fused ReLU (the FlowNet graphs have none there)");
    return fn2_upsample_flow_deconv_forward_into(bottom, packed_weight, bias, top, d->N, d->Hin, d->Win, top_channels, top_c0, stream);
  }
  return fn2::fail(FN2_ERR_UNSUPPORTED, "deconv_forward: no own kernel for Deconvolution{kernel %d, stride %d, pad %d} %d -> %d on %d x %d",
                   d->kernel, d->stride, d->pad, d->Cin, d->Cout, d->Hin, d->Win);
}

// =====================================================================================================================================
// Backward by descriptor (declarations and reference citations: include/flownet2_hip.h).  The same decisions flownet2_amd/functional.py
// made in Python through round 4 -- now one function for the autograd mirror and the Caffe adapter's Backward_gpu.
namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Geometry of a data gradient: top_diff [N, Ct, Ht, Wt] -> bottom_diff [N, Cb, Hb, Wb]; the kernels compute Cp >= Cb channels.
struct Bwd { int route, Ct, Ht, Wt, Cb, Hb, Wb, Cp; };

Bwd bwd_geom(const fn2_conv_desc* d, int transposed) {
  Bwd g{FN2_BWD_ROUTE_NONE, 0, 0, 0, 0, 0, 0, 0};
  if (!valid(d)) return g;
  const int k = d->kernel, s = d->stride, p = d->pad, N = d->N;
  g.Cb = d->Cin; g.Hb = d->Hin; g.Wb = d->Win; g.Ct = d->Cout; g.Cp = d->Cin;
  if (transposed) {
    // Deconvolution{4, 2, 1}: the gradient is the 4x4 / 2 / 1 CONVOLUTION of top_diff with the blob as it is ([Cin][Cout][4][4] = [out][in][k][k])
    if (k != 4 || s != 2 || p != 1 || d->Cout % 4 != 0) return g;
    g.Ht = 2 * d->Hin; g.Wt = 2 * d->Win; g.Cp = round_up(d->Cin, 64);
    if (fn2_conv_mfma_supported(g.Ct, g.Ht, g.Wt, g.Cp, 4, 2, 1)) g.route = FN2_BWD_ROUTE_DIRECT;
    else if (d->Cout % 8 == 0 && fn2_conv_plane_k_supported(N, g.Ct, g.Ht, g.Wt, g.Cp, 4, 2, 1)) g.route = FN2_BWD_ROUTE_PLANE;   // deconv5: 10x14 -> 5x7
    return g;
  }
  const Out o = conv_out(d);
  g.Ht = o.H; g.Wt = o.W;
  if (s == 2 && ((k == 5 && p == 2) || (k == 3 && p == 1)) && d->Cin % 64 == 0) {
    if (fn2_tconv_supported(g.Ct, g.Ht, g.Wt, g.Cb, g.Hb, g.Wb, k, p)) { g.route = FN2_BWD_ROUTE_TCONV; return g; }
    // maps whose width is no multiple of 4 (conv5, conv6: top_diff 10x14 / 5x7): the transposed 3x3 / 2 / 1 convolution IS the Deconvolution{4, 2, 1}
    // whose fourth tap row and column are zero (Y = 2 y - 1 + ky in both)
    if (k == 3 && g.Hb == 2 * g.Ht && g.Wb == 2 * g.Wt && fn2_deconv_plane_supported(N, g.Ct, g.Ht, g.Wt, g.Cb)) { g.route = FN2_BWD_ROUTE_DECONV_PLANE; return g; }
  }
  if (k == 1 && s == 1 && p == 0) {
    g.Cp = round_up(d->Cin, 32);
    if (fn2_conv_mfma_supported(g.Ct, g.Ht, g.Wt, g.Cp, 1, 1, 0)) g.route = FN2_BWD_ROUTE_DIRECT;
    return g;
  }
  if (k == 3 && s == 1 && p == 1) {
    g.Cp = round_up(d->Cin, 16);
    if (fn2_conv_wino_supported(g.Ct, g.Ht, g.Wt, g.Cp, 1)) { g.route = FN2_BWD_ROUTE_WINOGRAD; return g; }
    g.Cp = round_up(d->Cin, 64);       // 10x14, 5x7 (conv5_1, conv6_1): the small-map kernel on the same rotated weights
    if (d->Cout % 8 == 0 && fn2_conv_plane_supported(N, g.Ct, g.Ht, g.Wt, g.Cp, 1, 1)) g.route = FN2_BWD_ROUTE_PLANE;
  }
  return g;
}

size_t bwd_kernel_workspace(const fn2_conv_desc* d, int transposed, const Bwd& g) {
  if (g.route == FN2_BWD_ROUTE_PLANE)
    return transposed ? fn2_conv_plane_k_workspace_bytes(d->N, g.Ct, g.Ht, g.Wt, g.Cp, 4, 2, 1) : fn2_conv_plane_workspace_bytes(d->N, g.Ct, g.Ht, g.Wt, g.Cp, 1, 1);
  if (g.route == FN2_BWD_ROUTE_DECONV_PLANE) return fn2_deconv_plane_workspace_bytes(d->N, g.Ct, g.Ht, g.Wt, g.Cb);
  return 0;
}

// [Cout][Cin][3][3] -> [Cp][Cout][3][3]: rotated by 180 degrees, channel axes swapped, rows beyond Cin zero (the blob the Winograd packing reads)
__global__ void __launch_bounds__(256) rot180_swap(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int Cp) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)Cp * Cout * 9;
  if (i >= total) return;
  const int t = (int)(i % 9);
  const long long r = i / 9;
  const int co = (int)(r % Cout), ci = (int)(r / Cout);
  out[i] = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + (8 - t)] : 0.f;
}

}  // namespace

FN2_API int fn2_conv_backward_data_route(const fn2_conv_desc* d, int transposed) { return bwd_geom(d, transposed).route; }

FN2_API size_t fn2_conv_backward_data_packed_weight_floats(const fn2_conv_desc* d, int transposed, int route) {
  const Bwd g = bwd_geom(d, transposed);
  if (route == FN2_BWD_ROUTE_NONE || route != g.route) return 0;
  switch (route) {
    case FN2_BWD_ROUTE_WINOGRAD: return fn2_conv_wino_packed_floats(g.Cp, g.Ct);
    case FN2_BWD_ROUTE_TCONV: return fn2_conv_mfma_packed_floats(g.Cb, g.Ct, d->kernel);
    case FN2_BWD_ROUTE_DECONV_PLANE: return fn2_deconv_plane_packed_floats(g.Ct, g.Cb);
    default: return fn2_conv_mfma_packed_floats(g.Cp, g.Ct, d->kernel);       // PLANE, DIRECT
  }
}

FN2_API size_t fn2_conv_backward_data_pack_workspace_bytes(const fn2_conv_desc* d, int transposed, int route) {
  const Bwd g = bwd_geom(d, transposed);
  return (route == FN2_BWD_ROUTE_WINOGRAD && route == g.route) ? sizeof(float) * (size_t)g.Cp * g.Ct * 9 : 0;
}

FN2_API int fn2_conv_backward_data_pack_weights(const fn2_conv_desc* d, int transposed, int route, const float* weight, float* packed,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  const Bwd g = bwd_geom(d, transposed);
  if (!weight || !packed) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_backward_data_pack_weights: NULL blob");
  if (route == FN2_BWD_ROUTE_NONE || route != g.route)
    return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_backward_data_pack_weights: route %d is not this layer's (%d)", route, g.route);
  const int k = d->kernel, Cin = d->Cin, Cout = d->Cout;
  switch (route) {
    case FN2_BWD_ROUTE_WINOGRAD: {
      const size_t need = fn2_conv_backward_data_pack_workspace_bytes(d, transposed, route);
      if (!workspace || workspace_bytes < need) return fn2::fail(FN2_ERR_WORKSPACE, "conv_backward_data_pack_weights: %zu bytes of scratch needed for the rotated blob", need);
      const long long total = (long long)g.Cp * Cout * 9;
      hipLaunchKernelGGL(rot180_swap, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), weight,
                         static_cast<float*>(workspace), Cout, Cin, g.Cp);
      const int rc = fn2::check_launch("conv_backward_data_pack_weights (rotate)");
      if (rc) return rc;
      return fn2_conv_wino_pack_weights(static_cast<const float*>(workspace), packed, g.Cp, Cout, stream);
    }
    case FN2_BWD_ROUTE_TCONV:          // operand [Cin][Cout][k][k]: the Convolution's own blob with its channel axes swapped
      return fn2_conv_mfma_pack_weights_view(weight, packed, Cin, Cout, k, Cin, Cout, (long long)k * k, (long long)Cin * k * k, 0, stream);
    case FN2_BWD_ROUTE_DECONV_PLANE:   // the [Cout][Cin][3][3] blob read as a Deconvolution blob [in = Cout][out = Cin] with zero taps
      return fn2_deconv_plane_pack_weights_k(weight, packed, Cout, Cin, 3, stream);
    case FN2_BWD_ROUTE_PLANE:
      if (transposed) return fn2_conv_mfma_pack_weights_view(weight, packed, g.Cp, Cout, 4, Cin, Cout, (long long)Cout * 16, 16, 0, stream);
      return fn2_conv_mfma_pack_weights_view(weight, packed, g.Cp, Cout, 3, Cin, Cout, 9, (long long)Cin * 9, 1, stream);
    default:                           // DIRECT
      if (transposed) return fn2_conv_mfma_pack_weights_view(weight, packed, g.Cp, Cout, 4, Cin, Cout, (long long)Cout * 16, 16, 0, stream);
      return fn2_conv_mfma_pack_weights_view(weight, packed, g.Cp, Cout, 1, Cin, Cout, 1, Cin, 0, stream);
  }
}

FN2_API size_t fn2_conv_backward_data_workspace_bytes(const fn2_conv_desc* d, int transposed, int route) {
  const Bwd g = bwd_geom(d, transposed);
  if (route == FN2_BWD_ROUTE_NONE || route != g.route) return 0;
  size_t b = align256(bwd_kernel_workspace(d, transposed, g));
  if (g.Cp != g.Cb) b += sizeof(float) * (size_t)d->N * g.Cp * g.Hb * g.Wb;       // the padded result, copied into bottom_diff afterwards
  return b;
}

// the same for a caller whose bottom_diff blob has room for `bottom_room` channels: with room for the computed channels
// (fn2_conv_backward_data_computed_channels) only the kernel's own scratch is needed -- no padded copy
FN2_API size_t fn2_conv_backward_data_workspace_bytes_with_room(const fn2_conv_desc* d, int transposed, int route, int bottom_room) {
  const Bwd g = bwd_geom(d, transposed);
  if (route == FN2_BWD_ROUTE_NONE || route != g.route) return 0;
  if (bottom_room >= g.Cp) return bwd_kernel_workspace(d, transposed, g);
  return fn2_conv_backward_data_workspace_bytes(d, transposed, route);
}

FN2_API int fn2_conv_backward_data_computed_channels(const fn2_conv_desc* d, int transposed, int route) {
  const Bwd g = bwd_geom(d, transposed);
  return (route == FN2_BWD_ROUTE_NONE || route != g.route) ? 0 : g.Cp;
}

namespace fn2 {
int tconv_forward_masked(const float* bottom, const float* packed_weight, const float* bias, float* top,
                         int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                         int Cout, int Hout, int Wout, int top_channels, int top_c0, int kernel, int pad,
                         int relu, float negative_slope, const float* mask, int mask_channels, int mask_c0, float mask_slope, void* stream);
}

FN2_API int fn2_conv_backward_data_masked_supported(const fn2_conv_desc* d, int transposed, int route) {
  const Bwd g = bwd_geom(d, transposed);
  return route == FN2_BWD_ROUTE_TCONV && route == g.route && g.Cp == g.Cb;
}

FN2_API int fn2_conv_backward_data_masked(const fn2_conv_desc* d, int transposed, int route, const float* top_diff, int top_channels, int top_c0,
                                          const float* packed, float* bottom_diff, int bottom_channels, int bottom_c0,
                                          const float* bottom_data, int data_channels, int data_c0, float negative_slope, void* stream) {
  if (!fn2_conv_backward_data_masked_supported(d, transposed, route))
    return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_backward_data_masked: only the transposed-convolution route folds the ReLU derivative of the layer in front");
  if (!top_diff || !packed || !bottom_diff || !bottom_data) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_backward_data_masked: NULL blob");
  const Bwd g = bwd_geom(d, transposed);
  if (top_c0 < 0 || top_c0 + g.Ct > top_channels || bottom_c0 < 0 || bottom_c0 + g.Cb > bottom_channels)
    return fn2::fail(FN2_ERR_INVALID_ARG, "conv_backward_data_masked: channel slice outside its blob");
  return fn2::tconv_forward_masked(top_diff, packed, nullptr, bottom_diff, d->N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cb, g.Hb, g.Wb, bottom_channels,
                                   bottom_c0, d->kernel, d->pad, 0, 0.f, bottom_data, data_channels, data_c0, negative_slope, stream);
}

FN2_API int fn2_conv_backward_data(const fn2_conv_desc* d, int transposed, int route, const float* top_diff, int top_channels, int top_c0,
                                   const float* packed, float* bottom_diff, int bottom_channels, int bottom_c0, int bottom_room,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  const Bwd g = bwd_geom(d, transposed);
  if (!top_diff || !packed || !bottom_diff) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_backward_data: NULL blob");
  if (route == FN2_BWD_ROUTE_NONE || route != g.route)
    return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_backward_data: no own kernel for %s{kernel %d, stride %d, pad %d} %d -> %d on %d x %d (route %d)",
                     transposed ? "Deconvolution" : "Convolution", d ? d->kernel : 0, d ? d->stride : 0, d ? d->pad : 0, d ? d->Cin : 0, d ? d->Cout : 0,
                     d ? d->Hin : 0, d ? d->Win : 0, route);
  if (top_c0 < 0 || top_c0 + g.Ct > top_channels || bottom_c0 < 0 || bottom_room < g.Cb || bottom_c0 + bottom_room > bottom_channels)
    return fn2::fail(FN2_ERR_INVALID_ARG, "conv_backward_data: channel slice outside its blob");
  const size_t kws = bwd_kernel_workspace(d, transposed, g);
  const bool padded = g.Cp > bottom_room;               // no room for the surplus channels: through the workspace
  const size_t need = padded ? fn2_conv_backward_data_workspace_bytes(d, transposed, route) : kws;
  if (need && (!workspace || workspace_bytes < need)) return fn2::fail(FN2_ERR_WORKSPACE, "conv_backward_data: workspace too small (%zu < %zu)", workspace_bytes, need);
  float* out = padded ? reinterpret_cast<float*>(static_cast<char*>(workspace) + align256(kws)) : bottom_diff;
  const int oc = padded ? g.Cp : bottom_channels, o0 = padded ? 0 : bottom_c0;
  const int N = d->N;
  int rc;
  switch (route) {
    case FN2_BWD_ROUTE_WINOGRAD:
      rc = fn2_conv_wino_forward(top_diff, packed, nullptr, out, N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cp, oc, o0, 1, 0, 0.f, stream);
      break;
    case FN2_BWD_ROUTE_TCONV:
      rc = fn2_tconv_forward(top_diff, packed, nullptr, out, N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cb, g.Hb, g.Wb, oc, o0, d->kernel, d->pad, 0, 0.f, stream);
      break;
    case FN2_BWD_ROUTE_DECONV_PLANE:
      rc = fn2_deconv_plane_forward(top_diff, packed, nullptr, out, N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cb, oc, o0, 0, 0.f, workspace, kws, stream);
      break;
    case FN2_BWD_ROUTE_PLANE:
      if (transposed)
        rc = fn2_conv_plane_k_forward(top_diff, packed, nullptr, out, N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cp, oc, o0, 4, 2, 1, 0, 0.f, workspace, kws, stream);
      else
        rc = fn2_conv_plane_forward(top_diff, packed, nullptr, out, N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cp, oc, o0, 1, 1, 0, 0.f, workspace, kws, stream);
      break;
    default:
      rc = transposed ? fn2_conv_mfma_forward(top_diff, packed, nullptr, out, N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cp, oc, o0, 4, 2, 1, 0, 0.f, stream)
                      : fn2_conv_mfma_forward(top_diff, packed, nullptr, out, N, g.Ct, g.Ht, g.Wt, top_channels, top_c0, g.Cp, oc, o0, 1, 1, 0, 0, 0.f, stream);
      break;
  }
  if (rc || !padded) return rc;
  // the first Cb of the Cp computed channels of every sample -> the layer's slice of bottom_diff (one strided copy)
  const size_t plane = sizeof(float) * (size_t)g.Hb * g.Wb;
  if (hipMemcpy2DAsync(bottom_diff + (size_t)bottom_c0 * g.Hb * g.Wb, (size_t)bottom_channels * plane, out, (size_t)g.Cp * plane, (size_t)g.Cb * plane, (size_t)N,
                       hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
    (void)hipGetLastError();
    return fn2::fail(FN2_ERR_LAUNCH, "conv_backward_data: copy of the padded result failed");
  }
  return FN2_OK;
}

// ---- weight gradient ----
namespace {
bool stem_class(const fn2_conv_desc* d, int transposed) {
  return !transposed && d->kernel == 7 && d->stride == 2 && d->pad == 3 && fn2_conv_k7s2_wgrad_supported(d->N, d->Cin, d->Hin, d->Win, d->Cout) != 0;
}
struct WG { int Ca, Ha, Wa, Cb, Hb, Wb; };
WG wg_geom(const fn2_conv_desc* d, int transposed) {          // `a`: the map at the convolution's OUTPUT resolution (conv_wgrad.hip)
  if (transposed) return {d->Cin, d->Hin, d->Win, d->Cout, 2 * d->Hin, 2 * d->Win};
  const Out o = conv_out(d);
  return {d->Cout, o.H, o.W, d->Cin, d->Hin, d->Win};
}
}  // namespace

FN2_API int fn2_conv_backward_weights_supported(const fn2_conv_desc* d, int transposed) {
  if (!valid(d)) return 0;
  if (transposed && (d->kernel != 4 || d->stride != 2 || d->pad != 1)) return 0;
  if (stem_class(d, transposed)) return 1;
  const WG w = wg_geom(d, transposed);
  if (w.Ca < 16 || w.Cb < 16) return 0;       // the 2-channel flow heads have kernels of their own (fn2_predict_flow_conv_backward, fn2_upsample_flow_deconv_backward)
  return fn2_conv_wgrad_supported(d->N, w.Ca, w.Ha, w.Wa, w.Cb, w.Hb, w.Wb, d->kernel, d->stride, d->pad) != 0;
}

FN2_API size_t fn2_conv_backward_weights_workspace_bytes(const fn2_conv_desc* d, int transposed) {
  if (!fn2_conv_backward_weights_supported(d, transposed)) return 0;
  if (stem_class(d, transposed)) return fn2_conv_k7s2_wgrad_workspace_bytes(d->N, d->Cin, d->Hin, d->Win, d->Cout);
  const WG w = wg_geom(d, transposed);
  return fn2_conv_wgrad_workspace_bytes(d->N, w.Ca, w.Ha, w.Wa, w.Cb, w.Hb, w.Wb, d->kernel, d->stride, d->pad);
}

namespace fn2 {
int conv_k7s2_wgrad_bias(const float* top_diff, const float* bottom, float* weight_diff, float* bias_diff, int N, int Cin, int Hin, int Win, int Cout,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream);
}

// weight_diff AND bias_diff of a layer from one pass over top_diff where a kernel has that form (the stem: csrc/conv_stem_wgrad.hip sums
// the operand it feeds to the matrix pipe); 1 = it has
FN2_API int fn2_conv_backward_weights_bias_fused(const fn2_conv_desc* d, int transposed) { return valid(d) && stem_class(d, transposed) ? 1 : 0; }

FN2_API int fn2_conv_backward_weights_bias(const fn2_conv_desc* d, int transposed, const float* bottom, const float* top_diff, float* weight_diff,
                                           float* bias_diff, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (!fn2_conv_backward_weights_bias_fused(d, transposed))
    return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_backward_weights_bias: this layer has no fused weight + bias gradient kernel");
  if (!bottom || !top_diff || !weight_diff || !bias_diff) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_backward_weights_bias: NULL blob");
  return fn2::conv_k7s2_wgrad_bias(top_diff, bottom, weight_diff, bias_diff, d->N, d->Cin, d->Hin, d->Win, d->Cout, accumulate, workspace, workspace_bytes, stream);
}

FN2_API int fn2_conv_backward_weights(const fn2_conv_desc* d, int transposed, const float* bottom, int bottom_channels, int bottom_c0,
                                      const float* top_diff, int top_channels, int top_c0, float* weight_diff, int accumulate,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!fn2_conv_backward_weights_supported(d, transposed))
    return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_backward_weights: no own kernel for this layer");
  if (!bottom || !top_diff || !weight_diff) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_backward_weights: NULL blob");
  if (stem_class(d, transposed)) {
    if (bottom_channels != d->Cin || bottom_c0 != 0 || top_channels != d->Cout || top_c0 != 0)
      return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_backward_weights: the stem kernel reads whole blobs, not channel slices");
    return fn2_conv_k7s2_wgrad(top_diff, bottom, weight_diff, d->N, d->Cin, d->Hin, d->Win, d->Cout, accumulate, workspace, workspace_bytes, stream);
  }
  const WG w = wg_geom(d, transposed);
  const float* a = transposed ? bottom : top_diff;
  const float* b = transposed ? top_diff : bottom;
  const int ac = transposed ? bottom_channels : top_channels, a0 = transposed ? bottom_c0 : top_c0;
  const int bc = transposed ? top_channels : bottom_channels, b0 = transposed ? top_c0 : bottom_c0;
  return fn2_conv_wgrad(a, b, weight_diff, d->N, w.Ca, w.Ha, w.Wa, ac, a0, w.Cb, w.Hb, w.Wb, bc, b0, d->kernel, d->stride, d->pad, accumulate,
                        workspace, workspace_bytes, stream);
}
