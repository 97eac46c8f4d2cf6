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
  return 0;
}

FN2_API int fn2_conv_pack_weights(const fn2_conv_desc* d, int route, const float* weight, float* packed, void* stream) {
  if (!valid(d) || !weight || !packed) return fn2::fail(FN2_ERR_INVALID_ARG, "conv_pack_weights: bad descriptor or NULL blob");
  if (route == FN2_CONV_ROUTE_WINOGRAD) return fn2_conv_wino_pack_weights(weight, packed, d->Cout, d->Cin, stream);
  if (route == FN2_CONV_ROUTE_DIRECT || route == FN2_CONV_ROUTE_PLANE) return fn2_conv_mfma_pack_weights(weight, packed, d->Cout, d->Cin, d->kernel, stream);
  return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_pack_weights: no own kernel for this layer (route %d)", route);
}

FN2_API size_t fn2_conv_workspace_bytes(const fn2_conv_desc* d, int route) {
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
    default:
      return fn2::fail(FN2_ERR_UNSUPPORTED, "conv_forward: no own kernel for Convolution{kernel %d, stride %d, pad %d} %d -> %d on %d x %d",
                       d->kernel, d->stride, d->pad, d->Cin, d->Cout, d->Hin, d->Win);
  }
}

// ---- Deconvolution{4, 2, 1}: Cin = bottom channels, Cout = top channels, top is [N, Cout, 2 Hin, 2 Win]; weight blob [Cin][Cout][4][4] ----
FN2_API int fn2_deconv_route(const fn2_conv_desc* d, int flags) {
  (void)flags;
  if (!valid(d) || d->kernel != 4 || d->stride != 2 || d->pad != 1) return FN2_DECONV_ROUTE_NONE;
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
  return 0;
}

FN2_API int fn2_deconv_pack_weights(const fn2_conv_desc* d, int route, const float* weight, float* packed, void* stream) {
  if (!valid(d) || !weight || !packed) return fn2::fail(FN2_ERR_INVALID_ARG, "deconv_pack_weights: bad descriptor or NULL blob");
  const int M = d->Cout * 16;
  // GEMM operand [M = Cout 16][Cin] straight from the [Cin][Cout][4][4] blob through the strided view (base_conv_layer.cpp:375-384's weight^T)
  if (route == FN2_DECONV_ROUTE_GEMM) return fn2_conv_mfma_pack_weights_view(weight, packed, M, d->Cin, 1, M, d->Cin, 1, M, 0, stream);
  if (route == FN2_DECONV_ROUTE_PLANE) return fn2_deconv_plane_pack_weights(weight, packed, d->Cin, d->Cout, stream);
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
  return fn2::fail(FN2_ERR_UNSUPPORTED, "deconv_forward: no own kernel for Deconvolution{kernel %d, stride %d, pad %d} %d -> %d on %d x %d",
                   d->kernel, d->stride, d->pad, d->Cin, d->Cout, d->Hin, d->Win);
}
