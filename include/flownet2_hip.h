/*
 * flownet2_hip.h -- C ABI of libflownet2_hip.so: the MI355X (gfx950) implementation of the
 * FlowNet2 hot-path operators of lmb-freiburg/flownet2 (a Caffe fork).
 *
 * Every entry point below replaces one Layer<Dtype>::Forward_gpu / Backward_gpu (Dtype = float) of
 * the reference; the reference interface it stands in for is cited as file:line relative to the
 * reference tree.  INTEGRATION.md shows the adapter (a Caffe Layer<float> subclass calling these
 * functions with Blob::gpu_data() pointers) that a maintainer of the reference would add.
 *
 * Conventions
 *  - All tensors are dense NCHW fp32 (Blob layout, include/caffe/blob.hpp:153-164), DEVICE pointers.
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is what the
 *    reference uses for every launch).
 *  - Functions never allocate, free or synchronise; scratch comes from the caller through
 *    (workspace, workspace_bytes) whose size the matching *_workspace_bytes() reports.
 *  - Return value: FN2_OK (0) or a negative fn2_status; fn2_last_error_string() returns a
 *    thread-local description of the last failure.  The reference aborts through glog CHECK /
 *    LOG(FATAL) in these situations (src/caffe/common.cpp:58); the adapter turns non-zero into
 *    LOG(FATAL), the Python host raises.
 *  - The library is stateless and re-entrant; it uses the current HIP device.
 *
 * The CPU oracle (oracle/fn2_oracle.c, test infrastructure only) exports the same functions with a
 * `_cpu` suffix, HOST pointers and no stream / workspace arguments.
 */
#ifndef FLOWNET2_HIP_H_
#define FLOWNET2_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FN2_VERSION_MAJOR 0
#define FN2_VERSION_MINOR 1

typedef enum fn2_status {
  FN2_OK = 0,
  FN2_ERR_INVALID_ARG = -1,   /* a CHECK in the reference's LayerSetUp/Reshape would have fired */
  FN2_ERR_UNSUPPORTED = -2,   /* configuration the reference rejects with LOG(FATAL)            */
  FN2_ERR_WORKSPACE = -3,     /* workspace NULL or too small                                     */
  FN2_ERR_LAUNCH = -4         /* hipGetLastError() after the launch was not hipSuccess           */
} fn2_status;

/* "maj.min (gfx950)" */
const char* fn2_version(void);
const char* fn2_last_error_string(void);

/* ------------------------------------------------------------------------------------------------
 * Correlation  (type: "Correlation")
 *   params  <- CorrelationParameter, src/caffe/proto/caffe.proto:628-644 (field for field)
 *   shapes  <- CorrelationLayer::Reshape, src/caffe/layers/correlation_layer.cpp:41-84
 *   forward <- CorrelationLayer::Forward_gpu, src/caffe/layers/correlation_layer.cu:431-504
 *              (blob_rearrange_kernel2 :23-42, CorrelateData :45-114, CorrelateDataSubtract :252-293)
 *   backward<- CorrelationLayer::Backward_gpu, src/caffe/layers/correlation_layer.cu:507-603
 *              (CorrelateDataBackward0/1 :117-249, ...Subtract :297-427)
 * ---------------------------------------------------------------------------------------------- */
enum { FN2_CORR_MULTIPLY = 0, FN2_CORR_SUBTRACT = 1 };   /* CorrelationParameter.CorrelationType */

typedef struct fn2_corr_params {
  int pad;               /* pad              = 2  [default 0]  */
  int kernel_size;       /* kernel_size      = 3  (odd, required) */
  int max_displacement;  /* max_displacement = 4  (required)   */
  int stride1;           /* stride_1         = 5  [default 1]  */
  int stride2;           /* stride_2         = 6  [default 1]  */
  int corr_type;         /* correlation_type = 15 [default MULTIPLY] */
  int do_abs;            /* do_abs = 7: parsed but never used by the reference (correlation_layer.cpp:29); ignored */
  int single_direction;  /* single_direction = 8 [default 0]: Correlation1D only (-1 left, 0 both, 1 right); the 2-D layer never reads it */
} fn2_corr_params;

/* Reshape: top = [N, topC, topH, topW].  Errors: even kernel_size, stride <= 0, top dims < 1
 * (correlation_layer.cpp:22,62-63), pad < max_displacement (the reference reads out of bounds
 * there; we refuse). */
int fn2_correlation_out_shape(const fn2_corr_params* p, int C, int H, int W,
                              int* topC, int* topH, int* topW);
/* No workspace is needed by the current kernels (the reference's rbot1_/rbot2_ padded NHWC
 * scratch, correlation_layer.cpp:76-82, is gone); always returns 0.  Kept so the ABI can grow. */
size_t fn2_correlation_workspace_bytes(const fn2_corr_params* p, int N, int C, int H, int W);
int fn2_correlation_forward(const fn2_corr_params* p,
                            const float* bottom0, const float* bottom1, float* top,
                            int N, int C, int H, int W,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same forward with the two passes that follow the layer in the FlowNetC graph folded into its epilogue:
 *   - the in-place ReLU{negative_slope} on the cost volume (ReLULayer::Forward_gpu, relu_layer.cu:8-27) when relu != 0;
 *   - the Concat with conv_redir (ConcatLayer::Forward_gpu, concat_layer.cu:8-52): the top is written as the channel slice
 *     [top_c0, top_c0 + topC) of a blob with top_channels channels (top_channels = 0: a plain [N, topC, topH, topW] top).
 * Same arithmetic as fn2_correlation_forward followed by those layers (the ReLU is applied to the final value). */
int fn2_correlation_forward_fused(const fn2_corr_params* p,
                                  const float* bottom0, const float* bottom1, float* top,
                                  int N, int C, int H, int W, int top_channels, int top_c0, int relu, float negative_slope,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* Writes (overwrites) both bottom diffs; like the reference it ignores propagate_down
 * (correlation_layer.cu:508-603).  Either diff pointer may be NULL to skip that half. */
int fn2_correlation_backward(const fn2_corr_params* p,
                             const float* bottom0, const float* bottom1, const float* top_diff,
                             float* bottom0_diff, float* bottom1_diff,
                             int N, int C, int H, int W,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Correlation1D  (type: "Correlation1D"; horizontal displacements only -- the DispNet variant of the layer)
 *   params  <- the same CorrelationParameter + single_direction, Correlation1DLayer::LayerSetUp, src/caffe/layers/correlation_layer1d.cpp:12-40
 *   shapes  <- Correlation1DLayer::Reshape, correlation_layer1d.cpp:42-92: zero padding in x only; topW = ceil((W+2p-2(md+kr))/s1),
 *              topH = ceil((H-2kr)/s1); topC = md/s2 + 1 (single_direction != 0) or 2(md/s2) + 1
 *   forward <- Correlation1DLayer::Forward_gpu, src/caffe/layers/correlation_layer1d.cu:429-510 (CorrelateData :47-113, ...Subtract :251-293);
 *              displacement of top channel c is (c + x_shift) * s2 with x_shift = -md/s2 (both), 0 (right), -topC (left, :466-471)
 *   backward<- Correlation1DLayer::Backward_gpu, correlation_layer1d.cu:513-616 (kernels :117-249, :295-423)
 * single_direction = -1 makes the reference read up to s2 elements before the start of a padded row (x_shift = -topC, one more than
 * the grid radius): in its flat [N,H,W+2p,C] scratch blob that is the end of the previous row (zero padding when p > 0), and for the
 * first row of the first sample memory in front of the blob (undefined; read as 0 here).  The kernels reproduce the flat indexing.
 * ---------------------------------------------------------------------------------------------- */
int fn2_correlation1d_out_shape(const fn2_corr_params* p, int C, int H, int W, int* topC, int* topH, int* topW);
int fn2_correlation1d_forward(const fn2_corr_params* p, const float* bottom0, const float* bottom1, float* top,
                              int N, int C, int H, int W, void* stream);
/* Overwrites both bottom diffs (propagate_down is ignored, correlation_layer1d.cu:513-616); either may be NULL to skip that half. */
int fn2_correlation1d_backward(const fn2_corr_params* p, const float* bottom0, const float* bottom1, const float* top_diff,
                               float* bottom0_diff, float* bottom1_diff, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FlowWarp  (type: "FlowWarp")
 *   params  <- FlowWarpParameter.fill_value {ZERO=1, NOT_A_NUMBER=2}, caffe.proto:553-560
 *   forward <- FlowWarpLayer::Forward_gpu, src/caffe/layers/flow_warp_layer.cu:357-458
 *              (CPU twin flow_warp_layer.cpp:58-117)
 *   backward<- FlowWarpLayer::Backward_gpu, src/caffe/layers/flow_warp_layer.cu:461-514
 *              (kernel :169-229; CPU twin flow_warp_layer.cpp:120-199)
 * ---------------------------------------------------------------------------------------------- */
enum { FN2_FILL_ZERO = 1, FN2_FILL_NAN = 2 };

/* image [N,C,H,W], flow [N,2,H,W] (channel 0 = x/u, 1 = y/v), warped [N,C,H,W]. */
int fn2_flow_warp_forward(const float* image, const float* flow, float* warped,
                          int N, int C, int H, int W, int fill_value, void* stream);
/* The same layer reading / writing CHANNEL SLICES of wider blobs: image = channels [image_c0, image_c0 + C) of a contiguous
 * [N, image_channels, H, W] blob, warped likewise inside [N, top_channels, H, W].  This is what the Concat layers around FlowWarp in
 * the FlowNet2 graphs amount to (concat_layer.cu:33-60 copies every bottom into the top; a producer that writes the slice and a
 * consumer that reads it make the copy disappear).  flow = channels [flow_c0, flow_c0 + 2) of a [N, flow_channels, H, W] blob. */
int fn2_flow_warp_forward_slices(const float* image, int image_channels, int image_c0,
                                 const float* flow, int flow_channels, int flow_c0,
                                 float* warped, int top_channels, int top_c0,
                                 int N, int C, int H, int W, int fill_value, void* stream);
/* image_diff [N,C,H,W] and flow_diff [N,2,H,W] are both overwritten.  propagate_* == 0 leaves the
 * corresponding diff zero, as flow_warp_layer.cu:507-508 does.  The reference accumulates image_diff with
 * float atomics (:197-200: order not deterministic); here the scatter is inverted through per-pixel
 * linked lists in the workspace (2 ints per pixel) and summed in a fixed order (bit-reproducible wherever
 * at most 24 source pixels land on one cell). */
size_t fn2_flow_warp_backward_workspace_bytes(int N, int C, int H, int W);
int fn2_flow_warp_backward(const float* image, const float* flow, const float* warped_diff,
                           float* image_diff, float* flow_diff,
                           int N, int C, int H, int W,
                           int propagate_image, int propagate_flow,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Resample  (type: "Resample")  -- forward only (resample_layer.cu:209-213 is LOG(FATAL))
 *   params  <- ResampleParameter {width=1,height=2,type=3,antialias=4,factor=5(unused)}, caffe.proto:665-677
 *   forward <- ResampleLayer::Forward_gpu, src/caffe/layers/resample_layer.cu:128-206
 *              (InterpolationKernel :39-95, NearestNeighborKernel :97-125)
 * ---------------------------------------------------------------------------------------------- */
enum { FN2_RESAMPLE_NEAREST = 1, FN2_RESAMPLE_LINEAR = 2, FN2_RESAMPLE_CUBIC = 3, FN2_RESAMPLE_AREA = 4 };

/* in [N,C,Hin,Win] -> out [N,C,Hout,Wout]; AREA -> FN2_ERR_UNSUPPORTED (resample_layer.cpp:17-20). */
int fn2_resample_forward(const float* in, float* out, int N, int C,
                         int Hin, int Win, int Hout, int Wout,
                         int type, int antialias, void* stream);
/* Resample with the layers the FlowNet2 graphs put around it folded in (FlowNet2_deploy.prototxt: Eltwise{coeff 20} -> Resample ->
 * {FlowWarp, Eltwise{coeff 0.05} -> Concat}): every tap is in * in_scale (one rounding, as the Eltwise top would have had:
 * eltwise_layer.cu:46-51 caffe_gpu_axpy), out goes to channels [top_c0, top_c0 + C) of a [N, top_channels, Hout, Wout] blob, and
 * out2 (may be NULL) = out * out2_scale, rounded from the rounded out, to a slice of a second blob.  in_scale = out2_scale = 1 and
 * full-width tops reproduce fn2_resample_forward bit for bit. */
int fn2_resample_forward_slices(const float* in, float in_scale, float* out, int top_channels, int top_c0,
                                float* out2, int top2_channels, int top2_c0, float out2_scale,
                                int N, int C, int Hin, int Win, int Hout, int Wout,
                                int type, int antialias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * L1Loss  (type: "L1Loss")
 *   params  <- L1LossParameter, caffe.proto:619-625
 *   forward <- L1LossLayer::Forward_gpu, src/caffe/layers/l1loss_layer.cu:67-143
 *   backward<- L1LossLayer::Backward_gpu, src/caffe/layers/l1loss_layer.cu:146-188
 * The reference keeps mask_/sign_/normalize_coeff_ as layer state between the two calls
 * (l1_loss_layer.hpp); here forward leaves {loss, normalize_coeff} in the workspace and backward
 * recomputes the element-wise terms from the inputs, so the pair stays stateless.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fn2_l1loss_params {
  int l2_per_location;          /* = 1 [false] */
  int l2_prescale_by_channels;  /* = 2 [false] */
  int normalize_by_num_entries; /* = 3 [false] */
  float epsilon;                /* = 4 [1e-2]  */
  float plateau;                /* = 3001 [0]  */
} fn2_l1loss_params;

size_t fn2_l1loss_workspace_bytes(int N, int C, int H, int W);
/* bottom1 may be NULL (single-bottom form, l1loss_layer.cpp:15-17).  loss_out: DEVICE float[1]
 * (the reference writes the scalar on the host, l1loss_layer.cu:142; we keep it on the device so
 * there is no sync).  The first 2 floats of the workspace hold {loss, normalize_coeff} afterwards. */
int fn2_l1loss_forward(const fn2_l1loss_params* p, const float* bottom0, const float* bottom1,
                       float* loss_out, int N, int C, int H, int W,
                       void* workspace, size_t workspace_bytes, void* stream);
/* top_diff = top[0]->cpu_diff()[0] (the loss weight, layer.hpp:444-458).  Must be called with the
 * workspace the matching forward filled.  bottom1_diff may be NULL. */
int fn2_l1loss_backward(const fn2_l1loss_params* p, const float* bottom0, const float* bottom1,
                        float top_diff, float* bottom0_diff, float* bottom1_diff,
                        int N, int C, int H, int W,
                        void* workspace, size_t workspace_bytes, void* stream);

/* All L1Loss layers of a net in ONE launch per direction (FlowNet's training prototxts hold five: flow_loss2 .. flow_loss6, each with
 * its loss_weight; Net::ForwardFromTo adds loss_weight * loss in layer order, net.cpp:565-579 / layer.hpp:434-440).  Per scale the
 * results are bit-identical to fn2_l1loss_forward / fn2_l1loss_backward with top_diff = loss_weight * total_diff[0].
 *   losses: DEVICE float[nscales] or NULL; total: DEVICE float[1] = sum_s loss_weight[s] * loss[s] (scale order, float) or NULL;
 *   total_diff: DEVICE float[1] (the objective's derivative w.r.t. total; NULL = 1): no host read of a device scalar anywhere;
 *   workspace: fn2_l1loss_multi_workspace_bytes(nscales), filled by the forward, read by the backward;
 *   sync: fn2_l1loss_multi_sync_bytes() of device memory that is ZERO before the first call (every call leaves it zero; one per
 *   stream that may run a forward concurrently). */
typedef struct fn2_l1loss_scale {
  const float* bottom0; const float* bottom1;   /* bottom1 may be NULL */
  float* bottom0_diff; float* bottom1_diff;     /* backward only; bottom1_diff may be NULL */
  int N, C, H, W;
  float loss_weight;
} fn2_l1loss_scale;
size_t fn2_l1loss_multi_workspace_bytes(int nscales);
size_t fn2_l1loss_multi_sync_bytes(void);
int fn2_l1loss_forward_multi(const fn2_l1loss_params* p, int nscales, const fn2_l1loss_scale* scales, float* losses, float* total,
                             void* workspace, size_t workspace_bytes, void* sync, void* stream);
int fn2_l1loss_backward_multi(const fn2_l1loss_params* p, int nscales, const fn2_l1loss_scale* scales, const float* total_diff,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ChannelNorm  (type: "ChannelNorm")
 *   forward <- ChannelNormLayer::Forward_gpu, src/caffe/layers/channel_norm_layer.cu:50-67
 *   backward<- ChannelNormLayer::Backward_gpu, :69-90 (which passes host pointers to the kernel --
 *              a reference bug; we implement what NormBackward :37-47 computes)
 * ---------------------------------------------------------------------------------------------- */
int fn2_channel_norm_forward(const float* bottom, float* top, int N, int C, int H, int W, void* stream);
/* ChannelNorm over channel slices of wider blobs; minus != NULL: the norm of (bottom - minus), i.e. the Eltwise{SUM, coeff 1, -1}
 * the FlowNet2 graphs put in front of the layer (img0 - warped img1) folded in: the difference is rounded to fp32 before it is
 * squared, as the Eltwise top would have been.  top = channel top_c0 of a [N, top_channels, H, W] blob. */
int fn2_channel_norm_forward_slices(const float* bottom, int bottom_channels, int bottom_c0,
                                    const float* minus, int minus_channels, int minus_c0,
                                    float* top, int top_channels, int top_c0,
                                    int N, int C, int H, int W, void* stream);
int fn2_channel_norm_backward(const float* bottom, const float* top, const float* top_diff,
                              float* bottom_diff, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Downsample  (type: "Downsample") -- forward only
 *   forward <- DownsampleLayer::Forward_gpu, src/caffe/layers/downsample_layer.cu:75-129
 *              (DownsampleFeatures :15-72).  Same-size input: the reference shares the blob
 *              (downsample_layer.cpp:53-56); we copy.
 * ---------------------------------------------------------------------------------------------- */
int fn2_downsample_forward(const float* bottom, float* top, int N, int C,
                           int Hin, int Win, int Hout, int Wout, void* stream);
/* `count` (1..8) Downsample layers on the SAME bottom in one launch -- the ground-truth pyramid of the multi-scale loss (five
 * Downsample layers on one blob in the training prototxts): tops[j] is [N, C, top_heights[j], top_widths[j]], every top at least
 * 2 x 2 and of another size than the bottom.  Same arithmetic and tap order as `count` calls of fn2_downsample_forward (bit-identical). */
int fn2_downsample_forward_multi(const float* bottom, float* const* tops, const int* top_heights, const int* top_widths,
                                 int count, int N, int C, int Hin, int Win, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flow heads of the FlowNet decoders -- stock Caffe layers in the reference, specialised here because a
 * 2-output-channel GEMM cannot fill a matrix tile:
 *   predict_flow*  = Convolution{kernel_size 3, stride 1, pad 1, num_output 2}
 *                    <- ConvolutionLayer::Forward_gpu, src/caffe/layers/conv_layer.cu:8-23
 *                       (BaseConvolutionLayer::forward_gpu_gemm, base_conv_layer.cpp:325-341; weight [2,C,3,3])
 *   upsample_flow* = Deconvolution{kernel_size 4, stride 2, pad 1, num_output 2} on a 2-channel flow
 *                    <- DeconvolutionLayer::Forward_gpu, src/caffe/layers/deconv_layer.cu (CPU twin deconv_layer.cpp:8-45;
 *                       weight [Cin=2, Cout=2, 4, 4], base_conv_layer.cpp:125-139)
 * bias may be NULL.
 * ---------------------------------------------------------------------------------------------- */
size_t fn2_predict_flow_conv_workspace_bytes(int N, int C, int H, int W);   /* the per-tap partial image (18 planes per channel split) */
int fn2_predict_flow_conv_forward(const float* in, const float* weight, const float* bias, float* out,
                                  int N, int C, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
int fn2_upsample_flow_deconv_forward(const float* in, const float* weight, const float* bias, float* out,
                                     int N, int H, int W, void* stream);
/* Backward of the two heads (round 3; csrc/flow_head_bwd.hip): streaming kernels over NCHW, fixed summation order (deterministic).
 *   predict_flow:  <- ConvolutionLayer::Backward_gpu, src/caffe/layers/conv_layer.cu:26-60 (backward_gpu_bias, weight_gpu_gemm,
 *                     backward_gpu_gemm: base_conv_layer.cpp:352-393) with weight [2, C, 3, 3]; bottom may be a channel slice
 *                     [bottom_c0, bottom_c0 + C) of a [N, bottom_channels, H, W] blob (the Concat it reads).
 *   upsample_flow: <- DeconvolutionLayer::Backward_gpu, src/caffe/layers/deconv_layer.cu:27-58 with weight [2, 2, 4, 4];
 *                     bottom [N, 2, H, W], top_diff [N, 2, 2H, 2W].
 * Any of bottom_diff / weight_diff / bias_diff may be NULL (= propagate_down false).  accumulate != 0 adds into weight_diff / bias_diff
 * like the reference (the solver clears the diffs once per iteration); bottom_diff is always overwritten.  Workspace: the per-part
 * partial sums (only needed for weight_diff / bias_diff).
 * fn2_predict_flow_conv_backward_supported: 1 when the weight-gradient kernel has a row band that fits LDS for this geometry (the band height
 * is a function of N, C, H and W: rows of up to 2,558 pixels), 0 otherwise -- ask before routing a layer here. */
int fn2_predict_flow_conv_backward_supported(int N, int C, int H, int W);
size_t fn2_predict_flow_conv_backward_workspace_bytes(int N, int C, int H, int W);
int fn2_predict_flow_conv_backward(const float* bottom, int bottom_channels, int bottom_c0, const float* weight, const float* top_diff,
                                   float* bottom_diff, float* weight_diff, float* bias_diff, int N, int C, int H, int W,
                                   int accumulate, void* workspace, size_t workspace_bytes, void* stream);
size_t fn2_upsample_flow_deconv_backward_workspace_bytes(int N, int H, int W);
int fn2_upsample_flow_deconv_backward(const float* bottom, const float* weight, const float* top_diff, float* bottom_diff,
                                      float* weight_diff, float* bias_diff, int N, int H, int W, int accumulate,
                                      void* workspace, size_t workspace_bytes, void* stream);
/* The same layer written into channels [top_c0, top_c0 + 2) of a wider top blob [N, top_channels, 2H, 2W]: the upsampled flow is
 * the last input of the refinement stages' Concat layers (concat_layer.cu:8-52), which then has nothing left to copy. */
int fn2_upsample_flow_deconv_forward_into(const float* in, const float* weight, const float* bias, float* top,
                                          int N, int H, int W, int top_channels, int top_c0, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution bias + leaky ReLU, in place (one pass): top[n,c,:,:] = f(top[n,c,:,:] + bias[c]), f(t) = t > 0 ? t : t * negative_slope.
 *   <- BaseConvolutionLayer::forward_gpu_bias, src/caffe/layers/base_conv_layer.cpp:343-348 (bias term of Convolution /
 *      Deconvolution) followed by the in-place ReLULayer::Forward_gpu, src/caffe/layers/relu_layer.cu:8-27
 *      (relu_param.negative_slope = 0.1 in every FlowNet prototxt).  bias may be NULL (plain leaky ReLU).
 * Forward only; the caller runs the convolution itself without its bias term.
 * ---------------------------------------------------------------------------------------------- */
int fn2_bias_leaky_relu_forward(float* data, const float* bias, int N, int C, int H, int W, float negative_slope,
                                void* stream);
/* Backward of the same pair: bottom_diff = top_diff * (top_data > 0 ? 1 : negative_slope)
 *   <- ReLULayer::Backward_gpu, src/caffe/layers/relu_layer.cu:33-60 (in-place blob: the output's sign is the input's),
 * and bias_diff[c] = sum over n, h, w of bottom_diff  <- backward_gpu_bias, base_conv_layer.cpp:389-393 (bias_diff may be
 * NULL).  bottom_diff may alias top_diff.  Deterministic (fixed-order partial sums in the workspace). */
size_t fn2_bias_leaky_relu_backward_workspace_bytes(int N, int C, int H, int W);
/* Deploy head in one pass: top[n, top_c0 + c] = bottom[n, c] * scale + shift[c]  (product rounded, then the sum rounded: no fma)
 *   <- EltwiseLayer::Forward_gpu SUM with one bottom and coeff = scale, src/caffe/layers/eltwise_layer.cu:46-52 (scripts/run-flownet.py's
 *      deploy nets scale the raw 0..255 images by 1/255), then the per-channel mean subtraction of the deploy-time DataAugmentation layer,
 *      data_augmentation_layer.cu:592-621 (shift = -mean), the LINEAR Resample between them being the identity at the ADAPTED size.
 *   top is [N, top_channels, H, W]: the two images land side by side in the blob conv1 reads.  shift may be NULL. */
int fn2_scale_shift_forward(const float* bottom, float* top, const float* shift, int N, int C, int H, int W,
                            int top_channels, int top_c0, float scale, void* stream);
int fn2_bias_leaky_relu_backward(const float* top_data, const float* top_diff, float* bottom_diff, float* bias_diff,
                                 int N, int C, int H, int W, float negative_slope, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* The same with top_diff = channels [diff_c0, diff_c0 + C) of a [N, diff_channels, H, W] blob: the gradient a Concat hands its bottoms
 * (concat_layer.cu:62-90) read in place instead of copied out first. */
int fn2_bias_leaky_relu_backward_slices(const float* top_data, const float* top_diff, int diff_channels, int diff_c0,
                                        float* bottom_diff, float* bias_diff, int N, int C, int H, int W, float negative_slope,
                                        void* workspace, size_t workspace_bytes, void* stream);
/* ... and with top_data = channels [data_c0, data_c0 + C) of a [N, data_channels, H, W] blob as well: a layer whose output was written straight
 * into its consumer's Concat blob (the refinement stages of a training graph, round 5). */
int fn2_bias_leaky_relu_backward_slices2(const float* top_data, int data_channels, int data_c0, const float* top_diff, int diff_channels,
                                         int diff_c0, float* bottom_diff, float* bias_diff, int N, int C, int H, int W, float negative_slope,
                                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stem convolution of the FlowNet encoders, fused with its bias and ReLU:
 *   top = leaky_relu(Convolution{kernel_size 7, stride 2, pad 3}(bottom) + bias, negative_slope)
 *   <- conv1 + ReLU1 of FlowNetC / FlowNetS: ConvolutionLayer::Forward_gpu, src/caffe/layers/conv_layer.cu:8-23
 *      (forward_gpu_gemm + forward_gpu_bias, base_conv_layer.cpp:325-348; weight [Cout, Cin, 7, 7]) followed by the in-place
 *      ReLULayer::Forward_gpu, src/caffe/layers/relu_layer.cu:8-27.
 *   top is [N, Cout, (Hin - 1) / 2 + 1, (Win - 1) / 2 + 1].  fn2_conv_k7s2_relu_supported tells whether this build has a
 *   kernel for the shape (Cin 3, 6 or 12, Cout % 64 == 0, Win % 8 == 0); callers keep the library convolution otherwise.
 *   Forward only.
 * ---------------------------------------------------------------------------------------------- */
int fn2_conv_k7s2_relu_supported(int Cin, int Hin, int Win, int Cout);
int fn2_conv_k7s2_relu_forward(const float* bottom, const float* weight, const float* bias, float* top,
                               int N, int Cin, int Hin, int Win, int Cout, float negative_slope, void* stream);
/* Weight gradient of the same stem (round 4; csrc/conv_stem_wgrad.hip) <- ConvolutionLayer::Backward_gpu -> weight_gpu_gemm,
 * src/caffe/layers/conv_layer.cu:40-52, base_conv_layer.cpp:368-384 (per sample im2col + SGEMM(top_diff x col^T), beta = 1):
 *   weight_diff[co][ci][ky][kx] (+)= sum_{n, y, x} top_diff[n][co][y][x] * bottom[n][ci][2 y + ky - 3][2 x + kx - 3]
 * on the fp32 MFMA with the TAP axis as the GEMM's N axis (3 bottom channels would fill 3 / 16 of a channel tile), deterministic: the
 * pixels are cut into fn2_conv_k7s2_wgrad_ksplit() parts whose partial sums (workspace) are added in part order.  Cin in {3, 6, 12},
 * Cout == 64, width % 8 == 0, both blobs 16-byte aligned and contiguous.  accumulate != 0 adds into weight_diff like the reference. */
int fn2_conv_k7s2_wgrad_supported(int N, int Cin, int Hin, int Win, int Cout);
int fn2_conv_k7s2_wgrad_ksplit(int N, int Cin, int Hin, int Win, int Cout);
size_t fn2_conv_k7s2_wgrad_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout);
int fn2_conv_k7s2_wgrad(const float* top_diff, const float* bottom, float* weight_diff, int N, int Cin, int Hin, int Win, int Cout,
                        int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Direct convolution of the FlowNet encoders on the fp32 matrix cores, fused with its bias and (optionally) its ReLU:
 *   top[:, top_c0 : top_c0 + Cout] = act(Convolution{kernel_size, stride, pad}(bottom[:, bottom_c0 : bottom_c0 + Cin]) + bias)
 *   <- ConvolutionLayer::Forward_gpu, src/caffe/layers/conv_layer.cu:8-23 (per-sample im2col_gpu + cublasSgemm in
 *      forward_gpu_gemm, base_conv_layer.cpp:326-341, then forward_gpu_bias :343-348; weight [Cout, Cin, k, k]) and, when
 *      relu != 0, the in-place ReLULayer::Forward_gpu, src/caffe/layers/relu_layer.cu:8-27.
 *   One launch for the mini-batch, NCHW in and out, no column matrix.  bottom / top may be channel slices of wider blobs
 *   (bottom_channels / top_channels = the blobs' channel counts): a Concat consumer (concat_layer.cu) can be written in place.
 *   The weight operand is the PACKED form of the layer's weight blob (MFMA operand order, zero-padded to whole channel
 *   quads): fn2_conv_mfma_packed_floats() floats, written by fn2_conv_mfma_pack_weights() -- once per weight update
 *   (LayerSetUp / after Solver::ApplyUpdate), not per forward.
 *   Supported (fn2_conv_mfma_supported): kernel 3 (stride 1 or 2), 5 (stride 2) or 7 (stride 2), pad <= 4, Cout % 64 == 0, Win % 4 == 0,
 *   16-byte aligned blobs; callers keep the library convolution otherwise.  Forward only.  Exact fp32 (k-ordered fma chains;
 *   every kernel variant produces the same bits).
 * ---------------------------------------------------------------------------------------------- */
int fn2_conv_mfma_supported(int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad);
size_t fn2_conv_mfma_packed_floats(int Cout, int Cin, int kernel);
int fn2_conv_mfma_pack_weights(const float* weight, float* packed, int Cout, int Cin, int kernel, void* stream);
/* The same operand from a strided VIEW of a weight blob: logical element (co, ci, ky, kx) of the [Cout][Cin][k][k] operand =
 * weight[co * stride_cout + ci * stride_cin + (flip ? k*k - 1 - (ky*k + kx) : ky*k + kx)] for co < src_cout, ci < src_cin, 0 beyond.
 * One launch instead of transpose / flip / zero-pad passes in front of the packing: the channel-swapped blob of a data gradient
 * (ConvolutionLayer::Backward_gpu, conv_layer.cu:53-57; DeconvolutionLayer::Backward_gpu, deconv_layer.cu:52-56), its 180-degree
 * rotation (3x3 / stride 1), channel padding up to the kernels' group sizes, and the [Cout k k][Cin] GEMM operand of a
 * Deconvolution (kernel 1, stride_cout 1, stride_cin Cout k k: base_conv_layer.cpp:375-384's weight^T). */
int fn2_conv_mfma_pack_weights_view(const float* weight, float* packed, int Cout, int Cin, int kernel, int src_cout, int src_cin,
                                    long long stride_cout, long long stride_cin, int flip, void* stream);
int fn2_conv_mfma_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                          int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                          int Cout, int top_channels, int top_c0, int kernel, int stride, int pad,
                          int relu, float negative_slope, void* stream);
/* Test / profiling hooks: number of tile variants, and a forced variant (-1 = choose by the cost model; v = plain launch of
 * variant v; 1000 + v = its split-tail launch: whole rounds of tiles, the remainder as half-channel workgroups). */
int fn2_conv_mfma_num_variants(void);
int fn2_debug_set_conv_variant(int variant);

/* ------------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution (+ bias, + optional ReLU) as Winograd F(2x2, 3x3) on the fp32 matrix cores:
 *   same layer and blob conventions as fn2_conv_mfma_forward (conv_layer.cu:8-23, base_conv_layer.cpp:326-348, relu_layer.cu:8-27;
 *   channel slices on both blobs), 2.25x fewer multiplies: Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A per 2x2 output tile.
 *   The weight operand is U = G g G^T in MFMA operand order, fn2_conv_wino_packed_floats() floats written by
 *   fn2_conv_wino_pack_weights() once per weight update.  fp32 products and accumulation; the transforms add a few ulp of rounding
 *   relative to the direct sum (the library kernels the reference's cuDNN / MIOpen builds use for this layer are Winograd as well).
 *   Supported (fn2_conv_wino_supported): Cout % 16 == 0, Win % 4 == 0, pad 1, 16-byte aligned blobs.  Forward only.
 * ---------------------------------------------------------------------------------------------- */
int fn2_conv_wino_supported(int Cin, int Hin, int Win, int Cout, int pad);
size_t fn2_conv_wino_packed_floats(int Cout, int Cin);
int fn2_conv_wino_pack_weights(const float* weight, float* packed, int Cout, int Cin, void* stream);
int fn2_conv_wino_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                          int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                          int Cout, int top_channels, int top_c0, int pad, int relu, float negative_slope, void* stream);
int fn2_conv_wino_num_variants(void);
int fn2_debug_set_wino_variant(int variant);     /* as fn2_debug_set_conv_variant */

/* ------------------------------------------------------------------------------------------------
 * 3x3 convolution (+ bias, + optional ReLU) for SMALL feature maps (the encoder layers below 1/16 resolution: conv4 .. conv6_1):
 *   same layer, blob and weight conventions as fn2_conv_mfma_forward with kernel 3 (conv_layer.cu:8-23, base_conv_layer.cpp:326-348,
 *   relu_layer.cu:8-27; packed_weight = fn2_conv_mfma_pack_weights(kernel 3); channel slices on both blobs).
 *   The output pixels of a group of samples are flattened into MFMA M tiles, whole input planes live in LDS, and the channel
 *   (K) axis is split over fn2_conv_plane_ksplit() workgroups whose partial sums (workspace: fn2_conv_plane_workspace_bytes())
 *   a second kernel adds in part order before bias and ReLU.  ksplit depends on the layer geometry only, so the result is
 *   one fixed fp32 summation order: per part the k-ordered fma chain of fn2_conv_mfma_forward, parts added in order.
 *   Part p covers the 2-quad units [p U / ksplit, (p + 1) U / ksplit) of the channel axis, U = ceil(Cin / 8).
 *   Supported (fn2_conv_plane_supported): stride 1 or 2, pad 0 or 1, Cin % 8 == 0, Cout % 64 == 0, a padded input plane of at most
 *   768 floats (3072 when Win % 4 == 0); 16-byte aligned blobs and workspace.  Forward only.
 * ---------------------------------------------------------------------------------------------- */
int fn2_conv_plane_supported(int N, int Cin, int Hin, int Win, int Cout, int stride, int pad);
int fn2_conv_plane_ksplit(int N, int Cin, int Hin, int Win, int Cout, int stride, int pad);
size_t fn2_conv_plane_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout, int stride, int pad);
int fn2_conv_plane_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                           int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                           int Cout, int top_channels, int top_c0, int stride, int pad,
                           int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream);
/* The same kernel with a `kernel` argument: 3 (as above) or 4 with stride 2 / pad 1 -- the DATA GRADIENT of a Deconvolution{4, 2, 1}
 * on a small map (DeconvolutionLayer::Backward_gpu, deconv_layer.cu:52-56: forward_gpu_gemm of top_diff = the 4x4 / 2 / 1 CONVOLUTION of
 * top_diff with the layer's weight blob [Cin_deconv = output channels here][Cout_deconv = input channels here][4][4], packed by
 * fn2_conv_mfma_pack_weights); maps whose width is not a multiple of 4 included (deconv5 of FlowNetC: 10x14 -> 5x7) -- or 5 with
 * stride 2 / pad 2: conv2 / conv3 of the encoders (FlowNet2_deploy.prototxt.template: Convolution{kernel_size 5, stride 2, pad 2}) when ONE
 * sample has to fill the chip (FlowNet2 at batch 1: conv3 [1,128,112,256] -> 256 needs the K split the direct kernel does not have). */
int fn2_conv_plane_k_supported(int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad);
int fn2_conv_plane_k_ksplit(int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad);
size_t fn2_conv_plane_k_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout, int kernel, int stride, int pad);
int fn2_conv_plane_k_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                             int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                             int Cout, int top_channels, int top_c0, int kernel, int stride, int pad,
                             int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream);
int fn2_conv_plane_num_variants(void);
int fn2_debug_set_plane_variant(int variant);    /* as fn2_debug_set_conv_variant (no split-tail forms) */
int fn2_debug_set_plane_ksplit(int ksplit);      /* > 0: force the number of K parts (changes the summation order); 0: by geometry */
/* Batch-invariant summation orders (scripts/run-flownet-many.py:27-81 writes one .flo per pair: the bits of a pair must not depend
 * on the batch it was computed in, nor on how the list was sharded over GPUs).  on != 0: every quantity that fixes a summation order
 * and would otherwise depend on N (the K parts of fn2_conv_plane_forward / fn2_deconv_plane_forward) is computed for a batch of ONE
 * sample.  Every other kernel of this library is batch-invariant by construction (all tile variants run the same k-ordered chain). */
int fn2_set_batch_invariant(int on);
int fn2_get_batch_invariant(void);

/* ------------------------------------------------------------------------------------------------
 * Deconvolution{kernel 4, stride 2, pad 1} (+ bias, + optional ReLU) of the refinement stages (deconv5 .. deconv2), same kernel family:
 *   top[:, top_c0 : top_c0 + Cout] = act(Deconvolution(bottom[:, bottom_c0 : bottom_c0 + Cin]) + bias), top is [N, *, 2 Hin, 2 Win]
 *   <- DeconvolutionLayer::Forward_gpu, src/caffe/layers/deconv_layer.cu:8-26 (per sample backward_gpu_gemm = weight^T x bottom, then
 *      col2im_gpu: base_conv_layer.cpp:375-393, im2col.cu:246-318; then forward_gpu_bias; weight blob [Cin, Cout, 4, 4]) and, when
 *      relu != 0, the in-place ReLULayer::Forward_gpu, relu_layer.cu:8-27.
 *   No column matrix: every output parity class (Y % 2, X % 2) is a 2x2-tap convolution of the input, one class per wave on a shared
 *   LDS window; the top blob may be a channel slice of the consumer's Concat blob (concat_layer.cu).  packed_weight =
 *   fn2_deconv_plane_pack_weights(weight) (fn2_deconv_plane_packed_floats() floats), once per weight update.  K split and workspace as
 *   for fn2_conv_plane_forward (fn2_deconv_plane_ksplit / fn2_deconv_plane_workspace_bytes).
 *   Supported (fn2_deconv_plane_supported): Cout % 64 == 0, a padded input plane of at most 768 floats (3072 when Win % 4 == 0), any Cin;
 *   16-byte aligned blobs and workspace.  Forward only.
 * ---------------------------------------------------------------------------------------------- */
int fn2_deconv_plane_supported(int N, int Cin, int Hin, int Win, int Cout);
int fn2_deconv_plane_ksplit(int N, int Cin, int Hin, int Win, int Cout);
size_t fn2_deconv_plane_workspace_bytes(int N, int Cin, int Hin, int Win, int Cout);
size_t fn2_deconv_plane_packed_floats(int Cin, int Cout);
int fn2_deconv_plane_pack_weights(const float* weight, float* packed, int Cin, int Cout, void* stream);
/* src_kernel 4: as above; 3: a [Cin][Cout][3][3] blob read as the 4x4 blob whose fourth tap row and column are zero -- the transposed
 * 3x3 / 2 / 1 convolution that is the data gradient of conv5 / conv6 (conv_layer.cu:53-57) on maps the 16-byte kernels do not take. */
int fn2_deconv_plane_pack_weights_k(const float* weight, float* packed, int Cin, int Cout, int src_kernel, void* stream);
int fn2_deconv_plane_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                             int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                             int Cout, int top_channels, int top_c0, int relu, float negative_slope,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution / Deconvolution by DESCRIPTOR: the library picks the kernel family (csrc/conv_route.cpp).
 *   <- ConvolutionLayer / DeconvolutionLayer::{LayerSetUp, Reshape, Forward_gpu} (base_conv_layer.cpp:14-253 shapes, conv_layer.cu:8-23,
 *      deconv_layer.cu:8-26) + the in-place ReLU that follows them in the FlowNet graphs (relu_layer.cu:8-27).
 *   One decision function for every caller (the Python mirror and the Caffe adapter ask it): fn2_conv_route() returns which family serves
 *   Convolution{kernel, stride, pad} Cin -> Cout on [N, Cin, Hin, Win] -- Winograd F(2x2,3x3), the small-map kernel (deterministic K split),
 *   the direct kernel -- or NONE (the caller keeps its library layer).  The weight operand is packed per ROUTE
 *   (fn2_conv_packed_weight_floats / fn2_conv_pack_weights: once per weight update, LayerSetUp / after Solver::ApplyUpdate), the forward takes
 *   channel slices on both blobs like the kernels it dispatches to.  FN2_ROUTE_FORCE (tests): Winograd wherever it applies, the small-map
 *   kernel whatever the map size.  fn2_set_batch_invariant(1): the route is decided as for ONE sample.
 *   Deconvolution{4, 2, 1} (weight blob [Cin][Cout][4][4], top [N, Cout, 2 Hin, 2 Win]): GEMM (weight^T x bottom on the 1x1 form of the
 *   direct kernel into a column matrix in the WORKSPACE, then col2im + bias + ReLU in one pass) or the parity-class small-map kernel.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fn2_conv_desc {
  int N, Cin, Hin, Win;      /* bottom [N, Cin, Hin, Win] (a channel slice of a wider blob is given to the forward call) */
  int Cout, kernel, stride, pad;
} fn2_conv_desc;
/* STEM: Convolution{7, 2, 3} on 3 / 6 / 12 channels (csrc/conv_stem.hip); HEAD: the 2-channel flow heads -- Convolution{3, 1, 1} -> 2
 * (predict_flow*) and Deconvolution{4, 2, 1} 2 -> 2 (upsample_flow*), csrc/flow_head.hip.  Both read the weight blob as it is ("packing" is a
 * device copy) and whole blobs only (no channel slices on the bottom; the stem and predict_flow none on the top either). */
enum { FN2_CONV_ROUTE_NONE = 0, FN2_CONV_ROUTE_DIRECT = 1, FN2_CONV_ROUTE_WINOGRAD = 2, FN2_CONV_ROUTE_PLANE = 3, FN2_CONV_ROUTE_STEM = 4,
       FN2_CONV_ROUTE_HEAD = 5 };
enum { FN2_DECONV_ROUTE_NONE = 0, FN2_DECONV_ROUTE_GEMM = 1, FN2_DECONV_ROUTE_PLANE = 2, FN2_DECONV_ROUTE_HEAD = 3 };
enum { FN2_ROUTE_FORCE = 1 };
int fn2_conv_route(const fn2_conv_desc* desc, int flags);
size_t fn2_conv_packed_weight_floats(const fn2_conv_desc* desc, int route);
int fn2_conv_pack_weights(const fn2_conv_desc* desc, int route, const float* weight, float* packed, void* stream);
size_t fn2_conv_workspace_bytes(const fn2_conv_desc* desc, int route);
int fn2_conv_forward(const fn2_conv_desc* desc, int route, const float* bottom, int bottom_channels, int bottom_c0,
                     const float* packed_weight, const float* bias, float* top, int top_channels, int top_c0,
                     int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream);
int fn2_deconv_route(const fn2_conv_desc* desc, int flags);
size_t fn2_deconv_packed_weight_floats(const fn2_conv_desc* desc, int route);
int fn2_deconv_pack_weights(const fn2_conv_desc* desc, int route, const float* weight, float* packed, void* stream);
size_t fn2_deconv_workspace_bytes(const fn2_conv_desc* desc, int route);
int fn2_deconv_forward(const fn2_conv_desc* desc, int route, const float* bottom, int bottom_channels, int bottom_c0,
                       const float* packed_weight, const float* bias, float* top, int top_channels, int top_c0,
                       int relu, float negative_slope, void* workspace, size_t workspace_bytes, void* stream);

/* Backward by descriptor (round 5).  `transposed` = 0: a Convolution (desc as for fn2_conv_route), 1: a Deconvolution{4, 2, 1} (desc as for
 * fn2_deconv_route: Cin = bottom channels, Cout = top channels, top_diff is [N, Cout, 2 Hin, 2 Win]).  No activation inside: the ReLU of the
 * FlowNet graphs is undone first (fn2_bias_leaky_relu_backward, which also reduces the bias gradient), as ReLULayer::Backward_gpu does in Caffe.
 *   data gradient   <- ConvolutionLayer::Backward_gpu -> backward_gpu_gemm (weight^T x top_diff + col2im), conv_layer.cu:53-57,
 *                      base_conv_layer.cpp:352-366; DeconvolutionLayer::Backward_gpu -> forward_gpu_gemm of top_diff, deconv_layer.cu:52-56.
 *     fn2_conv_backward_data_route(): WINOGRAD (3x3 / 1 / 1: the forward Winograd kernel on the 180-degree-rotated, channel-swapped weights),
 *     TCONV (5x5 / 2 / 2 and 3x3 / 2 / 1: stride-2 transposed convolution, csrc/tconv_mfma.hip), DECONV_PLANE (3x3 / 2 / 1 on maps whose width is
 *     no multiple of 4: the small-map deconvolution kernel on the blob read as 4x4 with zero taps), PLANE (3x3 / 1 / 1 on maps the Winograd
 *     kernel does not take; Deconvolution on small maps: the small-map kernel with 4x4 / 2 taps), DIRECT (1x1 on the transposed weight;
 *     Deconvolution: its gradient IS the 4x4 / 2 / 1 convolution of top_diff), NONE.  The operand is packed per route
 *     (fn2_conv_backward_data_pack_weights; the WINOGRAD route needs fn2_conv_backward_data_pack_workspace_bytes() of scratch for the rotated
 *     blob).  bottom_diff is OVERWRITTEN (Caffe: col2im writes the blob).  The kernels compute channels in groups (16 / 32 / 64):
 *     fn2_conv_backward_data_computed_channels() >= Cin of them, the surplus being zeros.  `bottom_room` = how many channels from bottom_c0 on
 *     the call may overwrite: with room for the computed channels the kernel writes bottom_diff directly, otherwise (bottom_room = Cin: a
 *     Caffe blob) the result goes through the workspace and its first Cin channels are copied (fn2_conv_backward_data_workspace_bytes covers
 *     that and the K-split slabs).
 *   weight gradient <- weight_gpu_gemm, conv_layer.cu:40-52 / deconv_layer.cu:36-50, base_conv_layer.cpp:368-384 (beta = 1: accumulate != 0
 *     adds into weight_diff): fn2_conv_wgrad (every FlowNet class) or, for the 3-channel 7x7 / 2 stem, fn2_conv_k7s2_wgrad (contiguous blobs).
 *   bias gradient   <- backward_gpu_bias, base_conv_layer.cpp:389-393 (beta = 1): fn2_conv_backward_bias; workspace
 *     fn2_bias_leaky_relu_backward_workspace_bytes(N, C, H, W). */
enum { FN2_BWD_ROUTE_NONE = 0, FN2_BWD_ROUTE_WINOGRAD = 1, FN2_BWD_ROUTE_TCONV = 2, FN2_BWD_ROUTE_PLANE = 3, FN2_BWD_ROUTE_DIRECT = 4,
       FN2_BWD_ROUTE_DECONV_PLANE = 5 };
int fn2_conv_backward_data_route(const fn2_conv_desc* desc, int transposed);
size_t fn2_conv_backward_data_packed_weight_floats(const fn2_conv_desc* desc, int transposed, int route);
size_t fn2_conv_backward_data_pack_workspace_bytes(const fn2_conv_desc* desc, int transposed, int route);
int fn2_conv_backward_data_pack_weights(const fn2_conv_desc* desc, int transposed, int route, const float* weight, float* packed,
                                        void* workspace, size_t workspace_bytes, void* stream);
size_t fn2_conv_backward_data_workspace_bytes(const fn2_conv_desc* desc, int transposed, int route);
size_t fn2_conv_backward_data_workspace_bytes_with_room(const fn2_conv_desc* desc, int transposed, int route, int bottom_room);   /* bottom_room >= computed channels: the kernel's scratch only */
int fn2_conv_backward_data_computed_channels(const fn2_conv_desc* desc, int transposed, int route);
int fn2_conv_backward_data(const fn2_conv_desc* desc, int transposed, int route, const float* top_diff, int top_channels, int top_c0,
                           const float* packed_weight, float* bottom_diff, int bottom_channels, int bottom_c0, int bottom_room,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same data gradient with ReLUBackward of the layer IN FRONT folded into its epilogue (round 6): bottom_data is that layer's activated
 * output = this layer's bottom blob; bottom_diff = (weight^T x top_diff) * (bottom_data > 0 ? 1 : negative_slope)  -- relu_layer.cu:33-43
 * applied where the gradient is produced instead of in a pass of its own (the caller then hands the result to the layer in front as an
 * already masked top_diff).  Transposed-convolution route only (the stride-2 convolutions: conv2 / conv3 of the encoders); ask _supported. */
int fn2_conv_backward_data_masked_supported(const fn2_conv_desc* desc, int transposed, int route);
int fn2_conv_backward_data_masked(const fn2_conv_desc* desc, int transposed, int route, const float* top_diff, int top_channels, int top_c0,
                                  const float* packed_weight, float* bottom_diff, int bottom_channels, int bottom_c0,
                                  const float* bottom_data, int data_channels, int data_c0, float negative_slope, void* stream);
int fn2_conv_backward_weights_supported(const fn2_conv_desc* desc, int transposed);
size_t fn2_conv_backward_weights_workspace_bytes(const fn2_conv_desc* desc, int transposed);
int fn2_conv_backward_weights(const fn2_conv_desc* desc, int transposed, const float* bottom, int bottom_channels, int bottom_c0,
                              const float* top_diff, int top_channels, int top_c0, float* weight_diff, int accumulate,
                              void* workspace, size_t workspace_bytes, void* stream);
/* weight_diff and bias_diff of a layer from ONE pass over top_diff (round 6) where a kernel has that form -- the 7x7 / 2 stem: the kernel sums
 * the top_diff operand it feeds to the matrix pipe (weight_gpu_gemm + backward_gpu_bias, base_conv_layer.cpp:368-393); whole blobs,
 * workspace of fn2_conv_backward_weights_workspace_bytes.  _fused tells whether the layer has one. */
int fn2_conv_backward_weights_bias_fused(const fn2_conv_desc* desc, int transposed);
int fn2_conv_backward_weights_bias(const fn2_conv_desc* desc, int transposed, const float* bottom, const float* top_diff, float* weight_diff,
                                   float* bias_diff, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int fn2_conv_backward_bias(const float* top_diff, int diff_channels, int diff_c0, float* bias_diff, int N, int C, int H, int W,
                           int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Transposed convolution, stride 2 (fp32 MFMA, NCHW, no column matrix / col2im):
 *     top[n][top_c0 + co][Y][X] = act(bias[co] + sum_{ci,ky,kx: Y = 2y - pad + ky, X = 2x - pad + kx} bottom[n][bottom_c0 + ci][y][x] * W[ci][co][ky][kx])
 *   <- DeconvolutionLayer::Forward_gpu, src/caffe/layers/deconv_layer.cu:8-26 (backward_gpu_gemm + col2im_gpu, base_conv_layer.cpp:375-393,
 *      forward_gpu_bias) with weight blob [Cin, Cout, 4, 4], kernel 4 / pad 1, Hout = 2 Hin, + the in-place ReLU (relu_layer.cu:8-27);
 *   <- ConvolutionLayer::Backward_gpu's data gradient (conv_layer.cu:53-57 -> backward_gpu_gemm, base_conv_layer.cpp:352-366) of a
 *      stride-2 Convolution: bottom = top_diff, W = the layer's weight [Cout_conv, Cin_conv, k, k] (already [in][out][k][k]), top =
 *      bottom_diff of size Hout x Wout (the layer's input size), kernel / pad 5 / 2 or 3 / 1, bias NULL, relu 0.
 *   packed_weight = fn2_conv_mfma_pack_weights of the [Cout][Cin][k][k] VIEW of W (its first two axes swapped), Cout % 64 == 0,
 *   Win % 4 == 0.  Same bits from every tile variant (fn2_debug_set_tconv_variant).  No workspace. */
int fn2_tconv_supported(int Cin, int Hin, int Win, int Cout, int Hout, int Wout, int kernel, int pad);
int fn2_tconv_forward(const float* bottom, const float* packed_weight, const float* bias, float* top,
                      int N, int Cin, int Hin, int Win, int bottom_channels, int bottom_c0,
                      int Cout, int Hout, int Wout, int top_channels, int top_c0, int kernel, int pad,
                      int relu, float negative_slope, void* stream);
int fn2_tconv_num_variants(void);
int fn2_debug_set_tconv_variant(int variant);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient of a Convolution or Deconvolution layer (fp32 MFMA, NCHW, deterministic):
 *     dw[ca][cb][ky][kx] (+)= sum_{n,y,x} a[n][a_c0 + ca][y][x] * b[n][b_c0 + cb][stride*y + ky - pad][stride*x + kx - pad]   (0 outside b)
 *   <- ConvolutionLayer::Backward_gpu, src/caffe/layers/conv_layer.cu:40-52 (weight_gpu_gemm, base_conv_layer.cpp:368-384: per sample
 *      im2col_gpu + cublasSgemm with beta = 1): a = top_diff [N, Cout, Hout, Wout], b = bottom, dw = weight_diff [Cout][Cin][k][k];
 *   <- DeconvolutionLayer::Backward_gpu, deconv_layer.cu:36-50 (weight_gpu_gemm(top_diff, bottom): the roles swapped): a = bottom
 *      [N, Cin, Hin, Win], b = top_diff, dw = weight_diff [Cin][Cout][k][k].
 *   accumulate != 0 adds into dw like the reference (the solver clears the diffs once per iteration); 0 overwrites.
 *   kernel / stride classes: 1/1, 3/1, 3/2, 4/2, 5/2; pad <= kernel - 1.  Summation order: the rows (n, y) of `a` are cut into
 *   fn2_conv_wgrad_ksplit(...) contiguous parts; per part one fma chain over its pixels in (n, y, x) order, the parts added in part
 *   order -- a function of the layer geometry only, restated by the oracle twin.  Workspace: the parts' partial sums (+ width-padded
 *   copies of maps whose width is not a multiple of 4), fn2_conv_wgrad_workspace_bytes.  The bias gradient: fn2_bias_leaky_relu_backward. */
int fn2_debug_set_wgrad_buffers(int two_buffers);   /* 0 / 1: force the one-buffer (2 workgroups per CU) / two-buffer kernels; -1: default.  Same bits. */
int fn2_debug_set_wgrad_chunk(int pixels);          /* force the chunk width (8, 16, 28, 56) where it applies; 0: by row width.  Same bits. */
int fn2_conv_wgrad_supported(int N, int Ca, int Ha, int Wa, int Cb, int Hb, int Wb, int kernel, int stride, int pad);
int fn2_conv_wgrad_ksplit(int N, int Ca, int Ha, int Wa, int Cb, int Hb, int Wb, int kernel, int stride, int pad);
size_t fn2_conv_wgrad_workspace_bytes(int N, int Ca, int Ha, int Wa, int Cb, int Hb, int Wb, int kernel, int stride, int pad);
int fn2_conv_wgrad(const float* a, const float* b, float* dw,
                   int N, int Ca, int Ha, int Wa, int a_channels, int a_c0,
                   int Cb, int Hb, int Wb, int b_channels, int b_c0,
                   int kernel, int stride, int pad, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * im2col / col2im of Caffe's GEMM convolution, batched over the mini-batch (square kernel, no dilation):
 *   fn2_im2col_forward            <- im2col_gpu, src/caffe/util/im2col.cu:8-72, as used by
 *                                    BaseConvolutionLayer::forward_gpu_gemm (base_conv_layer.cpp:325-341)
 *       col[n][(c*k + i)*k + j][yc*Wc + xc] = im[n][c][yc*stride - pad + i][xc*stride - pad + j]   (0 outside the image)
 *       Hc = (H + 2*pad - k) / stride + 1, Wc likewise.  The caller multiplies weight[Cout, C*k*k] with col[n].
 *   fn2_col2im_bias_relu_forward  <- col2im_gpu, im2col.cu:246-318, as used by the Deconvolution forward
 *                                    (backward_gpu_gemm, base_conv_layer.cpp:352-368; deconv_layer.cu:8-23), fused with
 *                                    forward_gpu_bias (:343-348) and, if apply_relu, the in-place ReLU (relu_layer.cu:8-14)
 *       im[n][c][y][x] = f(bias[c] + sum of the col entries that map onto it), col = weight^T[C*k*k, Cin] x bottom[n];
 *       (H, W) is the IMAGE (= deconvolution output) size; the column grid is Hc x Wc as above.
 * ---------------------------------------------------------------------------------------------- */
int fn2_im2col_forward(const float* im, float* col, int N, int C, int H, int W, int kernel, int pad, int stride, void* stream);
int fn2_col2im_bias_relu_forward(const float* col, const float* bias, float* im, int N, int C, int H, int W,
                                 int kernel, int pad, int stride, int apply_relu, float negative_slope, void* stream);
/* the same pass written into channels [top_c0, top_c0 + C) of a wider top blob [N, top_channels, H, W] (the deconvolution is an input
 * of a refinement Concat, concat_layer.cu:8-52) */
int fn2_col2im_bias_relu_forward_into(const float* col, const float* bias, float* top, int N, int C, int H, int W,
                                      int kernel, int pad, int stride, int apply_relu, float negative_slope,
                                      int top_channels, int top_c0, void* stream);

/* ------------------------------------------------------------------------------------------------
 * .caffemodel reader (host code): the trained blobs of a serialized NetParameter
 *   <- Net::CopyTrainedLayersFromBinaryProto / CopyTrainedLayersFrom, src/caffe/net.cpp:752-819 (layers matched BY NAME, blobs by
 *      index, shapes CHECKed) and Blob::FromProto, src/caffe/blob.cpp:459-508 (shape message or the legacy 4-D fields; double_data
 *      wins over data).  Layers with DoesUseCustomCopyBlobs (DataAugmentation: blobs 0 = iteration count, 1 = per-pixel mean,
 *      2 = per-channel mean, data_augmentation_layer.cpp:162-205) come out like any other layer; the caller picks blob 2.
 *   fn2_caffemodel_index fills up to max_entries entries (one per BlobProto, file order) and reports the total in *num_entries
 *   (call with max_entries 0 to size the array).  Offsets are byte offsets into the caller's buffer (names are not copied).
 *   fn2_caffemodel_read_blob copies one blob as floats (count elements).  Both the current `layer` (field 100) and the deprecated
 *   V1 `layers` (field 2) lists are read; V1 entries carry the LayerType enum in v1_type instead of a type string.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fn2_caffemodel_entry {
  size_t name_off, name_len;      /* layer name (LayerParameter.name) */
  size_t type_off, type_len;      /* layer type string (empty for V1 layers) */
  long long v1_type;              /* V1LayerParameter.type enum value, -1 otherwise */
  int v1;                         /* 1: entry of the deprecated `layers` list */
  int blob_index;                 /* index in the layer's blobs */
  int num_axes;
  long long dim[8];
  size_t count;                   /* elements of data (or double_data) */
  int is_double;                  /* stored as double_data */
  size_t blob_off, blob_len;      /* the BlobProto message */
} fn2_caffemodel_entry;
int fn2_caffemodel_index(const void* buf, size_t len, fn2_caffemodel_entry* entries, int max_entries, int* num_entries);
int fn2_caffemodel_read_blob(const void* buf, size_t len, const fn2_caffemodel_entry* entry, float* dst, size_t dst_floats);

/* ------------------------------------------------------------------------------------------------
 * .caffemodel.h5 reader (host code): the datasets of an HDF5 weight file
 *   <- Net::CopyTrainedLayersFrom dispatches on the ".h5" suffix (src/caffe/net.cpp:804-811) to CopyTrainedLayersFromHDF5
 *      (net.cpp:823-882): group "data", one group per layer name, datasets "0", "1", ... read by hdf5_load_nd_dataset
 *      (src/caffe/util/hdf5.cpp:9-79: H5T_FLOAT / H5T_INTEGER accepted, converted to float; the blob takes the dataset's shape).
 *      Net::ToHDF5 (net.cpp:896-950) writes that layout with H5LTmake_dataset_float (util/hdf5.cpp:81-101).
 *   A native walk of the file format (the image has no libhdf5): superblock v0-v3, object headers v1 / v2, old-style groups
 *   (v1 B-tree + local heap) and compact new-style groups, contiguous / compact / chunked layouts, deflate + shuffle filters,
 *   IEEE float32 / float64 and 1-8 byte integers of either byte order.  Dense link storage (libver=latest, > 8 links) is refused.
 *   fn2_hdf5_index lists every dataset of the file (absolute path, name order inside a group = H5_INDEX_NAME order,
 *   util/hdf5.cpp:168-181) -- call with max_entries 0 to size the array; fn2_hdf5_read_float copies one dataset converted to
 *   float (dst_floats must equal its element count).
 * ---------------------------------------------------------------------------------------------- */
typedef struct fn2_hdf5_entry {
  char path[256];                 /* "/data/conv1/0", NUL-terminated */
  int num_axes;
  long long dim[8];
  size_t count;                   /* elements */
  int type_class;                 /* 0 = H5T_INTEGER, 1 = H5T_FLOAT, other classes are listed but not readable */
  int type_size;                  /* bytes per element in the file */
  int type_signed;
  int big_endian;
  int layout;                     /* 0 compact, 1 contiguous, 2 chunked */
  int num_filters;
  size_t header_off;              /* the dataset's object header in the caller's buffer */
} fn2_hdf5_entry;
int fn2_hdf5_index(const void* buf, size_t len, fn2_hdf5_entry* entries, int max_entries, int* num_entries);
int fn2_hdf5_read_float(const void* buf, size_t len, const fn2_hdf5_entry* entry, float* dst, size_t dst_floats);

/* ------------------------------------------------------------------------------------------------
 * CustomData sample format  (type: "CustomData"; SURVEY.md 8f row 4: the on-disk format of the training sets)
 *   An LMDB value is a serialized `Datum` (src/caffe/proto/caffe.proto:30-41) whose `data` bytes hold, plane after plane,
 *   the slices named by DataParameter.slice_point / .encoding (caffe.proto:923-927, :979-980).  The FlyingChairs sets written by
 *   tools/convert_imageset_and_flow.cpp:142-206 have channels = 9, slice_point 3,6,8, encoding UINT8,UINT8,UINT16FLOW,BOOL1:
 *     image 0, image 1 : [3,H,W] uint8, planar, OpenCV channel order (B,G,R)
 *     flow             : [2,H,W] int16 little-endian = (short)(flow * 32) (C conversion: toward zero); SHRT_MAX marks NaN
 *     occlusions       : H*W bits, LSB first, (H*W - 1)/8 + 1 bytes
 *   decode  <- DecodeData, src/caffe/layers/custom_data_layer.cpp:44-136, followed by the per-slice copy of
 *              CustomDataLayerPrefetch (:209-300): top = (decoded - mean[data_index]) * scale.  The reference does this on ONE host
 *              thread and uploads the fp32 blobs (Forward_gpu = Forward_cpu, custom_data_layer.cu:19-23); here the raw bytes are
 *              uploaded (3.6x fewer) and decoded by a kernel.  crop_size > 0 is LOG(FATAL) in the reference (:503-506): no crop/mirror.
 *   The storage engine (LMDB's B-tree file) is out of scope: records arrive as (pointer, length) from whatever reads them.
 * ---------------------------------------------------------------------------------------------- */
enum { FN2_ENC_UINT8 = 1, FN2_ENC_UINT16FLOW = 2, FN2_ENC_BOOL1 = 3 };   /* DataParameter.CHANNELENCODING */

typedef struct fn2_datum_view {     /* Datum, caffe.proto:30-41; pointers point INTO the parsed buffer */
  int channels, height, width;      /* fields 1-3 (0 when absent)                                   */
  int label;                        /* field 5                                                      */
  int encoded;                      /* field 7                                                      */
  const unsigned char* data;        /* field 4, NULL when absent                                    */
  size_t data_bytes;
  size_t float_data_count;          /* field 6 (repeated float, packed or not); fetch with fn2_datum_float_data */
} fn2_datum_view;

/* HOST functions (no GPU work).  Protobuf wire format, proto2: unknown fields are skipped, the last occurrence of a scalar wins. */
int fn2_datum_parse(const void* buf, size_t len, fn2_datum_view* out);
int fn2_datum_float_data(const void* buf, size_t len, float* dst, size_t count);
/* Serializes {channels, height, width, data, label} the way Datum::SerializeToString does for the writer tool
 * (convert_imageset_and_flow.cpp:231-236: fields in number order, all five present).  Returns the size; writes when dst != NULL
 * and dst_bytes suffices, else FN2_ERR_WORKSPACE (as a negative size). */
long long fn2_datum_serialize(int channels, int height, int width, const void* data, size_t data_bytes, int label,
                              void* dst, size_t dst_bytes);

/* Bytes of one sample's `data` for the given slicing; 0 if the slicing is invalid (slice points not increasing / beyond channels,
 * unknown encoding, BOOL1 slice with more than one channel -- DecodeData walks H*W bits once per BOOL1 slice, :113-128). */
size_t fn2_custom_data_sample_bytes(int channels, int H, int W, const int* slice_points, int n_slice_points,
                                    const int* encodings, int n_encodings);
/* HOST: the writer (ImagePair::read_data, tools/convert_imageset_and_flow.cpp:142-206).  img0/img1: [H,W,3] uint8 interleaved as
 * cv::imread returns them; flow: [2,H,W] float planar as readFloFile returns it (util/output.cpp:31-37), NULL = zeros;
 * occlusion: [H,W] uint8 (non-zero = occluded), NULL = none.  dst must hold 10*H*W + (H*W-1)/8 + 1 bytes. */
int fn2_custom_data_encode_sample(const unsigned char* img0_hwc, const unsigned char* img1_hwc, const float* flow_chw,
                                  const unsigned char* occlusion, int H, int W, unsigned char* dst, size_t dst_bytes);
/* HOST: the staging step of a batch -- walks N serialized Datums (LMDB values), CHECKs that they share channels / height / width and
 * payload size (custom_data_layer.cpp:545 assumes it), copies every `data` payload to staging + i * sample_stride (a page-locked buffer
 * the caller uploads with one copy) and returns the labels (Datum.label, :297).  staging == NULL: only reports the shape and
 * *sample_bytes of the first record, so that the caller can size the buffer.  Records without `data` bytes (float_data datums) are
 * refused here. */
int fn2_custom_data_stage_records(const void* const* records, const size_t* record_bytes, int N, void* staging, size_t sample_stride,
                                  int* channels, int* height, int* width, size_t* sample_bytes, int* labels);
/* DEVICE: samples = N `data` payloads in device memory, sample_stride bytes apart (>= the sample size).  mean: device
 * [channels*H*W] floats or NULL (= 0, data_mean_ without mean_file / subtract, :612-615).  tops: HOST array of n_slice_points + 1
 * device pointers, top[s] = [N, slice channels, H, W].  float_data != 0: samples are channels*H*W floats each (Datum.float_data,
 * :53-61; encodings must then be empty, CHECK :55).  NaN flow decodes to a quiet NaN (the reference's signaling NaN is quieted by
 * the mean subtraction, :104-108, :282). */
int fn2_custom_data_decode_forward(const void* samples, size_t sample_stride, int N, int channels, int H, int W,
                                   const int* slice_points, int n_slice_points, const int* encodings, int n_encodings,
                                   int float_data, const float* mean, float scale, float* const* tops, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FlowAugmentation  (type: "FlowAugmentation"; SURVEY.md 8f row 3: the ground-truth flow under the spatial augmentation of both images)
 *   coefficient arrays <- AugmentationLayerBase::coeff_to_array / array_to_coeff, src/caffe/layers/augmentation_layer_base.cpp:352-380:
 *                         one AugmentationCoeff (caffe.proto:436-486, 42 float fields) per sample, index = declaration order, fields
 *                         with a non-zero default stored as log(value); only mirror, dx, dy, angle, zoom_x, zoom_y (0..5) matter here
 *   matrix             <- tTransMat::toIdentity / leftMultiply / fromCoeff / inverse, augmentation_layer_base.cpp:14-68
 *                         ( t0 t2 t4 ; t1 t3 t5 ), crop-centred mirror, rotation, translation by (dx*crop_w, dy*crop_h), 1/zoom, bottom-centred
 *   forward            <- FlowAugmentationLayer::Forward_gpu + WarpData, src/caffe/layers/flow_augmentation_layer.cu:23-88, :92-160:
 *                         p1 = M1 (x,y); f = flow at round-half-up(p1) (flat index, no bounds check); p3 = M2^-1 (p1 + f); top = p3 - (x,y)
 *   shapes             <- FlowAugmentationLayer::Reshape, flow_augmentation_layer.cpp:40-72: top [N,2,crop_height,crop_width]
 * The coefficient blobs are read on the HOST by the reference (cpu_data(), :123-124); they are host pointers here as well.
 * A flat source index outside the flow blob (the reference clamps only from above, to one element PAST the end, :50-56) reads 0.
 * ---------------------------------------------------------------------------------------------- */
#define FN2_AUG_NUM_PARAMS 42
/* HOST.  mat6 = {t0, t1, t2, t3, t4, t5}; invert != 0 returns tTransMat::inverse() of it (what the layer does for image 2). */
int fn2_augmentation_matrix(const float* coeffs, int crop_width, int crop_height, int bottom_width, int bottom_height,
                            int invert, float* mat6);
int fn2_flow_augmentation_forward(const float* flow, const float* coeffs1_host, const float* coeffs2_host, float* top,
                                  int N, int H, int W, int crop_height, int crop_width, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DataAugmentation  (type: "DataAugmentation"; the image half of the augmentation, applied for GIVEN coefficients)
 *   forward <- DataAugmentationLayer::Forward_gpu, src/caffe/layers/data_augmentation_layer.cu:320-637, the part after the coefficient
 *              blob exists (:452-637): per sample array_to_coeff + clear_defaults (a field within 1e-3 of its default is dropped,
 *              augmentation_layer_base.cpp:340-350) -> tTransMat / tChromaticCoeffs / tChromaticEigenCoeffs / tEffectCoeffs, then
 *                SpatialAugmentation        :24-69   bilinear sample of the source at M (x, y), clamped to [0, W-1.05] x [0, H-1.05]
 *                ChromaticEigenAugmentation :192-291 (with ComputeChromaticEigenspace :147-187 over the SOURCE batch)
 *                ColorContrastAugmentation  :72-116  colour, brightness compensation, gamma, brightness, contrast
 *                ApplyEffects               :295-317 half-plane shadow
 *              and the mean subtraction :592-635 (per pixel :613-616, per channel :617-634).  One kernel does all of it per pixel.
 *   shapes  <- DataAugmentationLayer::Reshape, data_augmentation_layer.cpp:74-160: top [N,C,crop_height,crop_width]; without a crop size
 *              the layer copies the bottom (:590) and only the mean is applied.
 * Not part of this entry point: drawing the coefficients (generate_*_coeffs use boost generators: their stream cannot be reproduced),
 * fog / motion blur (coefficients exist, the reference has no kernel for them either), the running re-computation of the mean over
 * the first iterations (:597-606; state of the layer object: flownet2_amd.layers.DataAugmentationLayer keeps it -- pass the mean in).
 * The noise effect (:578-587: N(0, noise^2) per element from cuRAND's stream, which cannot be reproduced) draws from Philox4x32-10
 * keyed by noise_seed, counter (pixel, sample, channel triple, noise_stream): reproducible from the parameters alone; pass the
 * iteration number as noise_stream so that every step draws fresh noise.  Colour transforms need 3 channels (CHECKs :489,:540,:548,:556).
 * ---------------------------------------------------------------------------------------------- */
enum { FN2_MEAN_NONE = 0, FN2_MEAN_PER_CHANNEL = 1, FN2_MEAN_PER_PIXEL = 2 };
typedef struct fn2_data_aug_params {
  int crop_width, crop_height;     /* AugmentationParameter.crop_width = 33 / crop_height = 34; both 0 = no cropping, no augmentation */
  float max_multiplier;            /* max_multiplier = 3 [default 255] */
  int has_chromatic_eigvec;        /* chromatic_eigvec = 83 (9 floats), needed when a sample has chromatic-eigen coefficients */
  float chromatic_eigvec[9];
  int mean_mode;                   /* FN2_MEAN_*: what `mean` holds: C floats, or C*crop_height*crop_width floats */
  unsigned long long noise_seed;   /* key of the counter-based generator behind the noise effect (ours: the reference uses cuRAND's global stream) */
  unsigned long long noise_stream; /* high counter words: the layer's iteration count */
} fn2_data_aug_params;
/* Device scratch for the chromatic-eigen statistics of the batch. */
size_t fn2_data_augmentation_workspace_bytes(int N);
/* bottom [N,C,H,W] device; coeffs_host [N,42] HOST (coeff_to_array layout) or NULL = all defaults; mean device or NULL. */
int fn2_data_augmentation_forward(const fn2_data_aug_params* p, const float* bottom, const float* coeffs_host, const float* mean,
                                  float* top, int N, int C, int H, int W, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* FLOWNET2_HIP_H_ */
