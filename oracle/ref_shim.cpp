// C API over the REFERENCE's own layer classes (compiled in place from /root/reference by oracle/ref_build.sh
// against the stand-in Caffe headers).  TEST INFRASTRUCTURE: the resulting oracle/_ref/libfn2_ref.so runs the
// reference's CUDA kernels -- built as HIP, unchanged -- on the MI355X, and pins the C oracle (tests/test_ref_pin.py).
// All pointers are HOST pointers; layers are created through the reference's LayerRegistry by their type string.
#include <cstring>

#include "caffe/blob.hpp"
#include "caffe/layer.hpp"
#include "caffe/layer_factory.hpp"

using namespace caffe;

static thread_local std::string g_err;

template <typename F>
static int guard(F&& body) {
  try { body(); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
}

extern "C" __attribute__((visibility("default"))) const char* fn2ref_last_error() { return g_err.c_str(); }

static void fill(Blob<float>& b, const float* src) { std::memcpy(b.mutable_cpu_data(), src, sizeof(float) * b.count()); }
static void fill_diff(Blob<float>& b, const float* src) { std::memcpy(b.mutable_cpu_diff(), src, sizeof(float) * b.count()); }
static void fetch(const Blob<float>& b, float* dst) { std::memcpy(dst, b.cpu_data(), sizeof(float) * b.count()); }
static void fetch_diff(const Blob<float>& b, float* dst) { std::memcpy(dst, b.cpu_diff(), sizeof(float) * b.count()); }

extern "C" __attribute__((visibility("default")))
int fn2ref_correlation(int pad, int kernel_size, int max_displacement, int stride1, int stride2, int corr_type,
                       const float* b0, const float* b1, int N, int C, int H, int W,
                       float* top_out, int* top_shape /* [4] */,
                       const float* top_diff /* nullable */, float* b0_diff, float* b1_diff) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_type("Correlation");
    CorrelationParameter* cp = lp.mutable_correlation_param();
    cp->set_pad(pad); cp->set_kernel_size(kernel_size); cp->set_max_displacement(max_displacement);
    cp->set_stride_1(stride1); cp->set_stride_2(stride2);
    cp->set_correlation_type(corr_type ? CorrelationParameter_CorrelationType_SUBTRACT : CorrelationParameter_CorrelationType_MULTIPLY);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> bot0(N, C, H, W), bot1(N, C, H, W), top;
    fill(bot0, b0); fill(bot1, b1);
    vector<Blob<float>*> bottom{&bot0, &bot1}, tops{&top};
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) top_shape[i] = top.shape(i);
    if (top_out) fetch(top, top_out);
    if (top_diff) {
      fill_diff(top, top_diff);
      layer->Backward(tops, vector<bool>{true, true}, bottom);
      if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipDeviceSynchronize());
      fetch_diff(bot0, b0_diff); fetch_diff(bot1, b1_diff);
    }
  });
}

extern "C" __attribute__((visibility("default")))
int fn2ref_correlation1d(int pad, int kernel_size, int max_displacement, int stride1, int stride2, int corr_type, int single_direction,
                         const float* b0, const float* b1, int N, int C, int H, int W,
                         float* top_out, int* top_shape /* [4] */,
                         const float* top_diff /* nullable */, float* b0_diff, float* b1_diff) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_type("Correlation1D");
    CorrelationParameter* cp = lp.mutable_correlation_param();
    cp->set_pad(pad); cp->set_kernel_size(kernel_size); cp->set_max_displacement(max_displacement);
    cp->set_stride_1(stride1); cp->set_stride_2(stride2); cp->set_single_direction(single_direction);
    cp->set_correlation_type(corr_type ? CorrelationParameter_CorrelationType_SUBTRACT : CorrelationParameter_CorrelationType_MULTIPLY);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> bot0(N, C, H, W), bot1(N, C, H, W), top;
    fill(bot0, b0); fill(bot1, b1);
    vector<Blob<float>*> bottom{&bot0, &bot1}, tops{&top};
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    CUDA_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) top_shape[i] = top.shape(i);
    if (top_out) fetch(top, top_out);
    if (top_diff) {
      fill_diff(top, top_diff);
      layer->Backward(tops, vector<bool>{true, true}, bottom);
      CUDA_CHECK(hipDeviceSynchronize());
      fetch_diff(bot0, b0_diff); fetch_diff(bot1, b1_diff);
    }
  });
}

// FlowAugmentation through the registry: bottom = {flow, coefficient blob of image 1, coefficient blob of image 2}
extern "C" __attribute__((visibility("default")))
int fn2ref_flow_augmentation(const float* flow, const float* coeffs1, const float* coeffs2, int num_params, int N, int H, int W,
                             int crop_height, int crop_width, float* top_out) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_type("FlowAugmentation");
    lp.mutable_augmentation_param()->set_crop_width(crop_width);
    lp.mutable_augmentation_param()->set_crop_height(crop_height);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> fl(N, 2, H, W), c1(N, num_params, 1, 1), c2(N, num_params, 1, 1), top;
    fill(fl, flow); fill(c1, coeffs1); fill(c2, coeffs2);
    vector<Blob<float>*> bottom{&fl, &c1, &c2}, tops{&top};
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    CUDA_CHECK(hipDeviceSynchronize());
    CHECK_EQ(top.count(), N * 2 * crop_height * crop_width);
    fetch(top, top_out);
  });
}

// DataAugmentation with the coefficients given as bottom[1] (input_params_, data_augmentation_layer.cpp:87): the deterministic part of
// the layer.  mean3: per-channel mean from the proto (AugmentationParameter.mean, mean_per_pixel = false) or NULL.
extern "C" __attribute__((visibility("default")))
int fn2ref_data_augmentation(const float* bottom0, const float* coeffs /* nullable */, int num_params, int N, int C, int H, int W,
                             int crop_height, int crop_width, float max_multiplier, const float* chromatic_eigvec /* [9] or NULL */,
                             const float* mean3 /* [3] or NULL */, float* top_out) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_type("DataAugmentation");
    AugmentationParameter* ap = lp.mutable_augmentation_param();
    if (crop_width > 0 && crop_height > 0) { ap->set_crop_width(crop_width); ap->set_crop_height(crop_height); }
    ap->set_max_multiplier(max_multiplier);
    if (chromatic_eigvec) for (int i = 0; i < 9; ++i) ap->add_chromatic_eigvec(chromatic_eigvec[i]);
    if (mean3) { for (int i = 0; i < 3; ++i) ap->add_mean(mean3[i]); ap->set_mean_per_pixel(false); }
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> img(N, C, H, W), co(N, num_params, 1, 1), top;
    fill(img, bottom0);
    vector<Blob<float>*> bottom{&img}, tops{&top};
    if (coeffs) { fill(co, coeffs); bottom.push_back(&co); }
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    CUDA_CHECK(hipDeviceSynchronize());
    fetch(top, top_out);
  });
}

// mode: 0 = GPU kernels (flow_warp_layer.cu), 1 = the reference's CPU implementation (flow_warp_layer.cpp:58-199)
extern "C" __attribute__((visibility("default")))
int fn2ref_flow_warp(int mode, int fill_value, const float* image, const float* flow, int N, int C, int H, int W,
                     float* warped, const float* warped_diff /* nullable */, float* image_diff, float* flow_diff) {
  return guard([&] {
    Caffe::set_mode(mode ? Caffe::CPU : Caffe::GPU);
    LayerParameter lp;
    lp.set_type("FlowWarp");
    lp.mutable_flow_warp_param()->set_fill_value(fill_value == 2 ? FlowWarpParameter_FillParameter_NOT_A_NUMBER : FlowWarpParameter_FillParameter_ZERO);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> img(N, C, H, W), fl(N, 2, H, W), top;
    fill(img, image); fill(fl, flow);
    vector<Blob<float>*> bottom{&img, &fl}, tops{&top};
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipDeviceSynchronize());
    fetch(top, warped);
    if (warped_diff) {
      fill_diff(top, warped_diff);
      layer->Backward(tops, vector<bool>{true, true}, bottom);
      if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipDeviceSynchronize());
      fetch_diff(img, image_diff); fetch_diff(fl, flow_diff);
    }
    Caffe::set_mode(Caffe::GPU);
  });
}

extern "C" __attribute__((visibility("default")))
int fn2ref_resample(int type, int antialias, const float* in, int N, int C, int Hin, int Win, int Hout, int Wout, float* out) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_type("Resample");
    ResampleParameter* rp = lp.mutable_resample_param();
    rp->set_type((ResampleParameter_ResampleType)type); rp->set_antialias(antialias != 0); rp->set_width(Wout); rp->set_height(Hout);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> bot(N, C, Hin, Win), top;
    fill(bot, in);
    vector<Blob<float>*> bottom{&bot}, tops{&top};
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipDeviceSynchronize());
    fetch(top, out);
  });
}

// mode: 0 = GPU forward (channel_norm_layer.cu), 1 = the reference's CPU forward + backward (channel_norm_layer.cpp:43-124).
// The reference's GPU backward passes host pointers to its kernel (channel_norm_layer.cu:80-83) and is not run.
extern "C" __attribute__((visibility("default")))
int fn2ref_channel_norm(int mode, const float* in, int N, int C, int H, int W, float* out,
                        const float* top_diff /* nullable, CPU mode only */, float* bottom_diff) {
  return guard([&] {
    Caffe::set_mode(mode ? Caffe::CPU : Caffe::GPU);
    LayerParameter lp;
    lp.set_type("ChannelNorm");
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> bot(N, C, H, W), top;
    fill(bot, in);
    vector<Blob<float>*> bottom{&bot}, tops{&top};
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipDeviceSynchronize());
    fetch(top, out);
    if (top_diff && mode) {
      fill_diff(top, top_diff);
      layer->Backward(tops, vector<bool>{true}, bottom);
      fetch_diff(bot, bottom_diff);
    }
    Caffe::set_mode(Caffe::GPU);
  });
}

extern "C" __attribute__((visibility("default")))
int fn2ref_downsample(const float* in, int N, int C, int Hin, int Win, int Hout, int Wout, float* out) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_type("Downsample");
    lp.mutable_downsample_param()->set_top_height(Hout); lp.mutable_downsample_param()->set_top_width(Wout);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> bot(N, C, Hin, Win), top;
    fill(bot, in);
    vector<Blob<float>*> bottom{&bot}, tops{&top};
    layer->SetUp(bottom, tops);
    layer->Forward(bottom, tops);
    if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipDeviceSynchronize());
    fetch(top, out);
  });
}


// Forward + Backward of a Convolution / Deconvolution layer object: the parameter diffs start from the given values (the reference ACCUMULATES
// into them: weight_gpu_gemm / backward_gpu_bias with beta = 1), bottom_diff is overwritten.
static void conv_fwd_bwd(Layer<float>& layer, const float* x, int N, int C, int H, int W, const float* weight, const float* bias,
                         const float* top_diff, const float* weight_diff0, const float* bias_diff0,
                         float* bottom_diff, float* weight_diff, float* bias_diff) {
  Blob<float> bot(N, C, H, W), top;
  fill(bot, x);
  vector<Blob<float>*> bottom{&bot}, tops{&top};
  layer.SetUp(bottom, tops);
  fill(*layer.blobs()[0], weight);
  fill_diff(*layer.blobs()[0], weight_diff0);
  if (bias) { fill(*layer.blobs()[1], bias); fill_diff(*layer.blobs()[1], bias_diff0); }
  layer.Forward(bottom, tops);
  fill_diff(top, top_diff);
  layer.Backward(tops, vector<bool>{true}, bottom);
  CUDA_CHECK(hipDeviceSynchronize());
  fetch_diff(bot, bottom_diff);
  fetch_diff(*layer.blobs()[0], weight_diff);
  if (bias) fetch_diff(*layer.blobs()[1], bias_diff);
}

static void conv_param(LayerParameter& lp, int kernel, int stride, int pad, int num_output, bool bias) {
  ConvolutionParameter* cp = lp.mutable_convolution_param();
  cp->set_num_output(num_output); cp->add_kernel_size(kernel); cp->add_stride(stride); cp->add_pad(pad);
  cp->set_bias_term(bias);
  cp->mutable_weight_filler()->set_type("constant");
  cp->mutable_bias_filler()->set_type("constant");
}

#ifdef FN2_SHIM_STOCK
#include "caffe/layers/conv_layer.hpp"
#include "caffe/layers/deconv_layer.hpp"
#include "caffe/layers/relu_layer.hpp"

// Stock Convolution / Deconvolution (square kernel, given weights and optional bias), optionally followed by the in-place
// ReLU{negative_slope}: the reference's own conv_layer / deconv_layer / base_conv_layer / im2col / relu_layer sources; the
// SGEMM underneath is the plain stand-in of oracle/ref_compat.  Pins the stock-layer fast paths of libflownet2_hip.so.
extern "C" __attribute__((visibility("default")))
int fn2ref_convolution(int deconv, int kernel, int stride, int pad, int num_output, int relu, float negative_slope,
                       const float* x, int N, int C, int H, int W, const float* weight, const float* bias /* nullable */,
                       float* out /* nullable */, int* out_shape /* [4] */) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    ConvolutionParameter* cp = lp.mutable_convolution_param();
    cp->set_num_output(num_output); cp->add_kernel_size(kernel); cp->add_stride(stride); cp->add_pad(pad);
    cp->set_bias_term(bias != nullptr);
    cp->mutable_weight_filler()->set_type("constant");
    cp->mutable_bias_filler()->set_type("constant");
    shared_ptr<Layer<float> > layer;
    if (deconv) layer.reset(new DeconvolutionLayer<float>(lp)); else layer.reset(new ConvolutionLayer<float>(lp));
    Blob<float> bot(N, C, H, W), top;
    fill(bot, x);
    vector<Blob<float>*> bottom{&bot}, tops{&top};
    layer->SetUp(bottom, tops);
    fill(*layer->blobs()[0], weight);
    if (bias) fill(*layer->blobs()[1], bias);
    layer->Forward(bottom, tops);
    if (relu) {
      LayerParameter rp;
      rp.mutable_relu_param()->set_negative_slope(negative_slope);
      ReLULayer<float> act(rp);
      act.SetUp(tops, tops);            // in place, like the prototxts
      act.Forward(tops, tops);
    }
    CUDA_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) out_shape[i] = top.shape(i);
    if (out) fetch(top, out);
  });
}
#endif

#ifdef FN2_SHIM_STOCK
// The reference's ConvolutionLayer / DeconvolutionLayer::Backward_gpu (conv_layer.cu:26-60, deconv_layer.cu:27-58, base_conv_layer.cpp:352-393).
extern "C" __attribute__((visibility("default")))
int fn2ref_convolution_backward(int deconv, int kernel, int stride, int pad, int num_output, const float* x, int N, int C, int H, int W,
                                const float* weight, const float* bias /* nullable */, const float* top_diff, const float* weight_diff0,
                                const float* bias_diff0, float* bottom_diff, float* weight_diff, float* bias_diff) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    conv_param(lp, kernel, stride, pad, num_output, bias != nullptr);
    shared_ptr<Layer<float> > layer;
    if (deconv) layer.reset(new DeconvolutionLayer<float>(lp)); else layer.reset(new ConvolutionLayer<float>(lp));
    conv_fwd_bwd(*layer, x, N, C, H, W, weight, bias, top_diff, weight_diff0, bias_diff0, bottom_diff, weight_diff, bias_diff);
  });
}
#endif

#ifdef FN2_SHIM_CONV_REGISTRY
// Convolution / Deconvolution created BY TYPE STRING through LayerRegistry (the adapter build: its plug-ins registered "Convolution" and
// "Deconvolution"): square kernel, given weights and optional bias.  Same argument list as fn2ref_convolution (relu must be 0).
extern "C" __attribute__((visibility("default")))
int fn2ref_convolution_by_registry(int deconv, int kernel, int stride, int pad, int num_output, int relu, float negative_slope,
                                   const float* x, int N, int C, int H, int W, const float* weight, const float* bias /* nullable */,
                                   float* out /* nullable */, int* out_shape /* [4] */) {
  return guard([&] {
    (void)negative_slope;
    CHECK(relu == 0) << "the registry-driven convolution shim has no ReLU";
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_name("conv_under_test");
    lp.set_type(deconv ? "Deconvolution" : "Convolution");
    ConvolutionParameter* cp = lp.mutable_convolution_param();
    cp->set_num_output(num_output); cp->add_kernel_size(kernel); cp->add_stride(stride); cp->add_pad(pad);
    cp->set_bias_term(bias != nullptr);
    cp->mutable_weight_filler()->set_type("constant");
    cp->mutable_bias_filler()->set_type("constant");
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> bot(N, C, H, W), top;
    fill(bot, x);
    vector<Blob<float>*> bottom{&bot}, tops{&top};
    layer->SetUp(bottom, tops);
    fill(*layer->blobs()[0], weight);
    if (bias) fill(*layer->blobs()[1], bias);
    layer->Forward(bottom, tops);
    CUDA_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) out_shape[i] = top.shape(i);
    if (out) fetch(top, out);
  });
}
// The same through LayerRegistry (the adapter's plug-ins): forward + Backward.
extern "C" __attribute__((visibility("default")))
int fn2ref_convolution_backward_by_registry(int deconv, int kernel, int stride, int pad, int num_output, const float* x, int N, int C, int H, int W,
                                            const float* weight, const float* bias /* nullable */, const float* top_diff, const float* weight_diff0,
                                            const float* bias_diff0, float* bottom_diff, float* weight_diff, float* bias_diff) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_name("conv_under_test");
    lp.set_type(deconv ? "Deconvolution" : "Convolution");
    conv_param(lp, kernel, stride, pad, num_output, bias != nullptr);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    conv_fwd_bwd(*layer, x, N, C, H, W, weight, bias, top_diff, weight_diff0, bias_diff0, bottom_diff, weight_diff, bias_diff);
  });
}
#endif

#ifdef FN2_SHIM_L1LOSS

// The reference's L1LossLayer (oracle/_ref: built with its stock sub-layers, see oracle/README.md) or the adapter's.
extern "C" __attribute__((visibility("default")))
int fn2ref_l1loss(int l2_per_location, int prescale, int normalize, float epsilon, float plateau, float loss_weight,
                  const float* b0, const float* b1 /* nullable */, int N, int C, int H, int W,
                  float* loss_out, float* weighted_loss_out, float* b0_diff, float* b1_diff) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    LayerParameter lp;
    lp.set_type("L1Loss");
    lp.add_loss_weight(loss_weight);
    L1LossParameter* p = lp.mutable_l1_loss_param();
    p->set_l2_per_location(l2_per_location); p->set_l2_prescale_by_channels(prescale); p->set_normalize_by_num_entries(normalize);
    p->set_epsilon(epsilon); p->set_plateau(plateau);
    shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
    Blob<float> bot0(N, C, H, W), bot1(N, C, H, W), top;
    fill(bot0, b0);
    vector<Blob<float>*> bottom{&bot0}, tops{&top};
    if (b1) { fill(bot1, b1); bottom.push_back(&bot1); }
    layer->SetUp(bottom, tops);
    const float total = layer->Forward(bottom, tops);
    CUDA_CHECK(hipDeviceSynchronize());
    *loss_out = top.cpu_data()[0];
    *weighted_loss_out = total;
    layer->Backward(tops, vector<bool>(bottom.size(), true), bottom);
    CUDA_CHECK(hipDeviceSynchronize());
    fetch_diff(bot0, b0_diff);
    if (b1) fetch_diff(bot1, b1_diff);
  });
}
#endif

#ifdef FN2_SHIM_DATA
// The reference's CustomDataLayer (custom_data_layer.cpp, compiled in place) over the in-memory LMDB stand-in: registers the given
// (key, value) records as a database, creates the layer through the registry, runs n_forward batches on the CPU path (the layer has
// no other: Forward_gpu calls Forward_cpu) and copies every top out.  tops_out[s] receives n_forward * batch * slice_channels * H * W
// floats; labels_out (nullable) n_forward * batch.
#include "lmdb.h"
extern "C" __attribute__((visibility("default")))
int fn2ref_custom_data(const char* const* keys, const unsigned char* const* values, const size_t* value_bytes, int n_records,
                       int batch_size, const int* slice_points, int n_slice_points, const int* encodings, int n_encodings,
                       float scale, const float* subtract, int n_subtract, int range_start, int range_end, int n_forward,
                       float* const* tops_out, float* labels_out, int* top_shapes /* [(n_slice_points + 1) * 4] */) {
  return guard([&] {
    static int counter = 0;
    const std::string source = "mem:" + std::to_string(counter++);
    fn2_fake_db& db = fn2_fake_lmdb_sources()[source];
    for (int i = 0; i < n_records; ++i) db[keys[i]] = std::string(reinterpret_cast<const char*>(values[i]), value_bytes[i]);
    Caffe::set_mode(Caffe::CPU);
    {
      LayerParameter lp;
      lp.set_type("CustomData");
      DataParameter* dp = lp.mutable_data_param();
      dp->set_source(source); dp->set_backend(DataParameter_DB_LMDB); dp->set_batch_size(batch_size); dp->set_scale(scale);
      dp->set_range_start(range_start); dp->set_range_end(range_end);
      for (int i = 0; i < n_slice_points; ++i) dp->add_slice_point(slice_points[i]);
      for (int i = 0; i < n_encodings; ++i) dp->add_encoding(encodings[i]);
      for (int i = 0; i < n_subtract; ++i) dp->add_subtract(subtract[i]);
      shared_ptr<Layer<float> > layer = LayerRegistry<float>::CreateLayer(lp);
      const int ntop = n_slice_points + 1 + (labels_out ? 1 : 0);
      std::vector<Blob<float> > blobs(ntop);
      vector<Blob<float>*> bottom, tops;
      for (auto& b : blobs) tops.push_back(&b);
      layer->SetUp(bottom, tops);
      for (int s = 0; s <= n_slice_points; ++s)
        for (int a = 0; a < 4; ++a) top_shapes[s * 4 + a] = blobs[s].shape(a);
      for (int f = 0; f < n_forward; ++f) {
        layer->Forward(bottom, tops);
        for (int s = 0; s <= n_slice_points; ++s)
          std::memcpy(tops_out[s] + (size_t)f * blobs[s].count(), blobs[s].cpu_data(), sizeof(float) * blobs[s].count());
        if (labels_out) std::memcpy(labels_out + (size_t)f * batch_size, blobs[n_slice_points + 1].cpu_data(), sizeof(float) * batch_size);
      }
    }   // the layer joins its prefetch thread and closes the "database" here
    fn2_fake_lmdb_sources().erase(source);
    Caffe::set_mode(Caffe::GPU);
  });
}
#endif

#ifdef FN2_SHIM_NET
// ------------------------------------------------------------------------------------------------------------------------------------------
// A FlowNetC deploy forward chained from LayerRegistry-created layers and timed like `caffe time` (tools/caffe.cpp:346-366: per-layer
// Timer around Layer::Forward over `iterations` passes, plus the total).  Adapter build only: "Convolution" / "Deconvolution" /
// "Correlation" resolve to the plug-ins of flownet2_amd/csrc/caffe_adapter/fn2_caffe_layers.cpp; "ReLU" and "Concat" are the REFERENCE's
// own layers (relu_layer / concat_layer compiled in place) -- what a Caffe tree with the plug-ins dropped in executes: separate in-place
// ReLU passes and Concat copies between the convolution kernels, which the fused graph (flownet2_amd/nets.py) does not have.
// The graph is the FlowNetC core (pre-processed image pair -> predict_flow2) with filler weights; values are not checked here (the layers'
// parity is tests/test_caffe_adapter.py), only finiteness of the output.  `use_cache` = 0 forces a weight repack in every forward (the
// round-5 behaviour) by touching the weight blobs' mutable pointers between passes.
#include <chrono>
#include <cmath>
#include <map>

#include "caffe/layers/relu_layer.hpp"
// "ReLU" is registered by layer_factory.cpp in the reference (GetReLULayer, layer_factory.cpp:145-167: the cuDNN / Caffe engine switch),
// not by relu_layer.cpp: the Caffe-engine branch of that creator, restated for the driver below
namespace caffe {
static shared_ptr<Layer<float> > Fn2ShimGetReLULayer(const LayerParameter& p) { return shared_ptr<Layer<float> >(new ReLULayer<float>(p)); }
static LayerRegisterer<float> g_fn2_shim_relu_creator("ReLU", Fn2ShimGetReLULayer);
}  // namespace caffe

namespace {
struct MiniNet {
  struct Step { std::string name; shared_ptr<Layer<float> > layer; vector<Blob<float>*> bottom, top; double ms = 0.0; };
  std::map<std::string, shared_ptr<Blob<float> > > blobs;
  vector<Step> steps;
  Blob<float>* blob(const std::string& n) {
    shared_ptr<Blob<float> >& b = blobs[n];
    if (!b) b.reset(new Blob<float>());
    return b.get();
  }
  void add(LayerParameter lp, const vector<std::string>& bottoms, const vector<std::string>& tops) {
    Step s;
    s.name = lp.name();
    lp.set_phase(TEST);
    s.layer = LayerRegistry<float>::CreateLayer(lp);
    for (const std::string& b : bottoms) s.bottom.push_back(blob(b));
    for (const std::string& t : tops) s.top.push_back(blob(t));
    s.layer->SetUp(s.bottom, s.top);
    steps.push_back(s);
  }
  void conv(const std::string& name, const std::string& in, const std::string& out, int k, int s, int p, int nout, bool deconv = false, bool relu = true) {
    LayerParameter lp;
    lp.set_name(name);
    lp.set_type(deconv ? "Deconvolution" : "Convolution");
    conv_param(lp, k, s, p, nout, true);
    add(lp, {in}, {out});
    // weights: a fixed pseudo-random pattern scaled like an MSRA filler (the stand-in headers carry the constant filler only)
    Blob<float>& wb = *steps.back().layer->blobs()[0];
    float* wp = wb.mutable_cpu_data();
    const float scale = std::sqrt(2.0f / (float)(wb.count() / wb.shape(0)));
    unsigned st = 12345u + (unsigned)steps.size();
    for (int i = 0; i < wb.count(); ++i) { st = st * 1664525u + 1013904223u; wp[i] = ((float)(st >> 8) / 8388608.0f - 1.0f) * scale; }
    if (relu) {
      LayerParameter rp;
      rp.set_name(name + "_relu");
      rp.set_type("ReLU");
      rp.mutable_relu_param()->set_negative_slope(0.1f);
      add(rp, {out}, {out});                // in place, like the prototxts
    }
  }
  void concat(const std::string& name, const vector<std::string>& in, const std::string& out) {
    LayerParameter lp;
    lp.set_name(name);
    lp.set_type("Concat");
    add(lp, in, {out});
  }
};
}  // namespace

extern "C" long long fn2_caffe_adapter_weight_packs();
extern "C" long long fn2_caffe_adapter_weight_pack_reuses();

extern "C" __attribute__((visibility("default")))
int fn2ref_flownetc_time(int N, int H, int W, int warmup, int iterations, int use_cache, double* total_ms_per_forward,
                         double* layer_ms /* [max_layers] */, char* layer_names /* max_layers x 48 */, int max_layers, int* num_layers,
                         long long* packs, long long* pack_reuses, int* output_finite) {
  return guard([&] {
    Caffe::set_mode(Caffe::GPU);
    MiniNet net;
    for (const char* t : {"a", "b"}) {
      Blob<float>* img = net.blob(std::string("img") + t);
      img->Reshape(N, 3, H, W);
      float* p = img->mutable_cpu_data();
      for (int i = 0; i < img->count(); ++i) p[i] = (float)((i * 2654435761u >> 8) & 255) / 255.f - 0.43f;
    }
    // siamese towers (the prototxts: two layer instances sharing weights by ParamSpec name; here two instances, same cost)
    for (const char* t : {"a", "b"}) {
      const std::string s(t);
      net.conv("conv1" + s, "img" + s, "conv1" + s, 7, 2, 3, 64);
      net.conv("conv2" + s, "conv1" + s, "conv2" + s, 5, 2, 2, 128);
      net.conv("conv3" + s, "conv2" + s, "conv3" + s, 5, 2, 2, 256);
    }
    {
      LayerParameter lp;
      lp.set_name("corr"); lp.set_type("Correlation");
      CorrelationParameter* cp = lp.mutable_correlation_param();
      cp->set_pad(20); cp->set_kernel_size(1); cp->set_max_displacement(20); cp->set_stride_1(1); cp->set_stride_2(2);
      net.add(lp, {"conv3a", "conv3b"}, {"corr"});
      LayerParameter rp;
      rp.set_name("corr_relu"); rp.set_type("ReLU"); rp.mutable_relu_param()->set_negative_slope(0.1f);
      net.add(rp, {"corr"}, {"corr"});
    }
    net.conv("conv_redir", "conv3a", "conv_redir", 1, 1, 0, 32);
    net.concat("blob20", {"conv_redir", "corr"}, "blob20");
    net.conv("conv3_1", "blob20", "conv3_1", 3, 1, 1, 256);
    net.conv("conv4", "conv3_1", "conv4", 3, 2, 1, 512);
    net.conv("conv4_1", "conv4", "conv4_1", 3, 1, 1, 512);
    net.conv("conv5", "conv4_1", "conv5", 3, 2, 1, 512);
    net.conv("conv5_1", "conv5", "conv5_1", 3, 1, 1, 512);
    net.conv("conv6", "conv5_1", "conv6", 3, 2, 1, 1024);
    net.conv("conv6_1", "conv6", "conv6_1", 3, 1, 1, 1024);
    // refinement: predict_flow (3x3 -> 2, no ReLU), deconv (4x4 / 2 + ReLU), upsample_flow (4x4 / 2, 2 -> 2, no ReLU), Concat
    struct Stage { const char* skip; int level; int deconv_out; };
    const Stage stages[] = {{"conv5_1", 5, 512}, {"conv4_1", 4, 256}, {"conv3_1", 3, 128}, {"conv2a", 2, 64}};
    std::string feat = "conv6_1";
    int lvl = 6;
    for (const Stage& st : stages) {
      const std::string l = std::to_string(lvl), m = std::to_string(st.level);
      net.conv("predict_flow" + l, feat, "predict_flow" + l, 3, 1, 1, 2, false, false);
      net.conv("deconv" + m, feat, "deconv" + m, 4, 2, 1, st.deconv_out, true, true);
      net.conv("upsample_flow" + l + "to" + m, "predict_flow" + l, "upsampled_flow" + l + "to" + m, 4, 2, 1, 2, true, false);
      net.concat("concat" + m, {st.skip, "deconv" + m, "upsampled_flow" + l + "to" + m}, "concat" + m);
      feat = "concat" + m;
      lvl = st.level;
    }
    net.conv("predict_flow2", feat, "predict_flow2", 3, 1, 1, 2, false, false);

    auto forward_all = [&](bool timed) {
      for (MiniNet::Step& s : net.steps) {
        if (!timed) { s.layer->Forward(s.bottom, s.top); continue; }
        CUDA_CHECK(hipDeviceSynchronize());          // caffe time: Timer (device events) around every layer's Forward
        const auto t0 = std::chrono::steady_clock::now();
        s.layer->Forward(s.bottom, s.top);
        CUDA_CHECK(hipDeviceSynchronize());
        s.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
    };
    auto touch_weights = [&] {                       // what a training net does to its weights between forwards: the cached operands must go
      for (MiniNet::Step& s : net.steps)
        for (auto& b : s.layer->blobs()) (void)b->mutable_gpu_data();
    };
    for (int i = 0; i < warmup; ++i) { if (!use_cache) touch_weights(); forward_all(false); }
    CUDA_CHECK(hipDeviceSynchronize());
    const long long p0 = fn2_caffe_adapter_weight_packs(), r0 = fn2_caffe_adapter_weight_pack_reuses();
    // total: all layers back to back, one synchronisation at the end of every pass (caffe.cpp:352-366 forward_timer)
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iterations; ++i) { if (!use_cache) touch_weights(); forward_all(false); }
    CUDA_CHECK(hipDeviceSynchronize());
    *total_ms_per_forward = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iterations;
    *packs = fn2_caffe_adapter_weight_packs() - p0;
    *pack_reuses = fn2_caffe_adapter_weight_pack_reuses() - r0;
    for (int i = 0; i < iterations; ++i) { if (!use_cache) touch_weights(); forward_all(true); }
    *num_layers = (int)net.steps.size();
    for (int i = 0; i < (int)net.steps.size() && i < max_layers; ++i) {
      layer_ms[i] = net.steps[i].ms / iterations;
      std::snprintf(layer_names + 48 * i, 48, "%s", net.steps[i].name.c_str());
    }
    const Blob<float>* out = net.blob("predict_flow2");
    const float* o = out->cpu_data();
    int ok = out->count() == N * 2 * (H / 4) * (W / 4);
    for (int i = 0; i < out->count(); ++i) ok &= std::isfinite(o[i]) ? 1 : 0;
    *output_finite = ok;
  });
}
#endif
