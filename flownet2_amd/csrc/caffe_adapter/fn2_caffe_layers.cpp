// Caffe adapter: the FlowNet2 custom layers as `caffe::Layer<float>` plug-ins whose Forward_gpu / Backward_gpu
// call libflownet2_hip.so (the C ABI in include/flownet2_hip.h) with the Blobs' device pointers.
//
// How a maintainer of the reference uses it: drop this file into src/caffe/layers/, REMOVE the
// reference's correlation_layer.{cpp,cu}, flow_warp_layer.{cpp,cu}, resample_layer.{cpp,cu}, l1loss_layer.{cpp,cu},
// channel_norm_layer.{cpp,cu}, downsample_layer.{cpp,cu} from the build (a type string can only be registered
// once, layer_factory.hpp:69-70), build Caffe for ROCm and link -lflownet2_hip.  Prototxts and .caffemodel files
// are untouched: same `type:` strings, same *_param fields, same blob shapes.
//
// In this repository the file is compiled against the stand-in headers in compat/ (no Caffe tree exists here),
// and the resulting plug-ins are driven through LayerRegistry exactly like the reference's own classes
// (tests/test_caffe_adapter.py compares the two side by side on the GPU).
//
// Class and member names follow the reference headers (include/caffe/layers/correlation_layer.hpp:26-77,
// flow_warp_layer.hpp, l1_loss_layer.hpp, channel_norm_layer.hpp, downsample_layer.hpp,
// src/caffe/layers/resample_layer.hpp) so that the class is recognisable; the bodies are new.
// Only Dtype = float is instantiated: the C ABI is fp32 (the reference's tools use float, tools/caffe.cpp:203).
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "caffe/blob.hpp"
#include "caffe/filler.hpp"
#include "caffe/layer.hpp"
#include "caffe/layer_factory.hpp"
#include "caffe/proto/caffe.pb.h"

#include "flownet2_hip.h"

namespace caffe {

#define FN2_CALL(expr)                                                                  \
  do {                                                                                  \
    const int fn2_rc_ = (expr);                                                         \
    if (fn2_rc_ != FN2_OK) LOG(FATAL) << #expr << " -> " << fn2_rc_ << ": " << fn2_last_error_string(); \
  } while (0)

// The C ABI is fp32.  In a real Caffe build INSTANTIATE_CLASS also instantiates Dtype = double
// (common.hpp:41-66); those instantiations must compile but abort if they are ever run.
template <typename Dtype> static inline const float* f32(const Dtype* p) {
  if constexpr (std::is_same<Dtype, float>::value) { return p; }
  else { LOG(FATAL) << "flownet2_hip layers are fp32 only (Dtype = double requested)"; return nullptr; }
}
template <typename Dtype> static inline float* f32(Dtype* p) {
  if constexpr (std::is_same<Dtype, float>::value) { return p; }
  else { LOG(FATAL) << "flownet2_hip layers are fp32 only (Dtype = double requested)"; return nullptr; }
}

// All reference launches go to the legacy default stream (no stream argument anywhere in the fork); so do we.
static void* const kStream = nullptr;

// ---------------------------------------------------------------------------------------------------------
template <typename Dtype>
class CorrelationLayer : public Layer<Dtype> {
 public:
  explicit CorrelationLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    const CorrelationParameter& cp = this->layer_param_.correlation_param();
    CHECK(cp.has_kernel_size()) << "Filter kernel_size is not set";
    CHECK(cp.has_max_displacement()) << "Max displacement is required.";
    if (cp.kernel_size() % 2 == 0) LOG(FATAL) << "Odd kernel size required";
    params_.pad = cp.pad();
    params_.kernel_size = cp.kernel_size();
    params_.max_displacement = cp.max_displacement();
    params_.stride1 = cp.stride_1();
    params_.stride2 = cp.stride_2();
    params_.corr_type = cp.correlation_type() == CorrelationParameter_CorrelationType_SUBTRACT ? FN2_CORR_SUBTRACT : FN2_CORR_MULTIPLY;
    params_.do_abs = cp.do_abs();
    params_.single_direction = 0;   // Correlation1D only
  }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom[0]->width(), bottom[1]->width()) << "Both bottom blobs must have same width";
    CHECK_EQ(bottom[0]->height(), bottom[1]->height()) << "Both bottom blobs must have same height";
    CHECK_EQ(bottom[0]->channels(), bottom[1]->channels()) << "Both bottom blobs must have same height";
    num_ = bottom[0]->num();
    FN2_CALL(fn2_correlation_out_shape(&params_, bottom[0]->channels(), bottom[0]->height(), bottom[0]->width(),
                                       &top_channels_, &top_height_, &top_width_));
    top[0]->Reshape(num_, top_channels_, top_height_, top_width_);
    // no rbot1_/rbot2_/rtopdiff_ scratch: the kernels read NCHW directly
  }
  virtual inline const char* type() const { return "Correlation"; }
  virtual inline int ExactNumBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom.size(), 2u);
    CHECK_EQ(top.size(), 1u);
    FN2_CALL(fn2_correlation_forward(&params_, f32(bottom[0]->gpu_data()), f32(bottom[1]->gpu_data()), f32(top[0]->mutable_gpu_data()),
                                     bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(), bottom[0]->width(),
                                     nullptr, 0, kStream));
  }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>&, const vector<Blob<Dtype>*>& bottom) {
    // like the reference: both diffs are always written, propagate_down is ignored
    FN2_CALL(fn2_correlation_backward(&params_, f32(bottom[0]->gpu_data()), f32(bottom[1]->gpu_data()), f32(top[0]->gpu_diff()),
                                      f32(bottom[0]->mutable_gpu_diff()), f32(bottom[1]->mutable_gpu_diff()),
                                      bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(), bottom[0]->width(),
                                      nullptr, 0, kStream));
  }
  fn2_corr_params params_;
  int num_, top_height_, top_width_, top_channels_;
};

// ---------------------------------------------------------------------------------------------------------
// Correlation1D (horizontal displacements; correlation_layer1d.cpp / .cu of the reference)
template <typename Dtype>
class Correlation1DLayer : public Layer<Dtype> {
 public:
  explicit Correlation1DLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    const CorrelationParameter& cp = this->layer_param_.correlation_param();
    CHECK(cp.has_kernel_size()) << "Filter kernel_size is not set";
    CHECK(cp.has_max_displacement()) << "Max displacement is required.";
    if (cp.kernel_size() % 2 == 0) LOG(FATAL) << "Odd kernel size required";
    if (cp.single_direction() < -1 || cp.single_direction() > 1) LOG(FATAL) << "single_direction must be -1 (left), 0 (off), or 1 (right)";
    params_.pad = cp.pad();
    params_.kernel_size = cp.kernel_size();
    params_.max_displacement = cp.max_displacement();
    params_.stride1 = cp.stride_1();
    params_.stride2 = cp.stride_2();
    params_.corr_type = cp.correlation_type() == CorrelationParameter_CorrelationType_SUBTRACT ? FN2_CORR_SUBTRACT : FN2_CORR_MULTIPLY;
    params_.do_abs = cp.do_abs();
    params_.single_direction = cp.single_direction();
  }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom[0]->width(), bottom[1]->width()) << "Both bottom blobs must have same width";
    CHECK_EQ(bottom[0]->height(), bottom[1]->height()) << "Both bottom blobs must have same height";
    CHECK_EQ(bottom[0]->channels(), bottom[1]->channels()) << "Both bottom blobs must have same number of channels";
    int tc, th, tw;
    FN2_CALL(fn2_correlation1d_out_shape(&params_, bottom[0]->channels(), bottom[0]->height(), bottom[0]->width(), &tc, &th, &tw));
    top[0]->Reshape(bottom[0]->num(), tc, th, tw);
  }
  virtual inline const char* type() const { return "Correlation1D"; }
  virtual inline int ExactNumBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom.size(), 2u);
    CHECK_EQ(top.size(), 1u);
    FN2_CALL(fn2_correlation1d_forward(&params_, f32(bottom[0]->gpu_data()), f32(bottom[1]->gpu_data()), f32(top[0]->mutable_gpu_data()),
                                       bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(), bottom[0]->width(), kStream));
  }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>&, const vector<Blob<Dtype>*>& bottom) {
    FN2_CALL(fn2_correlation1d_backward(&params_, f32(bottom[0]->gpu_data()), f32(bottom[1]->gpu_data()), f32(top[0]->gpu_diff()),
                                        f32(bottom[0]->mutable_gpu_diff()), f32(bottom[1]->mutable_gpu_diff()),
                                        bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(), bottom[0]->width(), kStream));
  }
  fn2_corr_params params_;
};

// ---------------------------------------------------------------------------------------------------------
// FlowAugmentation (flow_augmentation_layer.cpp / .cu of the reference): the coefficient blobs are read on the host, as there.
template <typename Dtype>
class FlowAugmentationLayer : public Layer<Dtype> {
 public:
  explicit FlowAugmentationLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_GT(this->layer_param_.augmentation_param().crop_width(), 0u) << "Please enter crop width if you want to perform augmentation";
    CHECK_GT(this->layer_param_.augmentation_param().crop_height(), 0u) << "Please enter crop height if you want to perform augmentation";
    this->layer_param_.set_reshape_every_iter(false);
  }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom.size(), 3u) << "Flow augmentation layer takes three input blobs: FlowField, Img1TransfParams, Img2TransfParams";
    CHECK_EQ(top.size(), 1u) << "Flow augmentation layer outputs one output blob: Augmented Flow";
    CHECK_EQ(bottom[0]->channels(), 2) << "Flow data must have two channels";
    cropped_width_ = this->layer_param_.augmentation_param().crop_width();
    cropped_height_ = this->layer_param_.augmentation_param().crop_height();
    top[0]->Reshape(bottom[0]->num(), 2, cropped_height_, cropped_width_);
  }
  virtual inline const char* type() const { return "FlowAugmentation"; }
  virtual inline bool AllowBackward() const { return false; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { LOG(FATAL) << "Forward CPU Augmentation not implemented."; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { LOG(FATAL) << "FlowAugmentationLayer cannot do backward."; }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { LOG(FATAL) << "FlowAugmentationLayer cannot do backward."; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom[1]->count() / bottom[0]->num(), FN2_AUG_NUM_PARAMS) << "coefficient blob does not hold one AugmentationCoeff per sample";
    CHECK_EQ(bottom[2]->count(), bottom[1]->count());
    FN2_CALL(fn2_flow_augmentation_forward(f32(bottom[0]->gpu_data()), f32(bottom[1]->cpu_data()), f32(bottom[2]->cpu_data()),
                                           f32(top[0]->mutable_gpu_data()), bottom[0]->num(), bottom[0]->height(), bottom[0]->width(),
                                           cropped_height_, cropped_width_, kStream));
  }
  int cropped_height_, cropped_width_;
};

// ---------------------------------------------------------------------------------------------------------
template <typename Dtype>
class FlowWarpLayer : public Layer<Dtype> {
 public:
  explicit FlowWarpLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom.size(), 2u) << "FlowWarpLayer takes two input blobs: image and flow.";
    CHECK_EQ(top.size(), 1u) << "FlowWarpLayer outputs one blob.";
    CHECK_EQ(bottom[0]->num(), bottom[1]->num()) << "Num of the inputs should be the same";
    CHECK_EQ(2, bottom[1]->channels()) << "Flow should have 2 channels: x-flow and y-flow";
    CHECK_EQ(bottom[0]->width(), bottom[1]->width()) << "Width of the inputs should be the same";
    CHECK_EQ(bottom[0]->height(), bottom[1]->height()) << "Height of the inputs should be the same";
    top[0]->Reshape(bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(), bottom[0]->width());
  }
  virtual inline const char* type() const { return "FlowWarp"; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    const int fill = this->layer_param_.flow_warp_param().fill_value() == FlowWarpParameter_FillParameter_ZERO ? FN2_FILL_ZERO : FN2_FILL_NAN;
    FN2_CALL(fn2_flow_warp_forward(f32(bottom[0]->gpu_data()), f32(bottom[1]->gpu_data()), f32(top[0]->mutable_gpu_data()), top[0]->num(),
                                   top[0]->channels(), top[0]->height(), top[0]->width(), fill, kStream));
  }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>& bottom) {
    // scratch for the inverted scatter (replaces the reference's transposed_image_ blob, flow_warp_layer.cpp:50)
    const size_t ws_bytes = fn2_flow_warp_backward_workspace_bytes(top[0]->num(), top[0]->channels(), top[0]->height(), top[0]->width());
    workspace_.Reshape(vector<int>{(int)((ws_bytes + sizeof(Dtype) - 1) / sizeof(Dtype))});
    FN2_CALL(fn2_flow_warp_backward(f32(bottom[0]->gpu_data()), f32(bottom[1]->gpu_data()), f32(top[0]->gpu_diff()), f32(bottom[0]->mutable_gpu_diff()),
                                    f32(bottom[1]->mutable_gpu_diff()), top[0]->num(), top[0]->channels(), top[0]->height(),
                                    top[0]->width(), propagate_down[0], propagate_down[1], workspace_.mutable_gpu_data(),
                                    workspace_.count() * sizeof(Dtype), kStream));
  }
  Blob<Dtype> workspace_;
};

// ---------------------------------------------------------------------------------------------------------
template <typename Dtype>
class ResampleLayer : public Layer<Dtype> {
 public:
  explicit ResampleLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) {
    const ResampleParameter_ResampleType t = this->layer_param().resample_param().type();
    if (t != ResampleParameter_ResampleType_CUBIC && t != ResampleParameter_ResampleType_LINEAR && t != ResampleParameter_ResampleType_NEAREST)
      LOG(FATAL) << "ResampleLayer: only CUBIC, LINEAR and NEAREST interpolation is supported for now";
  }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    this->layer_param_.set_reshape_every_iter(false);
    LOG(WARNING) << "ResampleLayer only runs Reshape on setup";
    CHECK_GE(bottom.size(), 1u);
    CHECK_LE(bottom.size(), 2u);
    CHECK_EQ(top.size(), 1u);
    int top_height, top_width;
    if (bottom.size() == 1) {
      top_height = this->layer_param_.resample_param().height();
      top_width = this->layer_param_.resample_param().width();
    } else {
      top_height = bottom[1]->height();
      top_width = bottom[1]->width();
    }
    CHECK_GE(top_height, 1) << "ResampleLayer must have top_height > 0";
    CHECK_GE(top_width, 1) << "ResampleLayer must have top_width > 0";
    top[0]->Reshape(bottom[0]->num(), bottom[0]->channels(), top_height, top_width);
  }
  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int MaxBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
  virtual inline bool AllowBackward() const { LOG(WARNING) << "ResampleLayer does not do backward."; return false; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { LOG(FATAL) << "ResampleLayer: CPU Forward not yet implemented."; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { LOG(FATAL) << "ResampleLayer cannot do backward."; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(top[0]->channels(), bottom[0]->channels()) << "ResampleLayer top channel count must match bottom channel count";
    const ResampleParameter& rp = this->layer_param().resample_param();
    FN2_CALL(fn2_resample_forward(f32(bottom[0]->gpu_data()), f32(top[0]->mutable_gpu_data()), bottom[0]->num(), bottom[0]->channels(),
                                  bottom[0]->height(), bottom[0]->width(), top[0]->height(), top[0]->width(), (int)rp.type(),
                                  rp.antialias(), kStream));
  }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>&) {
    for (size_t i = 0; i < propagate_down.size(); i++)
      if (propagate_down[i]) LOG(FATAL) << "ResampleLayer cannot do backward.";
  }
};

// ---------------------------------------------------------------------------------------------------------
template <typename Dtype>
class L1LossLayer : public Layer<Dtype> {
 public:
  explicit L1LossLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    if (this->layer_param_.loss_weight_size() == 0) this->layer_param_.add_loss_weight(Dtype(1));   // LossLayer::LayerSetUp, loss_layer.cpp:8-13
    if (bottom.size() != 1 && bottom.size() != 2) LOG(FATAL) << "L1LossLayer needs one or two input blobs.";
    const L1LossParameter& lp = this->layer_param_.l1_loss_param();
    params_.l2_per_location = lp.l2_per_location();
    params_.l2_prescale_by_channels = lp.l2_prescale_by_channels();
    params_.normalize_by_num_entries = lp.normalize_by_num_entries();
    params_.epsilon = lp.epsilon();
    params_.plateau = lp.plateau();
  }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    top[0]->Reshape(vector<int>());   // Loss layers output a scalar; 0 axes.
    const size_t need = fn2_l1loss_workspace_bytes(bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(), bottom[0]->width());
    workspace_.Reshape(vector<int>{(int)((need + sizeof(Dtype) - 1) / sizeof(Dtype))});
    loss_dev_.Reshape(vector<int>{1});
  }
  virtual inline const char* type() const { return "L1Loss"; }
  virtual inline int ExactNumBottomBlobs() const { return -1; }
  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int MaxBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
  virtual inline bool AllowForceBackward(const int) const { return true; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    FN2_CALL(fn2_l1loss_forward(&params_, f32(bottom[0]->gpu_data()), bottom.size() > 1 ? f32(bottom[1]->gpu_data()) : nullptr,
                                f32(loss_dev_.mutable_gpu_data()), bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(),
                                bottom[0]->width(), f32(workspace_.mutable_gpu_data()), workspace_.count() * sizeof(Dtype), kStream));
    top[0]->mutable_cpu_data()[0] = loss_dev_.cpu_data()[0];   // the reference also writes the scalar on the host (l1loss_layer.cu:142)
  }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>& bottom) {
    bool prop_down = propagate_down[0];
    if (bottom.size() > 1) prop_down |= propagate_down[1];
    if (!prop_down) return;
    FN2_CALL(fn2_l1loss_backward(&params_, f32(bottom[0]->gpu_data()), bottom.size() > 1 ? f32(bottom[1]->gpu_data()) : nullptr,
                                 top[0]->cpu_diff()[0], f32(bottom[0]->mutable_gpu_diff()),
                                 bottom.size() > 1 ? f32(bottom[1]->mutable_gpu_diff()) : nullptr, bottom[0]->num(), bottom[0]->channels(),
                                 bottom[0]->height(), bottom[0]->width(), f32(workspace_.mutable_gpu_data()),
                                 workspace_.count() * sizeof(Dtype), kStream));
  }
  fn2_l1loss_params params_;
  Blob<Dtype> workspace_, loss_dev_;
};

// ---------------------------------------------------------------------------------------------------------
template <typename Dtype>
class ChannelNormLayer : public Layer<Dtype> {
 public:
  explicit ChannelNormLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom.size(), 1u) << "ChannelNormLayer takes two input blobs: image and flow.";   // (sic) channel_norm_layer.cpp:30
    CHECK_EQ(top.size(), 1u) << "ChannelNormLayer outputs one blob.";
    top[0]->Reshape(bottom[0]->num(), 1, bottom[0]->height(), bottom[0]->width());
  }
  virtual inline const char* type() const { return "NormLayer"; }   // (sic) channel_norm_layer.hpp:22
  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    FN2_CALL(fn2_channel_norm_forward(f32(bottom[0]->gpu_data()), f32(top[0]->mutable_gpu_data()), bottom[0]->num(), bottom[0]->channels(),
                                      bottom[0]->height(), bottom[0]->width(), kStream));
  }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>&, const vector<Blob<Dtype>*>& bottom) {
    FN2_CALL(fn2_channel_norm_backward(f32(bottom[0]->gpu_data()), f32(top[0]->gpu_data()), f32(top[0]->gpu_diff()), f32(bottom[0]->mutable_gpu_diff()),
                                       bottom[0]->num(), bottom[0]->channels(), bottom[0]->height(), bottom[0]->width(), kStream));
  }
};

// ---------------------------------------------------------------------------------------------------------
template <typename Dtype>
class DownsampleLayer : public Layer<Dtype> {
 public:
  explicit DownsampleLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    this->layer_param_.set_reshape_every_iter(false);
    LOG(WARNING) << "DownsampleLayer only runs Reshape on setup";
    CHECK_GE(bottom.size(), 1u);
    CHECK_LE(bottom.size(), 2u);
    CHECK_EQ(top.size(), 1u);
    if (bottom.size() == 1) {
      top_height_ = this->layer_param_.downsample_param().top_height();
      top_width_ = this->layer_param_.downsample_param().top_width();
    } else {
      top_height_ = bottom[1]->height();
      top_width_ = bottom[1]->width();
    }
    CHECK_GE(top_height_, 1) << "DownsampleLayer must have top_height > 0";
    CHECK_GE(top_width_, 1) << "DownsampleLayer must have top_width > 0";
    top[0]->Reshape(bottom[0]->num(), bottom[0]->channels(), top_height_, top_width_);
    // the reference shares data/diff with the bottom when the sizes agree (downsample_layer.cpp:53-56); we copy in Forward
  }
  virtual inline const char* type() const { return "Downsample"; }
  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int MaxBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
  virtual inline bool AllowBackward() const { return false; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { LOG(FATAL) << "DownsampleLayer: CPU Forward not yet implemented."; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { LOG(FATAL) << "DownsampleLayer cannot do backward."; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    FN2_CALL(fn2_downsample_forward(f32(bottom[0]->gpu_data()), f32(top[0]->mutable_gpu_data()), bottom[0]->num(), bottom[0]->channels(),
                                    bottom[0]->height(), bottom[0]->width(), top_height_, top_width_, kStream));
  }
  virtual void Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>&) {
    for (size_t i = 0; i < propagate_down.size(); i++)
      if (propagate_down[i]) LOG(FATAL) << "DownsamplingLayer cannot do backward.";
  }
  int top_width_, top_height_;
};

// ---------------------------------------------------------------------------------------------------------
// Convolution / Deconvolution as plug-ins: one class for both (transposed_ tells them apart), the kernel family picked by the LIBRARY from
// the layer's geometry (fn2_conv_route / fn2_deconv_route, csrc/conv_route.cpp -- the routing the benchmarks ran with).
//   <- ConvolutionLayer / DeconvolutionLayer (conv_layer.cpp:8-40, deconv_layer.cpp:8-45) over BaseConvolutionLayer::LayerSetUp / Reshape
//      (base_conv_layer.cpp:14-253): same convolution_param fields (num_output, kernel_size, stride, pad, bias_term, weight / bias fillers),
//      same blob shapes -- weight [num_output, C, k, k] (Deconvolution: [C, num_output, k, k]), bias [num_output] --, same top shape.
// Scope: square kernels, group 1, dilation 1, one bottom / top pair, the geometries the library has a kernel for (every Convolution and
// Deconvolution of the FlowNet graphs); anything else aborts in Reshape with a message naming the layer -- keep the stock class for those
// (INTEGRATION.md shows the two-line factory that falls back to it).
// The packed weight operand (round 6): packed ONCE and kept while the weights provably have not changed -- Caffe has no "weights changed"
// signal below Solver::ApplyUpdate, so the layer reads what SyncedMemory already records: the operand is reused iff the layer runs in the
// TEST phase, the weight blob's memory is SYNCED on entry (no mutable_cpu_data / mutable_gpu_data since the last const access: a
// CopyTrainedLayersFrom, a pycaffe `net.params[...]` write or a device-side update all leave HEAD_AT_CPU / HEAD_AT_GPU), its SyncedMemory is
// not shared with another blob (a solver's test net shares the train net's weights, Net::ShareTrainedLayersWith: those are updated behind
// this layer's back and re-synced by snapshots), and the device pointer is the one that was packed.  Everything else -- every TRAIN-phase
// forward -- repacks.  fn2_caffe_adapter_stats() counts packs and reuses (tests/test_caffe_adapter.py).
// Backward_gpu (conv_layer.cu:26-60 / deconv_layer.cu:27-58) runs the library's own gradient routes; parameter diffs are accumulated.
struct Fn2AdapterStats { long long weight_packs, weight_pack_reuses; };
inline Fn2AdapterStats& fn2_adapter_stats() { static Fn2AdapterStats s{0, 0}; return s; }

template <typename Dtype>
class Fn2ConvolutionLayer : public Layer<Dtype> {
 public:
  explicit Fn2ConvolutionLayer(const LayerParameter& param, bool transposed) : Layer<Dtype>(param), transposed_(transposed) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    const ConvolutionParameter& cp = this->layer_param_.convolution_param();
    CHECK(cp.kernel_size_size() == 1 && !cp.has_kernel_h() && !cp.has_kernel_w()) << "square kernel_size required";
    CHECK_LE(cp.stride_size(), 1); CHECK_LE(cp.pad_size(), 1); CHECK_LE(cp.dilation_size(), 1);
    CHECK(cp.dilation_size() == 0 || cp.dilation(0) == 1) << "dilation 1 only";
    CHECK_EQ(cp.group(), 1u) << "group 1 only";
    kernel_ = (int)cp.kernel_size(0);
    stride_ = cp.stride_size() ? (int)cp.stride(0) : 1;
    pad_ = cp.pad_size() ? (int)cp.pad(0) : 0;
    num_output_ = (int)cp.num_output();
    CHECK_GT(num_output_, 0);
    bias_term_ = cp.bias_term();
    channels_ = bottom[0]->channels();
    if (this->blobs_.size() == 0) {
      this->blobs_.resize(bias_term_ ? 2 : 1);
      // base_conv_layer.cpp:125-139: [conv_out_channels, conv_in_channels / group, k, k]; for a Deconvolution the roles are swapped
      const int d0 = transposed_ ? channels_ : num_output_, d1 = transposed_ ? num_output_ : channels_;
      this->blobs_[0].reset(new Blob<Dtype>(d0, d1, kernel_, kernel_));
      shared_ptr<Filler<Dtype> > wf(GetFiller<Dtype>(cp.weight_filler()));
      wf->Fill(this->blobs_[0].get());
      if (bias_term_) {
        this->blobs_[1].reset(new Blob<Dtype>(vector<int>{num_output_}));
        shared_ptr<Filler<Dtype> > bf(GetFiller<Dtype>(cp.bias_filler()));
        bf->Fill(this->blobs_[1].get());
      }
    }
    this->param_propagate_down_.resize(this->blobs_.size(), true);
  }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CHECK_EQ(bottom[0]->channels(), channels_) << "Input size incompatible with convolution kernel.";
    desc_.N = bottom[0]->num(); desc_.Cin = channels_; desc_.Hin = bottom[0]->height(); desc_.Win = bottom[0]->width();
    desc_.Cout = num_output_; desc_.kernel = kernel_; desc_.stride = stride_; desc_.pad = pad_;
    int oh, ow;
    if (transposed_) {      // deconv_layer.cpp:8-24
      oh = stride_ * (desc_.Hin - 1) + kernel_ - 2 * pad_; ow = stride_ * (desc_.Win - 1) + kernel_ - 2 * pad_;
      route_ = fn2_deconv_route(&desc_, 0);
    } else {                // conv_layer.cpp:8-23
      oh = (desc_.Hin + 2 * pad_ - kernel_) / stride_ + 1; ow = (desc_.Win + 2 * pad_ - kernel_) / stride_ + 1;
      route_ = fn2_conv_route(&desc_, 0);
    }
    if (route_ == 0)
      LOG(FATAL) << this->layer_param_.name() << ": libflownet2_hip has no kernel for " << type() << "{kernel " << kernel_ << ", stride " << stride_
                 << ", pad " << pad_ << "} " << channels_ << " -> " << num_output_ << " on " << desc_.Hin << " x " << desc_.Win << "; keep the stock layer for it";
    top[0]->Reshape(desc_.N, num_output_, oh, ow);
    const size_t pf = transposed_ ? fn2_deconv_packed_weight_floats(&desc_, route_) : fn2_conv_packed_weight_floats(&desc_, route_);
    size_t wb = transposed_ ? fn2_deconv_workspace_bytes(&desc_, route_) : fn2_conv_workspace_bytes(&desc_, route_);
    packed_.Reshape(vector<int>{(int)pf});
    // backward: the library's own routes (csrc/conv_route.cpp); one scratch blob serves every call of the layer (they run in stream order)
    bwd_route_ = fn2_conv_backward_data_route(&desc_, transposed_);
    bwd_weights_ = fn2_conv_backward_weights_supported(&desc_, transposed_) != 0;
    const int tr = transposed_ ? 1 : 0;
    wb = std::max(wb, fn2_bias_leaky_relu_backward_workspace_bytes(desc_.N, num_output_, oh, ow));
    head_ = transposed_ ? route_ == FN2_DECONV_ROUTE_HEAD : route_ == FN2_CONV_ROUTE_HEAD;      // the 2-channel flow heads: kernels of their own, forward and backward
    if (head_) wb = std::max(wb, transposed_ ? fn2_upsample_flow_deconv_backward_workspace_bytes(desc_.N, desc_.Hin, desc_.Win)
                                             : fn2_predict_flow_conv_backward_workspace_bytes(desc_.N, channels_, desc_.Hin, desc_.Win));
    wb = std::max(wb, fn2_conv_backward_weights_workspace_bytes(&desc_, tr));
    if (bwd_route_ != FN2_BWD_ROUTE_NONE) {
      wb = std::max(wb, fn2_conv_backward_data_workspace_bytes(&desc_, tr, bwd_route_));
      wb = std::max(wb, fn2_conv_backward_data_pack_workspace_bytes(&desc_, tr, bwd_route_));
      packed_bwd_.Reshape(vector<int>{(int)fn2_conv_backward_data_packed_weight_floats(&desc_, tr, bwd_route_)});
    }
    workspace_.Reshape(vector<int>{(int)((wb + 3) / 4) + 4});
  }
  virtual inline const char* type() const { return transposed_ ? "Deconvolution" : "Convolution"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, const vector<Blob<Dtype>*>&) { NOT_IMPLEMENTED; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    // decided BEFORE gpu_data() touches the head state (see the class comment)
    const bool untouched = this->blobs_[0]->data()->head() == SyncedMemory::SYNCED && this->blobs_[0]->data().use_count() == 1;
    const float* w = f32(this->blobs_[0]->gpu_data());
    const float* b = bias_term_ ? f32(this->blobs_[1]->gpu_data()) : nullptr;
    const bool reuse = this->phase_ == TEST && untouched && packed_for_ == w && packed_route_ == route_ && packed_count_ == packed_.count();
    float* pk = f32(packed_.mutable_gpu_data());
    void* ws = workspace_.mutable_gpu_data();
    const size_t wsb = sizeof(Dtype) * (size_t)workspace_.count();
    if (reuse) {
      fn2_adapter_stats().weight_pack_reuses++;
    } else {
      if (transposed_) FN2_CALL(fn2_deconv_pack_weights(&desc_, route_, w, pk, kStream));
      else FN2_CALL(fn2_conv_pack_weights(&desc_, route_, w, pk, kStream));
      packed_for_ = w; packed_route_ = route_; packed_count_ = packed_.count();
      fn2_adapter_stats().weight_packs++;
    }
    if (transposed_) {
      FN2_CALL(fn2_deconv_forward(&desc_, route_, f32(bottom[0]->gpu_data()), channels_, 0, pk, b, f32(top[0]->mutable_gpu_data()), num_output_, 0,
                                  0, 0.f, ws, wsb, kStream));
    } else {
      FN2_CALL(fn2_conv_forward(&desc_, route_, f32(bottom[0]->gpu_data()), channels_, 0, pk, b, f32(top[0]->mutable_gpu_data()), num_output_, 0,
                                0, 0.f, ws, wsb, kStream));
    }
  }
  // conv_layer.cu:26-60 / deconv_layer.cu:27-58: bias_diff += sum of top_diff (backward_gpu_bias), weight_diff += ... (weight_gpu_gemm, beta = 1:
  // the solver clears the parameter diffs once per iteration), bottom_diff = ... (backward_gpu_gemm / forward_gpu_gemm: overwritten)
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>& bottom) {
    const int tr = transposed_ ? 1 : 0;
    const float* td = f32(top[0]->gpu_diff());
    void* ws = workspace_.mutable_gpu_data();
    const size_t wsb = sizeof(Dtype) * (size_t)workspace_.count();
    if (head_) {
      // predict_flow* / upsample_flow*: one call computes the three gradients (csrc/flow_head_bwd.hip); parameter diffs accumulate, NULL = not wanted
      float* bd = propagate_down[0] ? f32(bottom[0]->mutable_gpu_diff()) : nullptr;
      float* wd = this->param_propagate_down_[0] ? f32(this->blobs_[0]->mutable_gpu_diff()) : nullptr;
      float* db = (bias_term_ && this->param_propagate_down_[1]) ? f32(this->blobs_[1]->mutable_gpu_diff()) : nullptr;
      const float* w = f32(this->blobs_[0]->gpu_data());
      if (transposed_) {
        FN2_CALL(fn2_upsample_flow_deconv_backward(f32(bottom[0]->gpu_data()), w, td, bd, wd, db, desc_.N, desc_.Hin, desc_.Win, 1, ws, wsb, kStream));
      } else {
        if (!fn2_predict_flow_conv_backward_supported(desc_.N, channels_, desc_.Hin, desc_.Win))
          LOG(FATAL) << this->layer_param_.name() << ": libflownet2_hip has no backward kernel for this flow head; keep the stock layer for it";
        FN2_CALL(fn2_predict_flow_conv_backward(f32(bottom[0]->gpu_data()), channels_, 0, w, td, bd, wd, db, desc_.N, channels_, desc_.Hin, desc_.Win, 1, ws, wsb, kStream));
      }
      return;
    }
    if (bias_term_ && this->param_propagate_down_[1])
      FN2_CALL(fn2_conv_backward_bias(td, num_output_, 0, f32(this->blobs_[1]->mutable_gpu_diff()), desc_.N, num_output_, top[0]->height(), top[0]->width(),
                                      1, ws, wsb, kStream));
    if (this->param_propagate_down_[0]) {
      if (!bwd_weights_)
        LOG(FATAL) << this->layer_param_.name() << ": libflownet2_hip has no weight-gradient kernel for this " << type() << "; keep the stock layer for it";
      FN2_CALL(fn2_conv_backward_weights(&desc_, tr, f32(bottom[0]->gpu_data()), channels_, 0, td, num_output_, 0, f32(this->blobs_[0]->mutable_gpu_diff()),
                                         1, ws, wsb, kStream));
    }
    if (propagate_down[0]) {
      if (bwd_route_ == FN2_BWD_ROUTE_NONE)
        LOG(FATAL) << this->layer_param_.name() << ": libflownet2_hip has no data-gradient kernel for this " << type() << "; keep the stock layer for it";
      float* pk = f32(packed_bwd_.mutable_gpu_data());
      FN2_CALL(fn2_conv_backward_data_pack_weights(&desc_, tr, bwd_route_, f32(this->blobs_[0]->gpu_data()), pk, ws, wsb, kStream));
      // bottom_room = channels_: a Caffe blob has no room for the channel groups the kernels round up to (the result goes through the scratch)
      FN2_CALL(fn2_conv_backward_data(&desc_, tr, bwd_route_, td, num_output_, 0, pk, f32(bottom[0]->mutable_gpu_diff()), channels_, 0, channels_, ws, wsb, kStream));
    }
  }
  bool transposed_, bias_term_ = true, bwd_weights_ = false, head_ = false;
  int kernel_ = 0, stride_ = 1, pad_ = 0, num_output_ = 0, channels_ = 0, route_ = 0, bwd_route_ = 0;
  fn2_conv_desc desc_;
  Blob<Dtype> packed_, packed_bwd_, workspace_;
  const float* packed_for_ = nullptr;      // device pointer of the weights packed_ was built from
  int packed_route_ = -1, packed_count_ = -1;
};
template <typename Dtype> shared_ptr<Layer<Dtype> > Creator_Fn2Convolution(const LayerParameter& p) { return shared_ptr<Layer<Dtype> >(new Fn2ConvolutionLayer<Dtype>(p, false)); }
template <typename Dtype> shared_ptr<Layer<Dtype> > Creator_Fn2Deconvolution(const LayerParameter& p) { return shared_ptr<Layer<Dtype> >(new Fn2ConvolutionLayer<Dtype>(p, true)); }

INSTANTIATE_CLASS(CorrelationLayer);
REGISTER_LAYER_CLASS(Correlation);
INSTANTIATE_CLASS(Correlation1DLayer);
REGISTER_LAYER_CLASS(Correlation1D);
INSTANTIATE_CLASS(FlowAugmentationLayer);
REGISTER_LAYER_CLASS(FlowAugmentation);
INSTANTIATE_CLASS(FlowWarpLayer);
REGISTER_LAYER_CLASS(FlowWarp);
INSTANTIATE_CLASS(ResampleLayer);
REGISTER_LAYER_CLASS(Resample);
INSTANTIATE_CLASS(L1LossLayer);
REGISTER_LAYER_CLASS(L1Loss);
INSTANTIATE_CLASS(ChannelNormLayer);
REGISTER_LAYER_CLASS(ChannelNorm);
INSTANTIATE_CLASS(DownsampleLayer);
REGISTER_LAYER_CLASS(Downsample);
// one type string can be registered once (layer_factory.hpp:69-70): in a Caffe tree these two REPLACE GetConvolutionLayer /
// REGISTER_LAYER_CLASS(Deconvolution) of layer_factory.cpp:38-70 / deconv_layer.cpp:79-80 (or wrap them: INTEGRATION.md)
INSTANTIATE_CLASS(Fn2ConvolutionLayer);
REGISTER_LAYER_CREATOR(Convolution, Creator_Fn2Convolution);
REGISTER_LAYER_CREATOR(Deconvolution, Creator_Fn2Deconvolution);

}  // namespace caffe

// weight operands packed / reused by the Convolution and Deconvolution plug-ins of this process (the cache of Fn2ConvolutionLayer)
extern "C" __attribute__((visibility("default"))) long long fn2_caffe_adapter_weight_packs() { return caffe::fn2_adapter_stats().weight_packs; }
extern "C" __attribute__((visibility("default"))) long long fn2_caffe_adapter_weight_pack_reuses() { return caffe::fn2_adapter_stats().weight_pack_reuses; }
