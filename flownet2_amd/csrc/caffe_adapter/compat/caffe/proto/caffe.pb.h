// Stand-in for the protoc-generated caffe.pb.h: the messages the FlowNet2 hot-path layers read
// (src/caffe/proto/caffe.proto:312-425 LayerParameter; :553-560 FlowWarpParameter; :619-625 L1LossParameter;
// :628-644 CorrelationParameter; :646-649 DownsampleParameter; :665-677 ResampleParameter), with the accessor
// names protoc would generate.  Plain structs -- there is no protobuf runtime in this image.
#pragma once
#include <string>
#include <vector>

namespace caffe {

enum CorrelationParameter_CorrelationType { CorrelationParameter_CorrelationType_MULTIPLY = 0, CorrelationParameter_CorrelationType_SUBTRACT = 1 };
class CorrelationParameter {
 public:
  typedef CorrelationParameter_CorrelationType CorrelationType;
  static const CorrelationType MULTIPLY = CorrelationParameter_CorrelationType_MULTIPLY;
  static const CorrelationType SUBTRACT = CorrelationParameter_CorrelationType_SUBTRACT;
  unsigned pad() const { return pad_; }
  bool has_kernel_size() const { return has_kernel_size_; }
  unsigned kernel_size() const { return kernel_size_; }
  bool has_max_displacement() const { return has_max_displacement_; }
  unsigned max_displacement() const { return max_displacement_; }
  unsigned stride_1() const { return stride_1_; }
  unsigned stride_2() const { return stride_2_; }
  int single_direction() const { return single_direction_; }
  bool do_abs() const { return do_abs_; }
  CorrelationType correlation_type() const { return correlation_type_; }
  void set_pad(unsigned v) { pad_ = v; }
  void set_kernel_size(unsigned v) { kernel_size_ = v; has_kernel_size_ = true; }
  void set_max_displacement(unsigned v) { max_displacement_ = v; has_max_displacement_ = true; }
  void set_stride_1(unsigned v) { stride_1_ = v; }
  void set_stride_2(unsigned v) { stride_2_ = v; }
  void set_do_abs(bool v) { do_abs_ = v; }
  void set_correlation_type(CorrelationType v) { correlation_type_ = v; }
 private:
  unsigned pad_ = 0, kernel_size_ = 0, max_displacement_ = 0, stride_1_ = 1, stride_2_ = 1;
  int single_direction_ = 0;
  bool do_abs_ = false, has_kernel_size_ = false, has_max_displacement_ = false;
  CorrelationType correlation_type_ = CorrelationParameter_CorrelationType_MULTIPLY;
};

enum FlowWarpParameter_FillParameter { FlowWarpParameter_FillParameter_ZERO = 1, FlowWarpParameter_FillParameter_NOT_A_NUMBER = 2 };
class FlowWarpParameter {
 public:
  typedef FlowWarpParameter_FillParameter FillParameter;
  FillParameter fill_value() const { return fill_value_; }
  void set_fill_value(FillParameter v) { fill_value_ = v; }
 private:
  FillParameter fill_value_ = FlowWarpParameter_FillParameter_ZERO;
};

enum ResampleParameter_ResampleType { ResampleParameter_ResampleType_NEAREST = 1, ResampleParameter_ResampleType_LINEAR = 2,
                                      ResampleParameter_ResampleType_CUBIC = 3, ResampleParameter_ResampleType_AREA = 4 };
class ResampleParameter {
 public:
  typedef ResampleParameter_ResampleType ResampleType;
  bool antialias() const { return antialias_; }
  unsigned width() const { return width_; }
  unsigned height() const { return height_; }
  ResampleType type() const { return type_; }
  float factor() const { return factor_; }
  void set_antialias(bool v) { antialias_ = v; }
  void set_width(unsigned v) { width_ = v; }
  void set_height(unsigned v) { height_ = v; }
  void set_type(ResampleType v) { type_ = v; }
 private:
  bool antialias_ = true;
  unsigned width_ = 0, height_ = 0;
  ResampleType type_ = ResampleParameter_ResampleType_LINEAR;
  float factor_ = 1.0f;
};

class L1LossParameter {
 public:
  bool l2_per_location() const { return l2_per_location_; }
  bool l2_prescale_by_channels() const { return l2_prescale_by_channels_; }
  bool normalize_by_num_entries() const { return normalize_by_num_entries_; }
  float epsilon() const { return epsilon_; }
  float plateau() const { return plateau_; }
  void set_l2_per_location(bool v) { l2_per_location_ = v; }
  void set_l2_prescale_by_channels(bool v) { l2_prescale_by_channels_ = v; }
  void set_normalize_by_num_entries(bool v) { normalize_by_num_entries_ = v; }
  void set_epsilon(float v) { epsilon_ = v; }
  void set_plateau(float v) { plateau_ = v; }
 private:
  bool l2_per_location_ = false, l2_prescale_by_channels_ = false, normalize_by_num_entries_ = false;
  float epsilon_ = 1e-2f, plateau_ = 0.f;
};

class DownsampleParameter {
 public:
  unsigned top_height() const { return top_height_; }
  unsigned top_width() const { return top_width_; }
  void set_top_height(unsigned v) { top_height_ = v; }
  void set_top_width(unsigned v) { top_width_ = v; }
 private:
  unsigned top_height_ = 0, top_width_ = 0;
};

enum Phase { TRAIN = 0, TEST = 1 };

class LayerParameter {
 public:
  const std::string& name() const { return name_; }
  const std::string& type() const { return type_; }
  void set_name(const std::string& v) { name_ = v; }
  void set_type(const std::string& v) { type_ = v; }
  int loss_weight_size() const { return (int)loss_weight_.size(); }
  float loss_weight(int i) const { return loss_weight_[i]; }
  void add_loss_weight(float v) { loss_weight_.push_back(v); }
  bool reshape_every_iter() const { return reshape_every_iter_; }          // caffe.proto:424
  void set_reshape_every_iter(bool v) { reshape_every_iter_ = v; }
  Phase phase() const { return phase_; }
  void set_phase(Phase p) { phase_ = p; }
  const CorrelationParameter& correlation_param() const { return correlation_param_; }       // = 150
  CorrelationParameter* mutable_correlation_param() { return &correlation_param_; }
  const L1LossParameter& l1_loss_param() const { return l1_loss_param_; }                   // = 151
  L1LossParameter* mutable_l1_loss_param() { return &l1_loss_param_; }
  const ResampleParameter& resample_param() const { return resample_param_; }               // = 155
  ResampleParameter* mutable_resample_param() { return &resample_param_; }
  const DownsampleParameter& downsample_param() const { return downsample_param_; }         // = 156
  DownsampleParameter* mutable_downsample_param() { return &downsample_param_; }
  const FlowWarpParameter& flow_warp_param() const { return flow_warp_param_; }             // = 159
  FlowWarpParameter* mutable_flow_warp_param() { return &flow_warp_param_; }
 private:
  std::string name_, type_;
  std::vector<float> loss_weight_;
  bool reshape_every_iter_ = true;
  Phase phase_ = TEST;
  CorrelationParameter correlation_param_;
  L1LossParameter l1_loss_param_;
  ResampleParameter resample_param_;
  DownsampleParameter downsample_param_;
  FlowWarpParameter flow_warp_param_;
};

}  // namespace caffe
