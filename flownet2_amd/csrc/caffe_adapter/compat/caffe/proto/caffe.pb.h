// Stand-in for the protoc-generated caffe.pb.h: the messages the FlowNet2 hot-path layers read
// (src/caffe/proto/caffe.proto:312-425 LayerParameter; :553-560 FlowWarpParameter; :619-625 L1LossParameter;
// :628-644 CorrelationParameter; :646-649 DownsampleParameter; :665-677 ResampleParameter), with the accessor
// names protoc would generate.  Plain structs -- there is no protobuf runtime in this image.
#pragma once
#include <string>
#include <utility>
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
  void set_single_direction(int v) { single_direction_ = v; }
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

// Messages of the stock layers the reference's L1LossLayer is composed from (l1loss_layer.cpp:19-62): only the oracle/_ref
// build of that layer reads them (caffe.proto: FillerParameter, EltwiseParameter, PowerParameter, ConvolutionParameter).
enum FillerParameter_VarianceNorm { FillerParameter_VarianceNorm_FAN_IN = 0, FillerParameter_VarianceNorm_FAN_OUT = 1, FillerParameter_VarianceNorm_AVERAGE = 2 };
class FillerParameter {            // caffe.proto:43-64
 public:
  typedef FillerParameter_VarianceNorm VarianceNorm;
  const std::string& type() const { return type_; }
  float value() const { return value_; }
  float min() const { return min_; }
  float max() const { return max_; }
  float mean() const { return mean_; }
  float std() const { return std_; }
  int sparse() const { return sparse_; }
  VarianceNorm variance_norm() const { return variance_norm_; }
  int diag_val_size() const { return (int)diag_val_.size(); }
  float diag_val(int i) const { return diag_val_[i]; }
  void add_diag_val(float v) { diag_val_.push_back(v); }
  void set_type(const std::string& v) { type_ = v; }
  void set_value(float v) { value_ = v; }
 private:
  std::string type_ = "constant";
  float value_ = 0.f, min_ = 0.f, max_ = 1.f, mean_ = 0.f, std_ = 1.f;
  int sparse_ = -1;
  VarianceNorm variance_norm_ = FillerParameter_VarianceNorm_FAN_IN;
  std::vector<float> diag_val_;
};

enum EltwiseParameter_EltwiseOp { EltwiseParameter_EltwiseOp_PROD = 0, EltwiseParameter_EltwiseOp_SUM = 1, EltwiseParameter_EltwiseOp_MAX = 2 };
class EltwiseParameter {
 public:
  typedef EltwiseParameter_EltwiseOp EltwiseOp;
  EltwiseOp operation() const { return operation_; }
  void set_operation(EltwiseOp v) { operation_ = v; }
  int coeff_size() const { return (int)coeff_.size(); }
  float coeff(int i) const { return coeff_[i]; }
  void add_coeff(float v) { coeff_.push_back(v); }
  bool stable_prod_grad() const { return stable_prod_grad_; }
 private:
  EltwiseOp operation_ = EltwiseParameter_EltwiseOp_SUM;
  std::vector<float> coeff_;
  bool stable_prod_grad_ = true;
};

class PowerParameter {
 public:
  float power() const { return power_; }
  float scale() const { return scale_; }
  float shift() const { return shift_; }
  void set_power(float v) { power_ = v; }
  void set_scale(float v) { scale_ = v; }
  void set_shift(float v) { shift_ = v; }
 private:
  float power_ = 1.f, scale_ = 1.f, shift_ = 0.f;
};

enum ConvolutionParameter_Engine { ConvolutionParameter_Engine_DEFAULT = 0, ConvolutionParameter_Engine_CAFFE = 1, ConvolutionParameter_Engine_CUDNN = 2 };
class ConvolutionParameter {
 public:
  typedef ConvolutionParameter_Engine Engine;
  unsigned num_output() const { return num_output_; }
  void set_num_output(unsigned v) { num_output_ = v; }
  bool bias_term() const { return bias_term_; }
  void set_bias_term(bool v) { bias_term_ = v; }
#define FN2_REPEATED(name)                                         \
  int name##_size() const { return (int)name##_.size(); }          \
  unsigned name(int i) const { return name##_[i]; }                \
  void add_##name(unsigned v) { name##_.push_back(v); }            \
  const std::vector<unsigned>& name() const { return name##_; }
  FN2_REPEATED(pad) FN2_REPEATED(kernel_size) FN2_REPEATED(stride) FN2_REPEATED(dilation)
#undef FN2_REPEATED
#define FN2_OPTIONAL(name)                                         \
  bool has_##name() const { return has_##name##_; }                \
  unsigned name() const { return name##_; }                        \
  void set_##name(unsigned v) { name##_ = v; has_##name##_ = true; }
  FN2_OPTIONAL(pad_h) FN2_OPTIONAL(pad_w) FN2_OPTIONAL(kernel_h) FN2_OPTIONAL(kernel_w) FN2_OPTIONAL(stride_h) FN2_OPTIONAL(stride_w)
#undef FN2_OPTIONAL
  unsigned group() const { return group_; }
  const FillerParameter& weight_filler() const { return weight_filler_; }
  FillerParameter* mutable_weight_filler() { return &weight_filler_; }
  const FillerParameter& bias_filler() const { return bias_filler_; }
  FillerParameter* mutable_bias_filler() { return &bias_filler_; }
  Engine engine() const { return ConvolutionParameter_Engine_CAFFE; }
  int axis() const { return 1; }
  bool force_nd_im2col() const { return false; }
 private:
  unsigned num_output_ = 0, group_ = 1;
  bool bias_term_ = true;
  std::vector<unsigned> pad_, kernel_size_, stride_, dilation_;
  unsigned pad_h_ = 0, pad_w_ = 0, kernel_h_ = 0, kernel_w_ = 0, stride_h_ = 1, stride_w_ = 1;
  bool has_pad_h_ = false, has_pad_w_ = false, has_kernel_h_ = false, has_kernel_w_ = false, has_stride_h_ = false, has_stride_w_ = false;
  FillerParameter weight_filler_, bias_filler_;
};

class ReLUParameter {
 public:
  float negative_slope() const { return negative_slope_; }
  void set_negative_slope(float v) { negative_slope_ = v; }
 private:
  float negative_slope_ = 0.f;
};

// ConcatParameter, caffe.proto:780-789 (the adapter's FlowNetC timing driver chains the reference's own ConcatLayer between the plug-ins)
class ConcatParameter {
 public:
  int axis() const { return axis_; }
  void set_axis(int v) { axis_ = v; has_axis_ = true; }
  bool has_axis() const { return has_axis_; }
  bool has_concat_dim() const { return false; }
  unsigned concat_dim() const { return 1; }
 private:
  int axis_ = 1;
  bool has_axis_ = false;
};

// ---- data layers (oracle/_ref only: the reference's CustomData layer is compiled in place to pin the sample format) ----------
// Datum, caffe.proto:30-41.  ParseFromArray is a plain proto2 wire reader (harness code: the product's reader is
// fn2_datum_parse, pinned against the protobuf runtime by tests/test_sample_format.py).
class Datum {
 public:
  int channels() const { return channels_; }
  int height() const { return height_; }
  int width() const { return width_; }
  int label() const { return label_; }
  bool encoded() const { return encoded_; }
  const std::string& data() const { return data_; }
  int float_data_size() const { return (int)float_data_.size(); }
  float float_data(int i) const { return float_data_[i]; }
  bool ParseFromArray(const void* buf, int size) {
    *this = Datum();
    const unsigned char* p = static_cast<const unsigned char*>(buf);
    const unsigned char* end = p + size;
    auto varint = [&](unsigned long long* v) {
      *v = 0;
      for (int sh = 0; sh < 64 && p < end; sh += 7) { const unsigned char b = *p++; *v |= (unsigned long long)(b & 0x7f) << sh; if (!(b & 0x80)) return true; }
      return false;
    };
    auto f32 = [&](const unsigned char* q) { float f; __builtin_memcpy(&f, q, 4); return f; };
    while (p < end) {
      unsigned long long key, x;
      if (!varint(&key)) return false;
      const unsigned field = (unsigned)(key >> 3), wt = (unsigned)(key & 7);
      if (wt == 0) {
        if (!varint(&x)) return false;
        if (field == 1) channels_ = (int)x; else if (field == 2) height_ = (int)x; else if (field == 3) width_ = (int)x;
        else if (field == 5) label_ = (int)x; else if (field == 7) encoded_ = x != 0;
      } else if (wt == 2) {
        if (!varint(&x) || (unsigned long long)(end - p) < x) return false;
        if (field == 4) data_.assign(reinterpret_cast<const char*>(p), (size_t)x);
        if (field == 6) for (unsigned long long i = 0; i + 4 <= x; i += 4) float_data_.push_back(f32(p + i));
        p += x;
      } else if (wt == 5) {
        if (end - p < 4) return false;
        if (field == 6) float_data_.push_back(f32(p));
        p += 4;
      } else if (wt == 1) {
        if (end - p < 8) return false;
        p += 8;
      } else return false;
    }
    return true;
  }
 private:
  int channels_ = 0, height_ = 0, width_ = 0, label_ = 0;
  bool encoded_ = false;
  std::string data_;
  std::vector<float> float_data_;
};
// caffe.proto:5-22 (what Blob::FromProto / ToProto and the Layer constructor touch)
class BlobShape {
 public:
  int dim_size() const { return (int)dim_.size(); }
  long long dim(int i) const { return dim_[i]; }
  void add_dim(long long v) { dim_.push_back(v); }
  void clear_dim() { dim_.clear(); }
 private:
  std::vector<long long> dim_;
};
class BlobProto {
 public:
  bool has_shape() const { return has_shape_; }
  const BlobShape& shape() const { return shape_; }
  BlobShape* mutable_shape() { has_shape_ = true; return &shape_; }
  void clear_shape() { shape_ = BlobShape(); has_shape_ = false; }
  bool has_num() const { return has_legacy_; }
  bool has_channels() const { return has_legacy_; }
  bool has_height() const { return has_legacy_; }
  bool has_width() const { return has_legacy_; }
  int num() const { return legacy_[0]; }
  int channels() const { return legacy_[1]; }
  int height() const { return legacy_[2]; }
  int width() const { return legacy_[3]; }
  int data_size() const { return (int)data_.size(); }
  float data(int i) const { return data_[i]; }
  void add_data(float v) { data_.push_back(v); }
  void clear_data() { data_.clear(); }
  int diff_size() const { return (int)diff_.size(); }
  float diff(int i) const { return diff_[i]; }
  void add_diff(float v) { diff_.push_back(v); }
  void clear_diff() { diff_.clear(); }
  int double_data_size() const { return (int)double_data_.size(); }
  double double_data(int i) const { return double_data_[i]; }
  void add_double_data(double v) { double_data_.push_back(v); }
  void clear_double_data() { double_data_.clear(); }
  int double_diff_size() const { return (int)double_diff_.size(); }
  double double_diff(int i) const { return double_diff_[i]; }
  void add_double_diff(double v) { double_diff_.push_back(v); }
  void clear_double_diff() { double_diff_.clear(); }
 private:
  BlobShape shape_;
  bool has_shape_ = false, has_legacy_ = false;
  int legacy_[4] = {0, 0, 0, 0};
  std::vector<float> data_, diff_;
  std::vector<double> double_data_, double_diff_;
};


template <typename T>
class RepeatedField {
 public:
  typename std::vector<T>::const_iterator begin() const { return v_.begin(); }
  typename std::vector<T>::const_iterator end() const { return v_.end(); }
  int size() const { return (int)v_.size(); }
  T Get(int i) const { return v_[i]; }
  void Add(T x) { v_.push_back(x); }
 private:
  std::vector<T> v_;
};

enum DataParameter_DB { DataParameter_DB_LEVELDB = 0, DataParameter_DB_LMDB = 1 };
enum DataParameter_CHANNELENCODING { DataParameter_CHANNELENCODING_UINT8 = 1, DataParameter_CHANNELENCODING_UINT16FLOW = 2, DataParameter_CHANNELENCODING_BOOL1 = 3 };
enum DataParameter_RANDPERMORDER { DataParameter_RANDPERMORDER_FIRST_PERMUTE_THEN_RANGE = 0, DataParameter_RANDPERMORDER_FIRST_RANGE_THEN_PERMUTE = 1 };
class DataParameter {           // caffe.proto:918-986, the fields CustomDataLayer reads, with the proto defaults
 public:
  const std::string& source() const { return source_; }
  void set_source(const std::string& v) { source_ = v; }
  unsigned batch_size() const { return batch_size_; }
  void set_batch_size(unsigned v) { batch_size_ = v; }
  unsigned rand_skip() const { return 0; }
  DataParameter_DB backend() const { return backend_; }
  void set_backend(DataParameter_DB v) { backend_ = v; }
  float scale() const { return scale_; }
  void set_scale(float v) { scale_ = v; }
  bool has_mean_file() const { return false; }
  const std::string& mean_file() const { return mean_file_; }
  unsigned crop_size() const { return crop_size_; }
  void set_crop_size(unsigned v) { crop_size_ = v; }
  bool mirror() const { return false; }
  bool has_preselection_file() const { return false; }
  const std::string& preselection_file() const { return mean_file_; }
  bool has_preselection_label() const { return false; }
  int preselection_label() const { return 0; }
  int range_start() const { return range_start_; }
  void set_range_start(int v) { range_start_ = v; }
  int range_end() const { return range_end_; }
  void set_range_end(int v) { range_end_ = v; }
  bool rand_permute() const { return false; }
  DataParameter_RANDPERMORDER rand_permute_order() const { return DataParameter_RANDPERMORDER_FIRST_PERMUTE_THEN_RANGE; }
  unsigned rand_permute_seed() const { return 0; }
  const RepeatedField<unsigned>& slice_point() const { return slice_point_; }
  void add_slice_point(unsigned v) { slice_point_.Add(v); }
  const RepeatedField<int>& encoding() const { return encoding_; }
  void add_encoding(int v) { encoding_.Add(v); }
  bool verbose() const { return false; }
  const RepeatedField<float>& subtract() const { return subtract_; }
  void add_subtract(float v) { subtract_.Add(v); }
  unsigned permute_every_iter() const { return 0; }
  unsigned block_size() const { return 0; }
 private:
  std::string source_, mean_file_;
  unsigned batch_size_ = 1, crop_size_ = 0;
  DataParameter_DB backend_ = DataParameter_DB_LEVELDB;
  float scale_ = 1.f;
  int range_start_ = 0, range_end_ = -1;
  RepeatedField<unsigned> slice_point_;
  RepeatedField<int> encoding_;
  RepeatedField<float> subtract_;
};

// ---- augmentation messages (FlowAugmentation adapter + oracle/_ref) ------------------------------------------
// AugmentationCoeff, caffe.proto:436-486: 42 optional floats.  The layers move coefficients around as arrays through the
// protobuf reflection API (augmentation_layer_base.cpp:338-380): index = declaration order, value stored as log() when the field's
// default is non-zero.  The stand-in keeps the same order, defaults and has-bits behind a minimal Reflection / Descriptor.
#define FN2_AUG_COEFF_FIELDS(X)                                                                                              \
  X(mirror, 0) X(dx, 0) X(dy, 0) X(angle, 0) X(zoom_x, 1) X(zoom_y, 1)                                                        \
  X(gamma, 1) X(brightness, 0) X(contrast, 1) X(color1, 1) X(color2, 1) X(color3, 1)                                          \
  X(pow_nomean0, 1) X(pow_nomean1, 1) X(pow_nomean2, 1) X(add_nomean0, 0) X(add_nomean1, 0) X(add_nomean2, 0)                 \
  X(mult_nomean0, 1) X(mult_nomean1, 1) X(mult_nomean2, 1) X(pow_withmean0, 1) X(pow_withmean1, 1) X(pow_withmean2, 1)        \
  X(add_withmean0, 0) X(add_withmean1, 0) X(add_withmean2, 0) X(mult_withmean0, 1) X(mult_withmean1, 1) X(mult_withmean2, 1)  \
  X(lmult_pow, 1) X(lmult_add, 0) X(lmult_mult, 1) X(col_angle, 0)                                                            \
  X(fog_amount, 0) X(fog_size, 0) X(motion_blur_angle, 0) X(motion_blur_size, 0)                                              \
  X(shadow_angle, 0) X(shadow_distance, 0) X(shadow_strength, 0) X(noise, 0)
}  // namespace caffe
namespace google { namespace protobuf {
class FieldDescriptor {
 public:
  FieldDescriptor(int index, float def) : index_(index), default_(def) {}
  float default_value_float() const { return default_; }
  int index() const { return index_; }
 private:
  int index_;
  float default_;
};
class Descriptor {
 public:
  explicit Descriptor(std::vector<FieldDescriptor> f) : fields_(std::move(f)) {}
  int field_count() const { return (int)fields_.size(); }
  const FieldDescriptor* field(int i) const { return &fields_[i]; }
 private:
  std::vector<FieldDescriptor> fields_;
};
class Reflection {
 public:
  template <typename M> float GetFloat(const M& m, const FieldDescriptor* f) const { return m.has_[f->index()] ? m.value_[f->index()] : f->default_value_float(); }
  template <typename M> void SetFloat(M* m, const FieldDescriptor* f, float v) const { m->value_[f->index()] = v; m->has_[f->index()] = true; }
  template <typename M> void ClearField(M* m, const FieldDescriptor* f) const { m->has_[f->index()] = false; m->value_[f->index()] = f->default_value_float(); }
};
} }  // namespace google::protobuf
namespace caffe {
class AugmentationCoeff {
 public:
  enum { kNumFields = 42 };
  AugmentationCoeff() {
    int i = 0;
#define X(name, def) value_[i] = def; has_[i] = false; ++i;
    FN2_AUG_COEFF_FIELDS(X)
#undef X
  }
#define X(name, def)                                                                 \
  float name() const { return value_[k_##name]; }                                    \
  bool has_##name() const { return has_[k_##name]; }                                 \
  void set_##name(float v) { value_[k_##name] = v; has_[k_##name] = true; }          \
  void clear_##name() { value_[k_##name] = def; has_[k_##name] = false; }
  FN2_AUG_COEFF_FIELDS(X)
#undef X
  static const AugmentationCoeff& default_instance() { static const AugmentationCoeff d; return d; }
  const google::protobuf::Reflection* GetReflection() const { static const google::protobuf::Reflection r; return &r; }
  const google::protobuf::Descriptor* GetDescriptor() const {
    static const google::protobuf::Descriptor d([] {
      std::vector<google::protobuf::FieldDescriptor> f;
      int i = 0;
#define X(name, def) f.emplace_back(i++, (float)def);
      FN2_AUG_COEFF_FIELDS(X)
#undef X
      return f;
    }());
    return &d;
  }
 private:
  friend class google::protobuf::Reflection;
  enum {
#define X(name, def) k_##name,
    FN2_AUG_COEFF_FIELDS(X)
#undef X
    kCount
  };
  static_assert(kCount == kNumFields, "AugmentationCoeff has 42 fields");
  float value_[kNumFields];
  bool has_[kNumFields];
};

class CoeffScheduleParameter {       // caffe.proto:693-697
 public:
  float half_life() const { return half_life_; }
  float initial_coeff() const { return initial_coeff_; }
  float final_coeff() const { return final_coeff_; }
 private:
  float half_life_ = 1, initial_coeff_ = 1, final_coeff_ = 1;
};
class ParamSpec {                    // caffe.proto:284-308, the two setters DataAugmentationLayer::LayerSetUp calls
 public:
  void set_lr_mult(float v) { lr_mult_ = v; }
  void set_decay_mult(float v) { decay_mult_ = v; }
 private:
  float lr_mult_ = 1, decay_mult_ = 1;
};
class RandomGeneratorParameter {};   // caffe.proto:607-616: only handed to caffe_rng_generate, which the pins never reach
// AugmentationParameter, caffe.proto:489-546: crop size (read by the layers) and the generator sub-messages (named by
// augmentation_layer_base.cpp's generate_* functions, absent here: has_*() is false)
#define FN2_AUG_PARAM_GENERATORS(X)                                                                                         \
  X(mirror) X(translate) X(rotate) X(zoom) X(squeeze) X(translate_x) X(translate_y) X(gamma) X(brightness) X(contrast) X(color) \
  X(lmult_pow) X(lmult_mult) X(lmult_add) X(sat_pow) X(sat_mult) X(sat_add) X(col_pow) X(col_mult) X(col_add)                 \
  X(ladd_pow) X(ladd_mult) X(ladd_add) X(col_rotate) X(fog_amount) X(fog_size) X(motion_blur_angle) X(motion_blur_size)       \
  X(shadow_angle) X(shadow_distance) X(shadow_strength) X(noise)
class AugmentationParameter {
 public:
  unsigned crop_width() const { return crop_width_; }
  unsigned crop_height() const { return crop_height_; }
  bool has_crop_width() const { return has_crop_width_; }
  bool has_crop_height() const { return has_crop_height_; }
  void set_crop_width(unsigned v) { crop_width_ = v; has_crop_width_ = true; }
  void set_crop_height(unsigned v) { crop_height_ = v; has_crop_height_ = true; }
  // the fields DataAugmentationLayer reads (defaults of caffe.proto:492-505)
  bool has_write_augmented() const { return false; }
  const std::string& write_augmented() const { static const std::string e; return e; }
  float max_multiplier() const { return max_multiplier_; }
  void set_max_multiplier(float v) { max_multiplier_ = v; }
  bool augment_during_test() const { return false; }
  unsigned recompute_mean() const { return recompute_mean_; }
  void set_recompute_mean(unsigned v) { recompute_mean_ = v; }
  bool mean_per_pixel() const { return mean_per_pixel_; }
  void set_mean_per_pixel(bool v) { mean_per_pixel_ = v; }
  const RepeatedField<float>& mean() const { return mean_; }
  void add_mean(float v) { mean_.Add(v); }
  const RepeatedField<float>& chromatic_eigvec() const { return chromatic_eigvec_; }
  void add_chromatic_eigvec(float v) { chromatic_eigvec_.Add(v); }
#define X(name)                                                                                   \
  bool has_##name() const { return false; }                                                       \
  const RandomGeneratorParameter& name() const { static const RandomGeneratorParameter r; return r; }
  FN2_AUG_PARAM_GENERATORS(X)
#undef X
 private:
  unsigned crop_width_ = 0, crop_height_ = 0, recompute_mean_ = 0;
  bool has_crop_width_ = false, has_crop_height_ = false, mean_per_pixel_ = true;
  float max_multiplier_ = 255.f;
  RepeatedField<float> mean_, chromatic_eigvec_;
};

enum Phase { TRAIN = 0, TEST = 1 };

class LayerParameter {
 public:
  int blobs_size() const { return (int)blobs_.size(); }                     // = 7: trained blobs carried by the message (layer.hpp:46-52)
  const BlobProto& blobs(int i) const { return blobs_[i]; }
  BlobProto* add_blobs() { blobs_.emplace_back(); return &blobs_.back(); }
  void clear_blobs() { blobs_.clear(); }
  void Clear() { *this = LayerParameter(); }
  void CopyFrom(const LayerParameter& o) { *this = o; }
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
  const EltwiseParameter& eltwise_param() const { return eltwise_param_; }                   // stock layers (oracle/_ref only)
  EltwiseParameter* mutable_eltwise_param() { return &eltwise_param_; }
  const PowerParameter& power_param() const { return power_param_; }
  PowerParameter* mutable_power_param() { return &power_param_; }
  const ConvolutionParameter& convolution_param() const { return convolution_param_; }
  ConvolutionParameter* mutable_convolution_param() { return &convolution_param_; }
  const ConcatParameter& concat_param() const { return concat_param_; }
  ConcatParameter* mutable_concat_param() { return &concat_param_; }
  const ReLUParameter& relu_param() const { return relu_param_; }
  ReLUParameter* mutable_relu_param() { return &relu_param_; }
  const CoeffScheduleParameter& coeff_schedule_param() const { return coeff_schedule_param_; }   // = 148 (DataAugmentation, oracle/_ref only)
  ParamSpec* add_param() { param_.emplace_back(); return &param_.back(); }
  ParamSpec* mutable_param(int i) { return &param_[i]; }
  int param_size() const { return (int)param_.size(); }
  const AugmentationParameter& augmentation_param() const { return augmentation_param_; }     // = 149 (FlowAugmentation)
  AugmentationParameter* mutable_augmentation_param() { return &augmentation_param_; }
  const DataParameter& data_param() const { return data_param_; }                            // CustomData (oracle/_ref only)
  DataParameter* mutable_data_param() { return &data_param_; }
 private:
  DataParameter data_param_;
  AugmentationParameter augmentation_param_;
  CoeffScheduleParameter coeff_schedule_param_;
  std::vector<ParamSpec> param_;
  std::vector<BlobProto> blobs_;
  std::string name_, type_;
  std::vector<float> loss_weight_;
  bool reshape_every_iter_ = true;
  Phase phase_ = TEST;
  CorrelationParameter correlation_param_;
  L1LossParameter l1_loss_param_;
  ResampleParameter resample_param_;
  DownsampleParameter downsample_param_;
  FlowWarpParameter flow_warp_param_;
  EltwiseParameter eltwise_param_;
  PowerParameter power_param_;
  ConvolutionParameter convolution_param_;
  ReLUParameter relu_param_;
  ConcatParameter concat_param_;
};

}  // namespace caffe
