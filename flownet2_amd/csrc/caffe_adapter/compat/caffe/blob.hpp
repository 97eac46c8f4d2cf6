// Stand-in for include/caffe/blob.hpp + syncedmem.hpp: NCHW fp32 tensor with data and diff, lazily mirrored
// between host and device (the head-state machine of syncedmem.cpp:25-77, reduced to what layers use).
#pragma once
#include <cstring>
#include <sstream>
#include <string>

#include "caffe/common.hpp"
#include "caffe/proto/caffe.pb.h"

namespace caffe {

const int kMaxBlobAxes = 32;      // blob.hpp:12

class SyncedMemory {
 public:
  explicit SyncedMemory(size_t size) : size_(size) {}
  ~SyncedMemory() { if (cpu_) std::free(cpu_); if (gpu_) (void)hipFree(static_cast<char*>(gpu_) - kFrontGuard); }
  const void* cpu_data() { to_cpu(); return cpu_; }
  const void* gpu_data() { to_gpu(); return gpu_; }
  void* mutable_cpu_data() { to_cpu(); head_ = HEAD_AT_CPU; return cpu_; }
  void* mutable_gpu_data() { to_gpu(); head_ = HEAD_AT_GPU; return gpu_; }
  size_t size() const { return size_; }
  enum SyncedHead { UNINITIALIZED, HEAD_AT_CPU, HEAD_AT_GPU, SYNCED };      // syncedmem.hpp:62-63
  SyncedHead head() { return head_; }
 private:
  typedef SyncedHead Head;
  void to_cpu() {
    if (!cpu_) { cpu_ = std::calloc(size_ ? size_ : 1, 1); }
    if (head_ == HEAD_AT_GPU) { CUDA_CHECK(hipMemcpy(cpu_, gpu_, size_, hipMemcpyDeviceToHost)); head_ = SYNCED; }
    if (head_ == UNINITIALIZED) head_ = HEAD_AT_CPU;
  }
  void to_gpu() {
    if (!gpu_) {
      void* base = nullptr;
      CUDA_CHECK(hipMalloc(&base, kFrontGuard + (size_ ? size_ : 1)));
      CUDA_CHECK(hipMemset(base, 0, kFrontGuard + (size_ ? size_ : 1)));
      gpu_ = static_cast<char*>(base) + kFrontGuard;
    }
    if (head_ == HEAD_AT_CPU) { CUDA_CHECK(hipMemcpy(gpu_, cpu_, size_, hipMemcpyHostToDevice)); head_ = SYNCED; }
    if (head_ == UNINITIALIZED) head_ = HEAD_AT_GPU;
  }
  // Zeroed bytes in front of every device buffer.  The reference's Correlation1D with single_direction = -1 reads up to
  // stride_2 * channels floats in FRONT of its scratch blob (correlation_layer1d.cu:89 with x_shift = -grid_width, :467-468).
  // With cudaMalloc's sub-allocator that is readable neighbouring memory; a bare hipMalloc block here starts a mapping and the
  // read faults.  The guard makes that read defined (zeros) so the layer can be run and pinned.
  static constexpr size_t kFrontGuard = 16384;
  void* cpu_ = nullptr;
  void* gpu_ = nullptr;
  size_t size_;
  Head head_ = UNINITIALIZED;
};

template <typename Dtype>
class Blob {
 public:
  Blob() {}
  Blob(int num, int channels, int height, int width) { Reshape(num, channels, height, width); }
  explicit Blob(const vector<int>& shape) { Reshape(shape); }
  void FromProto(const BlobProto&) { LOG(FATAL) << "Blob::FromProto: not part of the stand-in (mean_file is not used by the pins)"; }
  void Reshape(int num, int channels, int height, int width) { Reshape(vector<int>{num, channels, height, width}); }
  void Reshape(const vector<int>& shape) {
    size_t c = 1;
    for (int s : shape) { CHECK_GE(s, 0); c *= (size_t)s; }
    shape_ = shape;
    count_ = (int)c;
    if (c > capacity_) {                      // blob.cpp:36-40: reallocate only when growing
      capacity_ = c;
      data_.reset(new SyncedMemory(capacity_ * sizeof(Dtype)));
      diff_.reset(new SyncedMemory(capacity_ * sizeof(Dtype)));
    }
  }
  void ReshapeLike(const Blob& o) { Reshape(o.shape()); }
  void CopyFrom(const Blob& source, bool copy_diff = false, bool reshape = false) {      // blob.cpp:421-457, host copy
    if (source.count() != count_ || source.shape() != shape_) {
      if (reshape) ReshapeLike(source);
      else LOG(FATAL) << "Trying to copy blobs of different sizes.";
    }
    if (copy_diff) std::memcpy(mutable_cpu_diff(), source.cpu_diff(), sizeof(Dtype) * count_);
    else std::memcpy(mutable_cpu_data(), source.cpu_data(), sizeof(Dtype) * count_);
  }
  const vector<int>& shape() const { return shape_; }
  int shape(int i) const { return shape_[i < 0 ? i + (int)shape_.size() : i]; }
  int num_axes() const { return (int)shape_.size(); }
  int count() const { return count_; }
  int count(int start_axis, int end_axis) const {            // blob.hpp:84-98
    CHECK_LE(start_axis, end_axis); CHECK_GE(start_axis, 0); CHECK_LE(end_axis, num_axes());
    int c = 1;
    for (int i = start_axis; i < end_axis; ++i) c *= shape_[i];
    return c;
  }
  int count(int start_axis) const { return count(start_axis, num_axes()); }
  int CanonicalAxisIndex(int axis_index) const {             // blob.hpp:121-132
    CHECK_GE(axis_index, -num_axes()); CHECK_LT(axis_index, num_axes());
    return axis_index < 0 ? axis_index + num_axes() : axis_index;
  }
  std::string shape_string() const {
    std::ostringstream os;
    for (size_t i = 0; i < shape_.size(); ++i) os << shape_[i] << " ";
    os << "(" << count_ << ")";
    return os.str();
  }
  const int* gpu_shape() const {                              // device copy of the shape (used by the N-d im2col path only)
    if (!shape_data_ || shape_data_->size() != shape_.size() * sizeof(int)) shape_data_.reset(new SyncedMemory(shape_.size() * sizeof(int)));
    int* h = (int*)shape_data_->mutable_cpu_data();
    for (size_t i = 0; i < shape_.size(); ++i) h[i] = shape_[i];
    return (const int*)shape_data_->gpu_data();
  }
  int LegacyShape(int i) const { CHECK_LE(num_axes(), 4); return i < num_axes() ? shape_[i] : 1; }
  int num() const { return LegacyShape(0); }
  int channels() const { return LegacyShape(1); }
  int height() const { return LegacyShape(2); }
  int width() const { return LegacyShape(3); }
  int offset(int n, int c = 0, int h = 0, int w = 0) const { return ((n * channels() + c) * height() + h) * width() + w; }
  const Dtype* cpu_data() const { CHECK(data_); return (const Dtype*)data_->cpu_data(); }
  const Dtype* gpu_data() const { CHECK(data_); return (const Dtype*)data_->gpu_data(); }
  const Dtype* cpu_diff() const { CHECK(diff_); return (const Dtype*)diff_->cpu_data(); }
  const Dtype* gpu_diff() const { CHECK(diff_); return (const Dtype*)diff_->gpu_data(); }
  Dtype* mutable_cpu_data() { CHECK(data_); return (Dtype*)data_->mutable_cpu_data(); }
  Dtype* mutable_gpu_data() { CHECK(data_); return (Dtype*)data_->mutable_gpu_data(); }
  Dtype* mutable_cpu_diff() { CHECK(diff_); return (Dtype*)diff_->mutable_cpu_data(); }
  Dtype* mutable_gpu_diff() { CHECK(diff_); return (Dtype*)diff_->mutable_gpu_data(); }
  Dtype data_at(int n, int c, int h, int w) const { return cpu_data()[offset(n, c, h, w)]; }
  Dtype diff_at(int n, int c, int h, int w) const { return cpu_diff()[offset(n, c, h, w)]; }
  const shared_ptr<SyncedMemory>& data() const { CHECK(data_); return data_; }      // blob.hpp:209-212
  const shared_ptr<SyncedMemory>& diff() const { CHECK(diff_); return diff_; }
  void ShareData(const Blob& other) { CHECK_EQ(count_, other.count()); data_ = other.data_; }
  void ShareDiff(const Blob& other) { CHECK_EQ(count_, other.count()); diff_ = other.diff_; }
 protected:
  shared_ptr<SyncedMemory> data_, diff_;
  mutable shared_ptr<SyncedMemory> shape_data_;
  vector<int> shape_;
  int count_ = 0;
  size_t capacity_ = 0;
};

}  // namespace caffe
