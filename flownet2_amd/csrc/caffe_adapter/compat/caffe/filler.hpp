// Stand-in for include/caffe/filler.hpp (filler.hpp:18-290 of the reference): the interface the Convolution / Deconvolution plug-ins use
// to initialise their parameter blobs, with the "constant" filler only (the tests fill the blobs themselves).  In a Caffe tree the real
// header provides every filler, DiagonalFiller of the fork included (filler.hpp:265-290).
#pragma once
#include <string>

#include "caffe/blob.hpp"
#include "caffe/common.hpp"
#include "caffe/proto/caffe.pb.h"

namespace caffe {

template <typename Dtype>
class Filler {
 public:
  explicit Filler(const FillerParameter& param) : filler_param_(param) {}
  virtual ~Filler() {}
  virtual void Fill(Blob<Dtype>* blob) = 0;
 protected:
  FillerParameter filler_param_;
};

template <typename Dtype>
class ConstantFiller : public Filler<Dtype> {
 public:
  explicit ConstantFiller(const FillerParameter& param) : Filler<Dtype>(param) {}
  virtual void Fill(Blob<Dtype>* blob) {
    Dtype* data = blob->mutable_cpu_data();
    const Dtype value = this->filler_param_.value();
    for (int i = 0; i < blob->count(); ++i) data[i] = value;
  }
};

template <typename Dtype>
Filler<Dtype>* GetFiller(const FillerParameter& param) {
  CHECK(param.type() == "constant") << "Unknown filler name: " << param.type() << " (stand-in header: constant only)";
  return new ConstantFiller<Dtype>(param);
}

}  // namespace caffe
