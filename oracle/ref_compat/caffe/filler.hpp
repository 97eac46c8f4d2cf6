// Stand-in for include/caffe/filler.hpp for the oracle/_ref build: only the "constant" filler the L1LossLayer's
// channel-sum convolution asks for (l1loss_layer.cpp:47-52); any other type aborts.
#pragma once
#include <string>

#include "caffe/blob.hpp"
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
  const std::string& type = param.type();
  if (type == "constant") return new ConstantFiller<Dtype>(param);
  CHECK(false) << "Unknown filler name: " << type << " (the oracle/_ref stand-in only has the constant filler)";
  return nullptr;
}

}  // namespace caffe
