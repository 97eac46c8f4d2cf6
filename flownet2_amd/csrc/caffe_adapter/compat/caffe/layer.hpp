// Stand-in for include/caffe/layer.hpp: the operator plug-in interface (same virtuals, same call order:
// SetUp = CheckBlobCounts -> LayerSetUp -> Reshape -> SetLossWeights, layer.hpp:69-76; Forward re-runs Reshape
// unless reshape_every_iter is false, :484-521), including the fork's additions (AllowBackward :322-324).
#pragma once
#include "caffe/blob.hpp"
#include "caffe/util/math_functions.hpp"   // layer.hpp:13 of the reference
#include "caffe/layer_factory.hpp"
#include "caffe/proto/caffe.pb.h"

namespace caffe {

template <typename Dtype> class Net;

template <typename Dtype>
class Layer {
 public:
  explicit Layer(const LayerParameter& param) : layer_param_(param) {}
  virtual ~Layer() {}

  void SetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CheckBlobCounts(bottom, top);
    LayerSetUp(bottom, top);
    Reshape(bottom, top);
    SetLossWeights(top);
  }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;

  inline Dtype Forward(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    Dtype loss = 0;
    if (layer_param_.reshape_every_iter()) Reshape(bottom, top);
    if (Caffe::mode() == Caffe::CPU) {
      Forward_cpu(bottom, top);
    } else {
      Forward_gpu(bottom, top);
    }
    for (size_t top_id = 0; top_id < top.size(); ++top_id) {
      if (!this->loss((int)top_id)) continue;
      const int count = top[top_id]->count();
      const Dtype* data = top[top_id]->cpu_data();
      const Dtype* w = top[top_id]->cpu_diff();
      for (int i = 0; i < count; ++i) loss += data[i] * w[i];
    }
    return loss;
  }
  inline void Backward(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>& bottom) {
    if (Caffe::mode() == Caffe::CPU) Backward_cpu(top, propagate_down, bottom);
    else Backward_gpu(top, propagate_down, bottom);
  }

  const LayerParameter& layer_param() const { return layer_param_; }
  vector<shared_ptr<Blob<Dtype> > >& blobs() { return blobs_; }          // learnable parameters (layer.hpp:126-128)
  virtual inline const char* type() const { return ""; }
  virtual inline int ExactNumBottomBlobs() const { return -1; }
  virtual inline int MinBottomBlobs() const { return -1; }
  virtual inline int MaxBottomBlobs() const { return -1; }
  virtual inline int ExactNumTopBlobs() const { return -1; }
  virtual inline int MinTopBlobs() const { return -1; }
  virtual inline int MaxTopBlobs() const { return -1; }
  virtual inline bool EqualNumBottomTopBlobs() const { return false; }
  virtual inline bool AutoTopBlobs() const { return false; }
  virtual inline bool AllowForceBackward(const int bottom_index) const { return true; }
  virtual inline bool AllowBackward() const { return true; }
  inline Dtype loss(const int top_index) const { return (int)loss_.size() > top_index ? loss_[top_index] : Dtype(0); }
  inline void set_loss(const int top_index, const Dtype value) {
    if ((int)loss_.size() <= top_index) loss_.resize(top_index + 1, Dtype(0));
    loss_[top_index] = value;
  }
  void SetNet(Net<Dtype>* net) { net_ = net; }
  Net<Dtype>* GetNet() { return net_; }

 protected:
  LayerParameter layer_param_;
  Phase phase_ = TEST;
  vector<shared_ptr<Blob<Dtype> > > blobs_;
  vector<bool> param_propagate_down_;
  vector<Dtype> loss_;

  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { return Forward_cpu(bottom, top); }
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>& bottom) = 0;
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, const vector<Blob<Dtype>*>& bottom) {
    Backward_cpu(top, propagate_down, bottom);
  }

  virtual void CheckBlobCounts(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    if (ExactNumBottomBlobs() >= 0) CHECK_EQ(ExactNumBottomBlobs(), (int)bottom.size()) << type() << " Layer takes " << ExactNumBottomBlobs() << " bottom blob(s) as input.";
    if (MinBottomBlobs() >= 0) CHECK_LE(MinBottomBlobs(), (int)bottom.size()) << type() << " Layer takes at least " << MinBottomBlobs() << " bottom blob(s) as input.";
    if (MaxBottomBlobs() >= 0) CHECK_GE(MaxBottomBlobs(), (int)bottom.size()) << type() << " Layer takes at most " << MaxBottomBlobs() << " bottom blob(s) as input.";
    if (ExactNumTopBlobs() >= 0) CHECK_EQ(ExactNumTopBlobs(), (int)top.size()) << type() << " Layer produces " << ExactNumTopBlobs() << " top blob(s) as output.";
    if (MinTopBlobs() >= 0) CHECK_LE(MinTopBlobs(), (int)top.size()) << type() << " Layer produces at least " << MinTopBlobs() << " top blob(s) as output.";
    if (MaxTopBlobs() >= 0) CHECK_GE(MaxTopBlobs(), (int)top.size()) << type() << " Layer produces at most " << MaxTopBlobs() << " top blob(s) as output.";
  }
  inline void SetLossWeights(const vector<Blob<Dtype>*>& top) {
    const int n = layer_param_.loss_weight_size();
    if (n) {
      CHECK_EQ((int)top.size(), n) << "loss_weight must be unspecified or specified once per top blob.";
      for (size_t top_id = 0; top_id < top.size(); ++top_id) {
        const Dtype w = layer_param_.loss_weight((int)top_id);
        if (w == Dtype(0)) continue;
        this->set_loss((int)top_id, w);
        Dtype* m = top[top_id]->mutable_cpu_diff();
        for (int i = 0; i < top[top_id]->count(); ++i) m[i] = w;
      }
    }
  }
 private:
  Net<Dtype>* net_ = nullptr;
  DISABLE_COPY_AND_ASSIGN(Layer);
};

}  // namespace caffe
