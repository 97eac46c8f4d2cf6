// Stand-in for include/caffe/layer_factory.hpp:53-141: registry keyed by the prototxt `type:` string.
#pragma once
#include <map>
#include "caffe/common.hpp"
#include "caffe/proto/caffe.pb.h"

namespace caffe {

template <typename Dtype> class Layer;

template <typename Dtype>
class LayerRegistry {
 public:
  typedef shared_ptr<Layer<Dtype> > (*Creator)(const LayerParameter&);
  typedef std::map<string, Creator> CreatorRegistry;
  static CreatorRegistry& Registry() { static CreatorRegistry* g = new CreatorRegistry(); return *g; }
  static void AddCreator(const string& type, Creator creator) {
    CreatorRegistry& registry = Registry();
    CHECK_EQ(registry.count(type), 0u) << "Layer type " << type << " already registered.";
    registry[type] = creator;
  }
  static shared_ptr<Layer<Dtype> > CreateLayer(const LayerParameter& param) {
    const string& type = param.type();
    CreatorRegistry& registry = Registry();
    CHECK_EQ(registry.count(type), 1u) << "Unknown layer type: " << type;
    return registry[type](param);
  }
  static vector<string> LayerTypeList() {
    vector<string> v;
    for (auto& kv : Registry()) v.push_back(kv.first);
    return v;
  }
};

template <typename Dtype>
class LayerRegisterer {
 public:
  LayerRegisterer(const string& type, shared_ptr<Layer<Dtype> > (*creator)(const LayerParameter&)) { LayerRegistry<Dtype>::AddCreator(type, creator); }
};

// Only the float instantiation exists in this stand-in (the tools of the reference use float, tools/caffe.cpp:203).
#define REGISTER_LAYER_CREATOR(type, creator) static LayerRegisterer<float> g_creator_f_##type(#type, creator<float>)
#define REGISTER_LAYER_CLASS(type)                                                    \
  template <typename Dtype>                                                           \
  shared_ptr<Layer<Dtype> > Creator_##type##Layer(const LayerParameter& param) {      \
    return shared_ptr<Layer<Dtype> >(new type##Layer<Dtype>(param));                  \
  }                                                                                   \
  REGISTER_LAYER_CREATOR(type, Creator_##type##Layer)

}  // namespace caffe
