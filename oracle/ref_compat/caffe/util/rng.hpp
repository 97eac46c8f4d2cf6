// Stand-in for include/caffe/util/rng.hpp in the pin build (the prefetch thread only draws numbers for mirror / crop, both unused).
#pragma once
#include <cmath>
#include <random>
#include "caffe/common.hpp"
#include "caffe/proto/caffe.pb.h"
namespace caffe {
typedef std::mt19937 rng_t;
inline unsigned int caffe_rng_rand() { return 1u; }
// include/caffe/util/rng.hpp:22 -- named by augmentation_layer_base.cpp's generate_* functions, which the pins never call
template <typename Dtype, typename Randtype>
inline Randtype caffe_rng_generate(const RandomGeneratorParameter&, Dtype discount_coeff = 1, Dtype prob0_value = NAN) {
  LOG(FATAL) << "caffe_rng_generate is not part of the pin harness";
  return Randtype();
}
}  // namespace caffe
