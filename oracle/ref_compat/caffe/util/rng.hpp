// Stand-in for include/caffe/util/rng.hpp in the pin build (the prefetch thread only draws numbers for mirror / crop, both unused).
#pragma once
#include <random>
#include "caffe/common.hpp"
namespace caffe {
typedef std::mt19937 rng_t;
inline unsigned int caffe_rng_rand() { return 1u; }
}  // namespace caffe
