// Stand-in for include/caffe/util/math_functions.hpp: the handful of helpers the hot-path layers call.
#pragma once
#include <cstring>
#include "caffe/common.hpp"

namespace caffe {
inline void caffe_memset(const size_t N, const int alpha, void* X) { std::memset(X, alpha, N); }
inline void caffe_gpu_memset(const size_t N, const int alpha, void* X) { CUDA_CHECK(hipMemset(X, alpha, N)); }
template <typename Dtype> inline void caffe_set(const int N, const Dtype alpha, Dtype* X) { for (int i = 0; i < N; ++i) X[i] = alpha; }
template <typename Dtype> inline void caffe_copy(const int N, const Dtype* X, Dtype* Y) {   // math_functions.cpp:86-98
  if (X == Y) return;
  if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipMemcpy(Y, X, sizeof(Dtype) * N, hipMemcpyDefault));
  else std::memcpy(Y, X, sizeof(Dtype) * N);
}
}  // namespace caffe
