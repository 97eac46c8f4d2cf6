// Stand-in for include/caffe/util/math_functions.hpp + src/caffe/util/math_functions.{cpp,cu}, used ONLY by the oracle/_ref
// build of the reference's L1LossLayer and of the stock layers it is composed from (Eltwise, Power, Convolution).  The
// reference implements these helpers on cuBLAS / CBLAS, neither of which is in this image; the stand-ins below are plain
// loops / plain HIP kernels with the same signatures and the textbook meaning of each BLAS call (fp32 accumulation in
// index order).  Header-only so that both Dtype instantiations of the reference's layers link.
#pragma once
#include <cmath>
#include <cstring>

#include "caffe/common.hpp"

enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112 };

namespace caffe {

// ---------------------------------------------------------------- host
inline void caffe_memset(const size_t N, const int alpha, void* X) { std::memset(X, alpha, N); }
inline void caffe_gpu_memset(const size_t N, const int alpha, void* X) { CUDA_CHECK(hipMemset(X, alpha, N)); }
// cuRAND in the reference (math_functions.cu:395-406); the pins never enable the noise effect
template <typename Dtype> inline void caffe_gpu_rng_gaussian(const int, const Dtype, const Dtype, Dtype*) { LOG(FATAL) << "caffe_gpu_rng_gaussian is not part of the pin harness"; }
inline void caffe_gpu_memcpy(const size_t N, const void* X, void* Y) { if (X != Y) CUDA_CHECK(hipMemcpy(Y, X, N, hipMemcpyDefault)); }   // math_functions.cu:94-98
template <typename Dtype> inline void caffe_set(const int N, const Dtype alpha, Dtype* X) { for (int i = 0; i < N; ++i) X[i] = alpha; }
template <typename Dtype> inline void caffe_copy(const int N, const Dtype* X, Dtype* Y) {   // math_functions.cpp:86-98
  if (X == Y) return;
  if (Caffe::mode() == Caffe::GPU) CUDA_CHECK(hipMemcpy(Y, X, sizeof(Dtype) * N, hipMemcpyDefault));
  else std::memcpy(Y, X, sizeof(Dtype) * N);
}
template <typename Dtype> inline void caffe_mul(const int N, const Dtype* a, const Dtype* b, Dtype* y) { for (int i = 0; i < N; ++i) y[i] = a[i] * b[i]; }
template <typename Dtype> inline void caffe_div(const int N, const Dtype* a, const Dtype* b, Dtype* y) { for (int i = 0; i < N; ++i) y[i] = a[i] / b[i]; }
template <typename Dtype> inline void caffe_scal(const int N, const Dtype alpha, Dtype* X) { for (int i = 0; i < N; ++i) X[i] *= alpha; }
template <typename Dtype> inline void caffe_axpy(const int N, const Dtype alpha, const Dtype* X, Dtype* Y) { for (int i = 0; i < N; ++i) Y[i] += alpha * X[i]; }
template <typename Dtype> inline void caffe_cpu_axpby(const int N, const Dtype alpha, const Dtype* X, const Dtype beta, Dtype* Y) {
  for (int i = 0; i < N; ++i) Y[i] = alpha * X[i] + beta * Y[i];
}
template <typename Dtype> inline void caffe_cpu_scale(const int N, const Dtype alpha, const Dtype* x, Dtype* y) { for (int i = 0; i < N; ++i) y[i] = alpha * x[i]; }
template <typename Dtype> inline void caffe_add_scalar(const int N, const Dtype alpha, Dtype* X) { for (int i = 0; i < N; ++i) X[i] += alpha; }
template <typename Dtype> inline void caffe_powx(const int N, const Dtype* a, const Dtype b, Dtype* y) { for (int i = 0; i < N; ++i) y[i] = std::pow(a[i], b); }
// C = alpha * op(A) * op(B) + beta * C, row-major (the reference calls cblas_sgemm with CblasRowMajor)
template <typename Dtype>
inline void caffe_cpu_gemm(const CBLAS_TRANSPOSE TransA, const CBLAS_TRANSPOSE TransB, const int M, const int N, const int K,
                           const Dtype alpha, const Dtype* A, const Dtype* B, const Dtype beta, Dtype* C) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      Dtype acc = 0;
      for (int k = 0; k < K; ++k) acc += (TransA == CblasNoTrans ? A[m * K + k] : A[k * M + m]) * (TransB == CblasNoTrans ? B[k * N + n] : B[n * K + k]);
      C[m * N + n] = alpha * acc + (beta == Dtype(0) ? Dtype(0) : beta * C[m * N + n]);
    }
}
template <typename Dtype>
inline void caffe_cpu_gemv(const CBLAS_TRANSPOSE TransA, const int M, const int N, const Dtype alpha, const Dtype* A, const Dtype* x,
                           const Dtype beta, Dtype* y) {
  const int rows = TransA == CblasNoTrans ? M : N, cols = TransA == CblasNoTrans ? N : M;
  for (int r = 0; r < rows; ++r) {
    Dtype acc = 0;
    for (int c = 0; c < cols; ++c) acc += (TransA == CblasNoTrans ? A[r * N + c] : A[c * N + r]) * x[c];
    y[r] = alpha * acc + (beta == Dtype(0) ? Dtype(0) : beta * y[r]);
  }
}

// ---------------------------------------------------------------- device
namespace refmath {
template <typename Dtype, typename F>
__global__ void map_kernel(const int n, F f) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) f(i); }
inline int blocks(int n) { int b = (n + 255) / 256; return b < 1 ? 1 : (b > 4096 ? 4096 : b); }
template <typename Dtype>
__global__ void gemm_kernel(const bool ta, const bool tb, const int M, const int N, const int K, const Dtype alpha, const Dtype* A,
                            const Dtype* B, const Dtype beta, Dtype* C) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (long long)M * N; i += (long long)blockDim.x * gridDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    Dtype acc = 0;
    for (int k = 0; k < K; ++k) acc += (ta ? A[(size_t)k * M + m] : A[(size_t)m * K + k]) * (tb ? B[(size_t)n * K + k] : B[(size_t)k * N + n]);
    C[i] = alpha * acc + (beta == Dtype(0) ? Dtype(0) : beta * C[i]);
  }
}
template <typename Dtype>
__global__ void dot_kernel(const int n, const Dtype* x, const Dtype* y, Dtype* out) {      // one block; fixed order
  __shared__ Dtype part[256];
  Dtype acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += x[i] * y[i];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) *out = part[0];
}
}  // namespace refmath

#define FN2_REF_MAP(n, ...) hipLaunchKernelGGL((refmath::map_kernel<Dtype>), dim3(refmath::blocks(n)), dim3(256), 0, 0, n, __VA_ARGS__); CUDA_POST_KERNEL_CHECK
template <typename Dtype> inline void caffe_gpu_set(const int N, const Dtype alpha, Dtype* X) { FN2_REF_MAP(N, [=] __device__(int i) { X[i] = alpha; }); }
template <typename Dtype> inline void caffe_gpu_mul(const int N, const Dtype* a, const Dtype* b, Dtype* y) { FN2_REF_MAP(N, [=] __device__(int i) { y[i] = a[i] * b[i]; }); }
template <typename Dtype> inline void caffe_gpu_div(const int N, const Dtype* a, const Dtype* b, Dtype* y) { FN2_REF_MAP(N, [=] __device__(int i) { y[i] = a[i] / b[i]; }); }
template <typename Dtype> inline void caffe_gpu_scal(const int N, const Dtype alpha, Dtype* X) { FN2_REF_MAP(N, [=] __device__(int i) { X[i] *= alpha; }); }
template <typename Dtype> inline void caffe_gpu_axpy(const int N, const Dtype alpha, const Dtype* X, Dtype* Y) { FN2_REF_MAP(N, [=] __device__(int i) { Y[i] += alpha * X[i]; }); }
template <typename Dtype> inline void caffe_gpu_axpby(const int N, const Dtype alpha, const Dtype* X, const Dtype beta, Dtype* Y) {
  // math_functions.cu:90-101: scal(beta, Y) then axpy(alpha, X, Y)
  FN2_REF_MAP(N, [=] __device__(int i) { Y[i] = Y[i] * beta; Y[i] += alpha * X[i]; });
}
template <typename Dtype> inline void caffe_gpu_scale(const int N, const Dtype alpha, const Dtype* x, Dtype* y) { FN2_REF_MAP(N, [=] __device__(int i) { y[i] = alpha * x[i]; }); }
template <typename Dtype> inline void caffe_gpu_add_scalar(const int N, const Dtype alpha, Dtype* X) { FN2_REF_MAP(N, [=] __device__(int i) { X[i] += alpha; }); }
template <typename Dtype> inline void caffe_gpu_powx(const int N, const Dtype* a, const Dtype b, Dtype* y) { FN2_REF_MAP(N, [=] __device__(int i) { y[i] = pow(a[i], b); }); }
template <typename Dtype> inline void caffe_gpu_dot(const int N, const Dtype* x, const Dtype* y, Dtype* out) {
  Dtype* d = nullptr;
  CUDA_CHECK(hipMalloc(&d, sizeof(Dtype)));
  hipLaunchKernelGGL((refmath::dot_kernel<Dtype>), dim3(1), dim3(256), 0, 0, N, x, y, d);
  CUDA_POST_KERNEL_CHECK;
  CUDA_CHECK(hipMemcpy(out, d, sizeof(Dtype), hipMemcpyDeviceToHost));
  CUDA_CHECK(hipFree(d));
}
template <typename Dtype>
inline void caffe_gpu_gemm(const CBLAS_TRANSPOSE TransA, const CBLAS_TRANSPOSE TransB, const int M, const int N, const int K,
                           const Dtype alpha, const Dtype* A, const Dtype* B, const Dtype beta, Dtype* C) {
  hipLaunchKernelGGL((refmath::gemm_kernel<Dtype>), dim3(refmath::blocks(M * N)), dim3(256), 0, 0, TransA != CblasNoTrans, TransB != CblasNoTrans,
                     M, N, K, alpha, A, B, beta, C);
  CUDA_POST_KERNEL_CHECK;
}
template <typename Dtype>
inline void caffe_gpu_gemv(const CBLAS_TRANSPOSE TransA, const int M, const int N, const Dtype alpha, const Dtype* A, const Dtype* x,
                           const Dtype beta, Dtype* y) {
  // y = alpha * op(A) x + beta y with A stored row-major M x N: a GEMM with one column
  if (TransA == CblasNoTrans) caffe_gpu_gemm<Dtype>(CblasNoTrans, CblasNoTrans, M, 1, N, alpha, A, x, beta, y);
  else caffe_gpu_gemm<Dtype>(CblasTrans, CblasNoTrans, N, 1, M, alpha, A, x, beta, y);
}

}  // namespace caffe
