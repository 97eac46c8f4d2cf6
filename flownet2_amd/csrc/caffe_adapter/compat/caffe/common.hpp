// Stand-in for include/caffe/common.hpp + util/device_alternate.hpp of the reference: just enough of the
// Caffe runtime (glog-style CHECK/LOG, mode, launch-configuration macros) to compile and RUN layer plug-ins
// outside a Caffe tree -- our adapter (flownet2_amd/csrc/caffe_adapter/) and, for the oracle pin, the
// reference's own layer sources (oracle/ref_build.sh).  Written for HIP on gfx950; CUDA runtime names the
// reference sources use are mapped onto their HIP twins here (test scaffolding only -- the product kernels
// never see these macros).
#pragma once
#include <random>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace caffe {

using std::string;
using std::vector;
using std::shared_ptr;     // boost::shared_ptr in the reference (common.hpp:80)
using std::ostringstream;

// ---- glog look-alikes: a failed CHECK / LOG(FATAL) throws caffe::Fatal (the reference aborts) -------------
struct Fatal : std::runtime_error { using std::runtime_error::runtime_error; };

class LogMessage {
 public:
  LogMessage(const char* file, int line, bool fatal, bool silent) : fatal_(fatal), silent_(silent) { s_ << file << ":" << line << "] "; }
  ~LogMessage() noexcept(false) {
    if (fatal_) {
      if (std::getenv("FN2_CAFFE_LOG")) std::cerr << "FATAL " << s_.str() << std::endl;
      throw Fatal(s_.str());
    }
    if (!silent_ && std::getenv("FN2_CAFFE_LOG")) std::cerr << s_.str() << std::endl;
  }
  std::ostream& stream() { return s_; }
 private:
  std::ostringstream s_;
  bool fatal_, silent_;
};
struct LogVoidify { void operator&(std::ostream&) {} };

#define FN2_LOG_INFO ::caffe::LogMessage(__FILE__, __LINE__, false, false).stream()
#define FN2_LOG_WARNING ::caffe::LogMessage(__FILE__, __LINE__, false, false).stream()
#define FN2_LOG_ERROR ::caffe::LogMessage(__FILE__, __LINE__, false, false).stream()
#define FN2_LOG_FATAL ::caffe::LogMessage(__FILE__, __LINE__, true, false).stream()
#define LOG(sev) FN2_LOG_##sev
#define DLOG(sev) LOG(sev)
#define LOG_IF(sev, cond) !(cond) ? (void)0 : ::caffe::LogVoidify() & LOG(sev)
#define CHECK(cond) (cond) ? (void)0 : ::caffe::LogVoidify() & ::caffe::LogMessage(__FILE__, __LINE__, true, false).stream() << "Check failed: " #cond " "
#define FN2_CHECK_OP(a, b, op) ((a) op (b)) ? (void)0 : ::caffe::LogVoidify() & ::caffe::LogMessage(__FILE__, __LINE__, true, false).stream() \
    << "Check failed: " #a " " #op " " #b " (" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) FN2_CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) FN2_CHECK_OP(a, b, !=)
#define CHECK_LE(a, b) FN2_CHECK_OP(a, b, <=)
#define CHECK_LT(a, b) FN2_CHECK_OP(a, b, <)
#define CHECK_GE(a, b) FN2_CHECK_OP(a, b, >=)
#define CHECK_GT(a, b) FN2_CHECK_OP(a, b, >)
#define DCHECK(c) CHECK(c)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
#define NOT_IMPLEMENTED LOG(FATAL) << "Not Implemented Yet"

// ---- common.hpp:24-66 ------------------------------------------------------------------------------------
#define DISABLE_COPY_AND_ASSIGN(classname) \
 private:                                  \
  classname(const classname&);             \
  classname& operator=(const classname&)
#define INSTANTIATE_CLASS(classname) \
  char gInstantiationGuard##classname; \
  template class classname<float>
#define INSTANTIATE_LAYER_GPU_FORWARD(classname) \
  template void classname<float>::Forward_gpu(const std::vector<Blob<float>*>& bottom, const std::vector<Blob<float>*>& top)
#define INSTANTIATE_LAYER_GPU_BACKWARD(classname) \
  template void classname<float>::Backward_gpu(const std::vector<Blob<float>*>& top, const std::vector<bool>& propagate_down, \
                                               const std::vector<Blob<float>*>& bottom)
#define INSTANTIATE_LAYER_GPU_FUNCS(classname) \
  INSTANTIATE_LAYER_GPU_FORWARD(classname);    \
  INSTANTIATE_LAYER_GPU_BACKWARD(classname)

class Caffe {
 public:
  enum Brew { CPU, GPU };
  static Brew& mode_ref() { static thread_local Brew m = GPU; return m; }
  static Brew mode() { return mode_ref(); }
  static void set_mode(Brew m) { mode_ref() = m; }
  // common.hpp:118-130 of the reference: a seeded generator handed out as void* (data layers only)
  class RNG {
   public:
    RNG() : g_(1) {}
    explicit RNG(unsigned int seed) : g_(seed) {}
    void* generator() { return &g_; }
   private:
    std::mt19937 g_;
  };
};

// ---- util/device_alternate.hpp:40-90 (CUDA names -> HIP) ---------------------------------------------------
#define CUDA_CHECK(condition)                                                              \
  do {                                                                                     \
    hipError_t error = (condition);                                                        \
    CHECK_EQ(error, hipSuccess) << " " << hipGetErrorString(error);                        \
  } while (0)
#define CUDA_KERNEL_LOOP(i, n) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)
#define CUDA_POST_KERNEL_CHECK CUDA_CHECK(hipPeekAtLastError())
const int CAFFE_CUDA_NUM_THREADS = 512;
inline int CAFFE_GET_BLOCKS(const int N) { return (N + CAFFE_CUDA_NUM_THREADS - 1) / CAFFE_CUDA_NUM_THREADS; }

#define cudaMemset hipMemset
#define cudaMemcpy hipMemcpy
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaPeekAtLastError hipPeekAtLastError

}  // namespace caffe
