// Shared host-side helpers for libflownet2_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/flownet2_hip.h"

#define FN2_API extern "C" __attribute__((visibility("default")))

namespace fn2 {

std::string& last_error();

// fn2_set_batch_invariant (api.cpp): with the flag on, every launch parameter that fixes a summation order and would otherwise depend
// on the batch size is computed for a batch of ONE sample (order_batch(N) == 1), so the bits of a sample do not depend on its batch.
int& batch_invariant_flag();
inline int order_batch(int N) { return batch_invariant_flag() ? 1 : N; }

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

// The reference checks every launch with CUDA_POST_KERNEL_CHECK = cudaPeekAtLastError
// (include/caffe/util/device_alternate.hpp:48-53,76); same here, but returned, not aborted.
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FN2_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return FN2_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned blocks_for(long long n, int threads, unsigned cap = 1u << 20) {
  long long b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > (long long)cap) b = cap;
  return (unsigned)b;
}

}  // namespace fn2
