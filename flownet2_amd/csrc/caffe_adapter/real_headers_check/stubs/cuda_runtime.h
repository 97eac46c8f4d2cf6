// third-party stand-in (compile-only check): the CUDA runtime names the reference's HEADERS use, mapped onto HIP
#pragma once
#include <hip/hip_runtime.h>
typedef hipError_t cudaError_t;
typedef hipStream_t cudaStream_t;
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaPeekAtLastError hipPeekAtLastError
inline hipError_t cudaMallocHost(void** p, size_t n) { return hipHostMalloc(p, n, 0); }
inline hipError_t cudaFreeHost(void* p) { return hipHostFree(p); }
#define cudaMemset hipMemset
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDefault hipMemcpyDefault
