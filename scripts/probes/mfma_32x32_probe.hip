// v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 on gfx950:
//   (1) are both the k-ordered fmaf chain (exact fp32), so that a kernel may swap one for the other without changing a bit?
//   (2) matrix rate of each alone, and with ONE LDS operand read per MFMA (the small-map / GEMM kernels issue about one read or load per
//       MFMA; round 4 measured 17-26 cycles of matrix time per such instruction at one wave per SIMD): a 32x32x2 MFMA is twice the
//       flops per operand pair.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_32x32_probe.hip -o /tmp/m32 && /tmp/m32
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void exact32(const float* A, const float* B, float* D, int K) {      // A [32][K], B [K][32] -> D [32][32]
  const int l = threadIdx.x;
  f32x16 acc;
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l % 32) * K + k + l / 32], B[(k + l / 32) * 32 + l % 32], acc, 0, 0, 0);
  for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * (l / 32) + v % 4) * 32 + l % 32] = acc[v];
}
__global__ void exact16(const float* A, const float* B, float* D, int K) {      // A [16][K], B [K][16] -> D [16][16]
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * K + k + l / 16], B[(k + l / 16) * 16 + l % 16], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + l % 16] = acc[v];
}

template <int BIG, int READS>      // BIG: 32x32x2 (8 accumulators) or 16x16x4 (32 accumulators): the same 128 accumulator registers, the same flops per trip
__global__ void __launch_bounds__(256) rate(float* out, int iters, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
  __syncthreads();
  const float* p = lds + threadIdx.x;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float s = 0.f;
  if constexpr (BIG) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (READS) { float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"((unsigned)(size_t)p), "n"(4 * 256 * 0)); asm volatile("" :: "v"(t)); }
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
      }
    }
    for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
  } else {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 32; ++q) {          // 32 x 2048 flops = 16 x 4096
        if (READS) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)p)); asm volatile("" :: "v"(t)); }
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
      }
    }
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int BIG, int READS>
static void run(float* d, const char* what) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 60; ++w) hipLaunchKernelGGL((rate<BIG, READS>), dim3(256), dim3(256), 0, 0, d, iters, 1.0f);      // steady clocks
  hipEventRecord(e0, 0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((rate<BIG, READS>), dim3(256), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 10.0 * 256 * 4 * (double)iters * (BIG ? 8 * 4096.0 : 32 * 2048.0);
  printf("%-44s %8.1f TFLOP/s\n", what, flops / (ms * 1e-3) / 1e12);
}

int main() {
  const int K = 64;
  std::vector<float> A(32 * K), B(K * 32), D(32 * 32), R(32 * 32);
  srand(1);
  for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : B) v = (float)rand() / RAND_MAX - 0.5f;
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1 << 24);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(exact32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
  hipMemcpy(D.data(), dD, 1024 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float r = 0.f; for (int k = 0; k < K; ++k) r = fmaf(A[i * K + k], B[k * 32 + j], r); bad += memcmp(&r, &D[i * 32 + j], 4) != 0; }
  printf("v_mfma_f32_32x32x2_f32 vs the k-ordered fmaf chain over K = %d: %d of 1024 elements differ\n", K, bad);
  std::vector<float> A16(16 * K), B16(K * 16);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < K; ++k) A16[i * K + k] = A[i * K + k];
  for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) B16[k * 16 + j] = B[k * 32 + j];
  hipMemcpy(dA, A16.data(), A16.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B16.data(), B16.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(exact16, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
  std::vector<float> D16(256);
  hipMemcpy(D16.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) bad += memcmp(&D16[i * 16 + j], &D[i * 32 + j], 4) != 0;
  printf("v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32 on the shared 16x16 corner: %d of 256 elements differ\n", bad);
  run<0, 0>(dD, "16x16x4, MFMAs alone");
  run<1, 0>(dD, "32x32x2, MFMAs alone");
  run<0, 1>(dD, "16x16x4, one ds_read_b32 per MFMA");
  run<1, 1>(dD, "32x32x2, one ds_read_b32 per MFMA");
  return 0;
}
